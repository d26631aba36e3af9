#!/bin/bash
# round 5: the workgroup prologue in one phase (all table loads in flight, histogram zeroed meanwhile, one barrier) against
# the three-phase form (tools/librawdev_r5prologue3.so = the commit before), on the launch-heavy shapes: BASELINE config 5 as
# worded (8 row-band launches per 100 MP f16 frame), one launch per 24 MP frame (RGBA8, f16), and the single-frame render.
# The old library is built from the parent of commit 674b3a9 ("Export kernel: one-phase workgroup prologue"):
#   git worktree add /tmp/before 674b3a9^ && hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared \
#       -std=c++17 -pthread -ldl -o tools/librawdev_r5prologue3.so /tmp/before/raweditor_amd/csrc/rawdev.hip
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r5prologue}; mkdir -p "$OUT"
OLD=$ROOT/tools/librawdev_r5prologue3.so
[ -f "$OLD" ] || { echo "build $OLD first (see the head of this script)"; exit 1; }
show() { python3 -c "
import json
d=json.load(open('$OUT/$1.json')); r=d['roofline']
print('%-30s %9.1f MP/s  %7.2f us/frame  launch %.2f us' % ('$1', d['value'], r['us_per_frame'], r['launch_us']))"; }
C5="--format f16 --width 11648 --height 8736 --frames 16 --ring 4 --row-bands 8 --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-alt-math --no-box"
P1="--frames 64 --ring 32 --steps 8 --warmup 2 --no-extra --no-cpu-baseline --no-alt-math --no-box"
for i in 1 2 3; do
  for v in new old; do
    ( [ $v = old ] && export RAWDEV_LIB=$OLD; export RD_BATCH_PERSISTENT=0
      timeout -k 10 300 python3 "$ROOT/bench.py" $C5 > "$OUT/c5_tiled_${v}_$i.json" 2> "$OUT/c5_tiled_${v}_$i.err" ); rc=$?; [ $rc -ge 124 ] && exit $rc; show c5_tiled_${v}_$i
    ( [ $v = old ] && export RAWDEV_LIB=$OLD; export RD_BATCH_PERSISTENT=0
      timeout -k 10 300 python3 "$ROOT/bench.py" --format u8 $P1 > "$OUT/u8_perframe_${v}_$i.json" 2> "$OUT/u8_perframe_${v}_$i.err" ); rc=$?; [ $rc -ge 124 ] && exit $rc; show u8_perframe_${v}_$i
    ( [ $v = old ] && export RAWDEV_LIB=$OLD; export RD_BATCH_PERSISTENT=0
      timeout -k 10 300 python3 "$ROOT/bench.py" --format f16 $P1 > "$OUT/f16_perframe_${v}_$i.json" 2> "$OUT/f16_perframe_${v}_$i.err" ); rc=$?; [ $rc -ge 124 ] && exit $rc; show f16_perframe_${v}_$i
  done
done
