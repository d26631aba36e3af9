#!/bin/bash
# round 5: where the batch's buffers live.  The headline with (a) one torch allocation per CFA plane and per surface (rounds
# 1-4), (b) the planes in one arena, (c) planes and output ring in one arena each (the default now), alternating on one box;
# then BASELINE config 5's literally tiled form with one and with two alternating streams.   bash tools/gpu_r5_arena.sh [tag] [rounds]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r5arena}; mkdir -p "$OUT"
N=${2:-3}
show() { python3 -c "
import json
d=json.load(open('$OUT/$1.json')); r=d['roofline']
print('%-26s %9.1f MP/s  %7.2f us/frame  launch %.1f us' % ('$1', d['value'], r['us_per_frame'], r['launch_us']))"; }
for i in $(seq 1 $N); do
  for v in "separate:--plane-stagger -1 --ring-arena 0" "planes:--plane-stagger 0 --ring-arena 0" "both:--plane-stagger 0 --ring-arena 1"; do
    name=${v%%:*}_$i; args=${v#*:}
    timeout -k 10 300 python3 "$ROOT/bench.py" --no-extra --no-cpu-baseline --no-alt-math $args > "$OUT/$name.json" 2> "$OUT/$name.err"
    rc=$?; [ $rc -ge 124 ] && exit $rc
    show $name
  done
done
C5="--format f16 --width 11648 --height 8736 --frames 16 --ring 4 --row-bands 8 --steps 4 --warmup 1 --no-extra --no-cpu-baseline --no-alt-math --no-box"
for i in 1 2; do
  RD_BATCH_PERSISTENT=0 timeout -k 10 300 python3 "$ROOT/bench.py" $C5 > "$OUT/c5_tiled_1stream_$i.json" 2> "$OUT/c5_tiled_1stream_$i.err"; rc=$?; [ $rc -ge 124 ] && exit $rc; show c5_tiled_1stream_$i
  RD_BATCH_PERSISTENT=0 RD_BATCH_STREAMS=2 timeout -k 10 300 python3 "$ROOT/bench.py" $C5 > "$OUT/c5_tiled_2streams_$i.json" 2> "$OUT/c5_tiled_2streams_$i.err"; rc=$?; [ $rc -ge 124 ] && exit $rc; show c5_tiled_2streams_$i
  timeout -k 10 300 python3 "$ROOT/bench.py" $C5 > "$OUT/c5_default_$i.json" 2> "$OUT/c5_default_$i.err"; rc=$?; [ $rc -ge 124 ] && exit $rc; show c5_default_$i
done
