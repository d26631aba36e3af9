#!/bin/bash
# round 5: does taking the f32 kernel's half-rate-by-SGPR instructions to the full rate pay on the
# driver-shaped run?  Alternates, on ONE box, the product build (six slider uniforms parked in VGPRs + amdgpu_waves_per_eu(8,8):
# 1290 issue cycles per tile) with round 4's arrangement (tools/librawdev_r5base.so: 1374) and the attribute alone
# (tools/librawdev_r5nopark.so), headline workload (256 x 24 MP f32, 20 steps); then the narrow surfaces, which the attribute
# touches too (batch of 64).     bash tools/gpu_r5_f32ab.sh [tag] [rounds]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r5f32ab}; mkdir -p "$OUT"
N=${2:-3}
run() { # name lib args...
  local name=$1 lib=$2; shift 2
  ( [ -n "$lib" ] && export RAWDEV_LIB=$lib
    timeout -k 10 300 python3 "$ROOT/bench.py" --no-extra --no-cpu-baseline --no-alt-math "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" )
  local rc=$?; [ $rc -ge 124 ] && exit $rc
  python3 -c "
import json,sys
d=json.load(open('$OUT/$name.json')); r=d['roofline']
print('%-22s %9.1f MP/s  %7.2f us/frame  frac %.4f  valu %s' % ('$name', d['value'], r['us_per_frame'], r['frac'], r.get('valu_issue_frac')))"
}
for i in $(seq 1 $N); do
  run f32_parked_$i $ROOT/tools/librawdev_r5park.so
  run f32_r4base_$i $ROOT/tools/librawdev_r5base.so
  run f32_nopark_$i $ROOT/tools/librawdev_r5nopark.so
done
for fmt in u8 f16; do
  for i in 1 2; do
    run ${fmt}_product_$i "" --format $fmt --frames 64 --ring 32 --steps 10
    run ${fmt}_r4base_$i $ROOT/tools/librawdev_r5base.so --format $fmt --frames 64 --ring 32 --steps 10
  done
done
