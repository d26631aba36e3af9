#!/bin/bash
# HBM traffic of the export kernel from the L2 memory-side counters, as MI355X_MICROARCH.md section HBM
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), with
# --kernel-trace only.  Two variants: RD_BURST=0 (every CFA line read exactly once: the calibration
# case for our 4-B-per-lane loads) and RD_BURST=1 (the shipped f32 configuration).
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for BURST in 0 1; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    echo "== RD_BURST=$BURST $CTR"
    RD_BURST=$BURST timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTR --output-format csv \
        -d "$OUT/b${BURST}_$CTR" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --frames 32 --no-cpu-baseline \
        > "$OUT/b${BURST}_$CTR.log" 2>&1
    rc=$?; echo "rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
  done
done
cd "$ROOT" && python3 tools/parse_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt"
