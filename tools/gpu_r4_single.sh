#!/bin/bash
# round 4, item 5: the single-frame enqueue with and without the two-node graph, then its kernel + HIP API trace
set -u
OUT=gpurun_out/${1:-r4c}
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/single_gap.txt"
for rep in 1 2; do
  for g in 0 1; do
    RD_GRAPH=$g timeout -k 10 120 python tools/bench_single_gap.py >> "$OUT/single_gap.txt" 2>&1 || { echo "bench_single_gap failed (RD_GRAPH=$g)"; tail -5 "$OUT/single_gap.txt"; exit 1; }
  done
done
grep -v amdgpu.ids "$OUT/single_gap.txt"
REPO=$PWD
for g in 0 1; do
  ( cd /tmp && RD_GRAPH=$g ITERS=30 timeout -k 10 300 rocprofv3 --kernel-trace --hip-trace --output-format csv -d "$REPO/$OUT/trace_g$g" -- python3 "$REPO/tools/bench_single_gap.py" > "$REPO/$OUT/trace_g$g.log" 2>&1 ) || { echo "rocprofv3 failed (RD_GRAPH=$g)"; tail -5 "$OUT/trace_g$g.log"; exit 1; }
done
find "$OUT" -name "*kernel_trace.csv" | head; find "$OUT" -name "*hip_api_trace.csv" | head
