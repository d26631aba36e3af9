#!/bin/bash
# Round-3 evidence pass on one GPU box (tools/gpu_check.sh's steps + the new entry points + fuzz).
#   bash tools/gpu_r3_evidence.sh [tag]
set -u
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() {   # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s) -> $log"
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    local rc=$?
    tail -n 3 "$log" | cut -c1-400
    echo "== rc=$rc"
    if [ $rc -ge 124 ]; then echo "step killed/timed out: stopping"; exit $rc; fi
    return $rc
}
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
step 300 "$OUT/smoke.log" python __graft_entry__.py --smoke
step 900 "$OUT/pytest_gpu.log" python -m pytest tests -m gpu -x -q
step 600 "$OUT/bench.log" python bench.py
grep -E '^\{' "$OUT/bench.log" > "$OUT/bench.json" || true
step 300 "$OUT/bench_static.log" python bench.py --static-descriptors --no-extra --no-cpu-baseline --no-alt-math
step 300 "$OUT/bench_node1.log" python bench.py --host node --gpus 1 --no-cpu-baseline
RD_NODE_REDUCE=host step 300 "$OUT/bench_node2_rehearsal.log" python bench.py --host node --gpus 2 --frames 64 --steps 5 --no-cpu-baseline
RAWDEV_DIST_BACKEND=gloo step 300 "$OUT/bench_ranks2_rehearsal.log" python bench.py --gpus 2 --frames 64 --steps 5 --no-cpu-baseline
( cd /tmp && step 600 "$OLDPWD/$OUT/rocprof_run.log" rocprofv3 --kernel-trace --stats --output-format csv \
    -d "$OLDPWD/$OUT/rocprof" -- python3 "$OLDPWD/bench.py" --no-cpu-baseline )
for f in $(find "$OUT/rocprof" -name "*kernel_stats*.csv" | head -1); do head -14 "$f" | cut -c1-200; done
step 600 "$OUT/fuzz.log" python -m tests.fuzz_parity 100000 33
echo "== done"
