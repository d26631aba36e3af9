#!/bin/bash
# One GPU-box pass: smoke -> pytest -m gpu -> bench -> rocprofv3 kernel trace of a short bench.
# Usage (from the repo root, via gpurun):  bash tools/gpu_check.sh [tag]
# A step that times out or is killed (rc >= 124) ends the script: no further GPU step is started.
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() {   # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s) -> $log"
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    local rc=$?
    tail -n 6 "$log"
    echo "== rc=$rc"
    if [ $rc -ge 124 ]; then echo "step killed/timed out: stopping"; exit $rc; fi
    return $rc
}
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4
nproc
step 300 "$OUT/smoke.log" python __graft_entry__.py --smoke
step 900 "$OUT/pytest_gpu.log" python -m pytest tests -m gpu -x -q
step 600 "$OUT/bench.log" python bench.py --steps 20 --warmup 3
grep -E '^\{' "$OUT/bench.log" > "$OUT/bench.json" || true
( cd /tmp && step 600 "$OLDPWD/$OUT/rocprof_run.log" rocprofv3 --kernel-trace --stats --output-format csv \
    -d "$OLDPWD/$OUT/rocprof" -- python3 "$OLDPWD/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-alt-math )
find "$OUT/rocprof" -name "*kernel_stats*.csv" | head -3
for f in $(find "$OUT/rocprof" -name "*kernel_stats*.csv" | head -1); do head -12 "$f"; done
echo "== done"
