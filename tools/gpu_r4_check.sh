#!/bin/bash
# Round-4 GPU-box pass: smoke -> the new full-resolution / bench tests (verbose) -> the whole -m gpu suite -> bench.py.
# A step that times out or is killed (rc >= 124) ends the script: no further GPU step is started.
set -u
TAG=${1:-r4a}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() {   # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s) -> $log"
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    local rc=$?
    tail -n 12 "$log"
    echo "== rc=$rc"
    if [ $rc -ge 124 ]; then echo "step killed/timed out: stopping"; exit $rc; fi
    return $rc
}
nproc
step 300 "$OUT/smoke.log" python __graft_entry__.py --smoke || exit 1
step 600 "$OUT/pytest_fullres.log" python -m pytest tests/test_gpu_fullres.py -m gpu -x -q -s
step 900 "$OUT/pytest_gpu.log" python -m pytest tests -m gpu -x -q
step 600 "$OUT/bench.log" python bench.py --steps 20 --warmup 3
grep -E '^\{' "$OUT/bench.log" > "$OUT/bench.json" || true
echo "== done"
