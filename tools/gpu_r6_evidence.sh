#!/bin/bash
# Round-6 evidence pass on one GPU box.   bash tools/gpu_r6_evidence.sh [tag] [part]
#   part a: smoke, the -m gpu suite, the driver's bench command (with the self-diagnosis), the other hosts, rocprofv3 kernel
#           stats of the same command (headline only, no diagnosis: the diagnostic instances would sit in the same table)
#   part b: differential fuzz (odd widths in the batch cases since this round) + soak of the final build (aligned, ragged, odd)
#   part c: rocprofv3 PMC passes (tools/gpu_pmc.sh)
#   part d: the visibility rehearsal (tools/gpu_r6_rehearsal.sh)
set -u
TAG=${1:-r06}; PART=${2:-a}
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
if [ "$PART" = a ]; then
    step 300 "$OUT/smoke.log" python __graft_entry__.py --smoke || exit 1
    step 900 "$OUT/pytest_gpu.log" python -m pytest tests -m gpu -x -q || exit 1
    step 600 "$OUT/bench.log" python bench.py --steps 20 --warmup 5
    grep -E '^\{' "$OUT/bench.log" > "$OUT/bench.json" || true
    step 300 "$OUT/bench_node1.log" python bench.py --host node --gpus 1 --no-cpu-baseline
    RD_NODE_REDUCE=host step 300 "$OUT/bench_node2_rehearsal.log" python bench.py --host node --gpus 2 --frames 64 --steps 5 --no-cpu-baseline
    RAWDEV_DIST_BACKEND=gloo step 300 "$OUT/bench_ranks2_rehearsal.log" python bench.py --gpus 2 --frames 64 --steps 5 --no-cpu-baseline
    step 300 "$OUT/bench_u8_gradient.log" python bench.py --format u8 --ring 32 --data gradient --no-cpu-baseline --no-alt-math --no-extra --steps 10
    # kernel stats of the headline command without the extras and without the diagnosis (whose probe / stamped instances are
    # other kernels, and whose event pairs stretch the launches they bracket)
    ( cd /tmp && step 600 "$OLDPWD/$OUT/rocprof_headline_run.log" rocprofv3 --kernel-trace --stats --output-format csv \
        -d "$OLDPWD/$OUT/rocprof_headline" -- python3 "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-diagnose )
    for f in $(find "$OUT/rocprof_headline" -name "*kernel_stats*.csv" | head -1); do head -4 "$f" | cut -c1-200; done
    # and with everything, for the extras' kernels and the diagnostic instances
    ( cd /tmp && step 600 "$OLDPWD/$OUT/rocprof_run.log" rocprofv3 --kernel-trace --stats --output-format csv \
        -d "$OLDPWD/$OUT/rocprof" -- python3 "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-cpu-baseline )
    for f in $(find "$OUT/rocprof" -name "*kernel_stats*.csv" | head -1); do head -16 "$f" | cut -c1-200; done
elif [ "$PART" = b ]; then
    step 1000 "$OUT/fuzz.log" python -m tests.fuzz_parity ${3:-100000} ${4:-61}
    step 600 "$OUT/soak.log" python tools/soak.py 10000
    step 600 "$OUT/soak_ragged.log" python tools/soak.py 5000 6000 4000
    step 600 "$OUT/soak_odd.log" python tools/soak.py 3000 6001 4001
elif [ "$PART" = c ]; then
    bash tools/gpu_pmc.sh "$TAG"
else
    bash tools/gpu_r6_rehearsal.sh "$TAG"
fi
echo "== done"
