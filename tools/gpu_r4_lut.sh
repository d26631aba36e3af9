#!/bin/bash
# Round-4, item 2: the LDS threshold-table gamma of the RGBA8 surface.  Exactness gate first (exhaustive self-tests + parity),
# then issue costs of the instructions around the table, then an alternating same-box A/B against the build without it
# (tools/librawdev_r4nolut.so = the same sources with -DRD_Q8_LUT=0) on i.i.d. noise (worst case for LDS bank conflicts)
# and on the gradient data.      bash tools/gpu_r4_lut.sh [tag] [rounds]
set -u
TAG=${1:-r4lut}; ROUNDS=${2:-3}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
step() {   # step <seconds> <logfile> <cmd...>
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s) -> $log"
    timeout -k 10 "$secs" "$@" > "$log" 2>&1
    local rc=$?
    tail -n 4 "$log"
    echo "== rc=$rc"
    if [ $rc -ge 124 ]; then echo "step killed/timed out: stopping"; exit $rc; fi
    return $rc
}
step 600 "$OUT/gate.log" python -m pytest tests/test_gpu_q8.py tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_export.py -m gpu -x -q || exit 1
step 120 "$OUT/valu_probe_new.txt" ./tools/valu_probe2 new
cat "$OUT/valu_probe_new.txt"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["roofline"]["us_per_frame"], "us/frame  verified", d["verified"])'
for data in uniform gradient; do
    for i in $(seq "$ROUNDS"); do
        for lib in tools/librawdev_r4nolut.so raweditor_amd/librawdev.so; do
            RAWDEV_LIB=$lib timeout -k 10 300 python bench.py --format u8 --ring 32 --data $data --no-cpu-baseline --no-alt-math --no-extra --no-box --steps 10 2>>"$OUT/ab.err" \
                | python -c "$pick" "u8 $data $(basename $lib)" | tee -a "$OUT/ab.txt"
            rc=${PIPESTATUS[0]}; if [ $rc -ge 124 ]; then echo "bench killed: stopping"; exit $rc; fi
        done
    done
done
for lib in tools/librawdev_r4nolut.so raweditor_amd/librawdev.so; do
    echo "--- typical edits, $lib" | tee -a "$OUT/stacks.txt"
    RAWDEV_LIB=$lib timeout -k 10 300 python tools/bench_stacks.py u8 2>>"$OUT/ab.err" | tee -a "$OUT/stacks.txt"
done
echo "== done"
