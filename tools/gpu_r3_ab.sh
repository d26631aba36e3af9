#!/bin/bash
# Round-3 GPU pass: exactness gate (exhaustive shortcut self-tests + node batch), then an alternating same-box A/B of
# two builds of librawdev.so (tools/librawdev_r3base.so = the build before this round's VALU work) on the three surface
# formats, then the full -m gpu suite.  A step that times out or is killed ends the script.
#   bash tools/gpu_r3_ab.sh [tag] [rounds]
set -u
TAG=${1:-r3ab}; ROUNDS=${2:-2}
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
step 600 "$OUT/gate.log" python -m pytest tests/test_gpu_q8.py tests/test_gpu_node_batch.py -x -q || exit 1
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["roofline"]["us_per_frame"], "us/frame  verified", d["verified"])'
for fmt in u8 f16 f32; do
    ring=8; [ $fmt != f32 ] && ring=32
    for i in $(seq "$ROUNDS"); do
        for lib in ${LIBS:-tools/librawdev_r3base.so raweditor_amd/librawdev.so}; do
            RAWDEV_LIB=$lib timeout -k 10 300 python bench.py --format $fmt --ring $ring --no-cpu-baseline --no-alt-math --no-extra --steps 10 2>>"$OUT/ab.err" \
                | python -c "$pick" "$fmt $(basename $lib)" | tee -a "$OUT/ab.txt"
            rc=${PIPESTATUS[0]}; if [ $rc -ge 124 ]; then echo "bench killed: stopping"; exit $rc; fi
        done
    done
done
step 900 "$OUT/pytest_gpu.log" python -m pytest tests -m gpu -x -q
echo "== done"
