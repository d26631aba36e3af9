#!/bin/bash
# Round 4: the two-level threshold tables of the RGBA-f16 surface.  Exactness gate first (the exhaustive self-test of the
# device code over all 2^32 encodings + every f16 parity test), then an alternating same-box A/B against the build without
# them (tools/librawdev_r4nof16lut.so = the same sources with -DRD_F16_LUT=0) on i.i.d. noise (worst case for the tables'
# LDS bank conflicts) and on gradient data, 24 MP and the 100 MP config-5 shape.      bash tools/gpu_r4_f16lut.sh [tag] [rounds]
set -u
TAG=${1:-r4f16}; ROUNDS=${2:-3}
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
step 300 "$OUT/selftest.txt" python - <<'PY' || exit 1
import ctypes as C, sys, time
from raweditor_amd import _lib
L = _lib.lib()
m, f, p = C.c_uint64(), C.c_uint32(), C.c_uint64()
t = time.time()
rc = L.rd_selftest_f16_lut(0, C.byref(m), C.byref(f), C.byref(p))
print(f"rd_selftest_f16_lut: rc {rc}, {m.value} mismatching of 2^32 encodings (first 0x{f.value:08x}), {p.value} encodings take the pinned evaluation, {time.time() - t:.1f} s")
sys.exit(1 if rc or m.value else 0)
PY
step 900 "$OUT/gate.log" python -m pytest tests/test_gpu_q8.py tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_export.py tests/test_gpu_fullres.py -m gpu -x -q || exit 1
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["roofline"]["us_per_frame"], "us/frame  verified", d["verified"])'
for data in uniform gradient; do
    for i in $(seq "$ROUNDS"); do
        for lib in tools/librawdev_r4nof16lut.so raweditor_amd/librawdev.so; do
            RAWDEV_LIB=$lib timeout -k 10 300 python bench.py --format f16 --ring 32 --data $data --no-cpu-baseline --no-alt-math --no-extra --no-box --steps 10 2>>"$OUT/ab.err" \
                | python -c "$pick" "f16 $data $(basename $lib)" | tee -a "$OUT/ab.txt"
            rc=${PIPESTATUS[0]}; if [ $rc -ge 124 ]; then echo "bench killed: stopping"; exit $rc; fi
        done
    done
done
for lib in tools/librawdev_r4nof16lut.so raweditor_amd/librawdev.so; do
    echo "--- typical edits, $lib" | tee -a "$OUT/stacks.txt"
    RAWDEV_LIB=$lib timeout -k 10 300 python tools/bench_stacks.py f16 2>>"$OUT/ab.err" | tee -a "$OUT/stacks.txt"
done
echo "== done"
