#!/bin/bash
# After a change to the pinned pow pair's instruction sequence (rd_math.h): the GPU's rd_gamma_clamp against the oracle for all
# 2^32 encodings, every exhaustive self-test, the whole -m gpu suite, then an alternating f32 A/B against the previous build.
#   bash tools/gpu_r4_gamma.sh <tag> <rounds> <previous lib> <new lib>
set -u
TAG=$1; ROUNDS=$2; OLD=$3; NEW=$4
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 600 ./tools/gamma_exhaustive > "$OUT/gamma_exhaustive.txt" 2>&1; rc=$?; tail -4 "$OUT/gamma_exhaustive.txt"; [ $rc -ne 0 ] && exit 1
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1; rc=$?; tail -2 "$OUT/pytest.log"; [ $rc -ne 0 ] && exit 1
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["roofline"]["us_per_frame"], "us/frame  verified", d["verified"])'
for i in $(seq "$ROUNDS"); do
    for lib in "$OLD" "$NEW"; do
        for data in uniform gradient; do
            RAWDEV_LIB=$lib timeout -k 10 300 python bench.py --data $data --no-cpu-baseline --no-alt-math --no-extra --no-box --steps 10 2>>"$OUT/ab.err" \
                | python -c "$pick" "f32 $data $(basename $lib)" | tee -a "$OUT/ab.txt"
            rc=${PIPESTATUS[0]}; if [ $rc -ge 124 ]; then echo "bench killed: stopping"; exit $rc; fi
        done
    done
done
echo "== done"
