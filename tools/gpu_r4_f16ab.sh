#!/bin/bash
# Quick alternating A/B of the RGBA-f16 batch (256 x 24 MP) between library builds on one box:
#   bash tools/gpu_r4_f16ab.sh <tag> <rounds> <lib> [<lib> ...]
set -u
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["roofline"]["us_per_frame"], "us/frame  verified", d["verified"])'
for data in uniform gradient; do
    for i in $(seq "$ROUNDS"); do
        for lib in "$@"; do
            RAWDEV_LIB=$lib timeout -k 10 300 python bench.py --format f16 --ring 32 --data $data --no-cpu-baseline --no-alt-math --no-extra --no-box --steps 10 2>>"$OUT/ab.err" \
                | python -c "$pick" "f16 $data $(basename $lib)" | tee -a "$OUT/ab.txt"
            rc=${PIPESTATUS[0]}; if [ $rc -ge 124 ]; then echo "bench killed: stopping"; exit $rc; fi
        done
    done
done
for i in $(seq "$ROUNDS"); do                                  # the 100 MP config-5 shape, 16 frames, multi-frame launches
    for lib in "$@"; do
        RAWDEV_LIB=$lib timeout -k 10 300 python bench.py --format f16 --width 11648 --height 8736 --frames 16 --ring 4 --no-cpu-baseline --no-alt-math --no-extra --no-box --steps 10 2>>"$OUT/ab.err" \
            | python -c "$pick" "f16 100MP uniform $(basename $lib)" | tee -a "$OUT/ab.txt"
        rc=${PIPESTATUS[0]}; if [ $rc -ge 124 ]; then echo "bench killed: stopping"; exit $rc; fi
    done
done
for lib in "$@"; do
    echo "--- typical edits, $lib" | tee -a "$OUT/stacks.txt"
    RAWDEV_LIB=$lib timeout -k 10 300 python tools/bench_stacks.py f16 2>>"$OUT/ab.err" | tee -a "$OUT/stacks.txt"
done
echo "== done"
