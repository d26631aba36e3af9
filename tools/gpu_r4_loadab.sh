#!/bin/bash
# A/B of how the tables are copied into LDS at the start of a launch (dword vs 16-byte loads): the launch-heavy shapes.
#   bash tools/gpu_r4_loadab.sh <tag> <rounds> <lib> [<lib> ...]
set -u
TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["roofline"]["us_per_frame"], "us/frame  verified", d["verified"])'
for i in $(seq "$ROUNDS"); do
    for lib in "$@"; do
        # config-5 shape as BASELINE words it: one launch per row band (8 per 100 MP frame)
        RAWDEV_LIB=$lib RD_BATCH_PERSISTENT=0 timeout -k 10 300 python bench.py --format f16 --width 11648 --height 8736 --frames 16 --ring 4 --row-bands 8 --no-cpu-baseline --no-alt-math --no-extra --no-box --steps 6 2>>"$OUT/ab.err" \
            | python -c "$pick" "f16 100MP 8 bands/frame $(basename $lib)" | tee -a "$OUT/ab.txt"
        # one launch per 24 MP frame, f16 and RGBA8
        for fmt in f16 u8; do
            RAWDEV_LIB=$lib RD_BATCH_PERSISTENT=0 timeout -k 10 300 python bench.py --format $fmt --ring 32 --no-cpu-baseline --no-alt-math --no-extra --no-box --steps 6 2>>"$OUT/ab.err" \
                | python -c "$pick" "$fmt 24MP one launch per frame $(basename $lib)" | tee -a "$OUT/ab.txt"
        done
    done
done
echo "== done"
