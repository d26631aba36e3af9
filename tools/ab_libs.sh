#!/bin/bash
# Same-box A/B of two builds of librawdev.so: alternates bench.py between $1 (e.g. a build of the previous commit) and the
# in-tree library, N rounds (default 3); prints us per frame of the headline step.  The box's shader clock drifts, so only
# alternating runs on ONE box compare.       tools/ab_libs.sh build/librawdev_old.so [rounds] [bench.py arguments...]
old=$1; rounds=${2:-3}; shift; shift
pick='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["roofline"]["us_per_frame"], "us/frame  verified", d["verified"])'
for i in $(seq "$rounds"); do
    RAWDEV_LIB=$old python bench.py --no-cpu-baseline --no-alt-math "$@" 2>/dev/null | python -c "$pick" old || exit 1
    python bench.py --no-cpu-baseline --no-alt-math "$@" 2>/dev/null | python -c "$pick" new || exit 1
done
