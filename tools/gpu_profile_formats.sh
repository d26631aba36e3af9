#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench for the narrow surfaces and the 100 MP shape (the f32 headline run is
# tools/gpu_check.sh).  Usage (repo root, via gpurun): bash tools/gpu_profile_formats.sh [tag]
set -u
TAG=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_formats_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" "$@" > "$OUT/$name.log" 2>&1
  rc=$?; echo "$name rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi; }
run u8  --format u8  --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-extra
run f16 --format f16 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-extra
run c5  --width 11648 --height 8736 --format f16 --row-bands 8 --frames 64 --ring 4 --steps 3 --warmup 1 --no-cpu-baseline --no-alt-math --no-extra
cd "$ROOT"; python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
for name in ("u8", "f16", "c5"):
    f = glob.glob(f"{out}/{name}/**/*kernel_stats.csv", recursive=True)
    line = [l for l in open(f"{out}/{name}.log") if l.startswith("{")]
    b = json.loads(line[0]) if line else {}
    for r in csv.DictReader(open(f[0])) if f else []:
        if "rd_develop" in r["Name"]:
            fpl = b.get("roofline", {}).get("frames_per_launch", 1)
            print(f"{name}: {r['Name'][:60]} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us per launch = "
                  f"{float(r['AverageNs'])/1e3/fpl:.1f} us per frame ({fpl:g} frames per launch); bench (un-profiled, same run): "
                  f"{b['roofline']['us_per_frame']} us per frame, {b['value']} MP/s, frac {b['roofline']['frac']}")
PY
