#!/bin/bash
# the f32 parking A/B (tools/gpu_r5_f32ab.sh) only where it can show: on a box whose memory system leaves the VALU co-critical
# (strict vs contracted differ there).  Probes the box with one headline run; below 77 us per frame it runs the A/B, else exits.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r5fast}; mkdir -p "$OUT"
timeout -k 10 300 python3 "$ROOT/bench.py" --no-extra --no-cpu-baseline > "$OUT/probe.json" 2> "$OUT/probe.err" || exit 1
US=$(python3 -c "import json; d=json.load(open('$OUT/probe.json')); print(d['roofline']['us_per_frame'], d['alt_math']['us_per_frame'])")
echo "probe: strict / contracted us per frame = $US"
python3 -c "import sys; sys.exit(0 if float('$US'.split()[0]) < 77.0 else 3)" || { echo "not a fast box: no A/B"; exit 0; }
bash "$ROOT/tools/gpu_r5_f32ab.sh" ${1:-r5fast} 4 2>&1 | grep f32_
