#!/bin/bash
# round 5 (last sentence): does a launch's time depend on where its eight CFA planes START?  The
# headline with one torch allocation per plane (the default), then with the planes carved out of one arena at a pitch of
# (plane rounded up to 2 MiB) + 0 / 4 KiB / 68 KiB / 1 MiB + 16 B, alternating on one box; the per-launch spread inside a run
# comes from the kernel trace of the first and the last variant.     bash tools/gpu_r5_stagger.sh [tag] [rounds]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r5stagger}; mkdir -p "$OUT"
N=${2:-2}
for i in $(seq 1 $N); do
  for st in -1 0 4096 69632 1048592; do
    name=stagger_${st}_$i
    timeout -k 10 300 python3 "$ROOT/bench.py" --no-extra --no-cpu-baseline --no-alt-math --plane-stagger $st > "$OUT/$name.json" 2> "$OUT/$name.err"
    rc=$?; [ $rc -ge 124 ] && exit $rc
    python3 -c "
import json
d=json.load(open('$OUT/$name.json')); r=d['roofline']
print('%-22s %9.1f MP/s  %7.2f us/frame  launch %.1f us  cfa mod 2MiB %s' % ('$name', d['value'], r['us_per_frame'], r['launch_us'], d['config'].get('buffers',{}).get('cfa_addr_mod_2MiB_first8')))"
  done
done
cd /tmp && export TMPDIR=/tmp
for st in -1 69632; do
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$st" -- python3 "$ROOT/bench.py" --no-extra --no-cpu-baseline --no-alt-math --no-box --plane-stagger $st > "$OUT/trace_$st.json" 2> "$OUT/trace_$st.err"
  rc=$?; [ $rc -ge 124 ] && exit $rc
  python3 - "$OUT/trace_$st" $st <<'PY'
import csv, glob, sys, statistics
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "rd_develop_batch" in r["Kernel_Name"]]
d = d[-640:]                                    # the 20 timed steps x 32 launches
pos = [statistics.mean(d[p::32]) / 1e3 for p in range(32)]
print(f"stagger {sys.argv[2]}: {len(d)} launches, mean {statistics.mean(d)/1e3:.1f} us; by position in the step: min {min(pos):.1f} (pos {pos.index(min(pos))}), max {max(pos):.1f} (pos {pos.index(max(pos))}), spread {(max(pos)-min(pos))/statistics.mean(pos)*100:.1f} %")
print("  per position:", " ".join(f"{x:.0f}" for x in pos))
PY
done
