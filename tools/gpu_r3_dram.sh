#!/bin/bash
# the L2's fabric requests classified by destination ("DRAM") next to the totals FETCH_SIZE /
# WRITE_SIZE derive from -- two counters per pass (six in one pass exceed what the hardware collects: rocprofv3 aborts).
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_dram; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
BENCH="--steps 2 --warmup 1 --frames 32 --no-cpu-baseline --no-alt-math --no-extra"
pass() { local name=$1; shift
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $BENCH > "$OUT/$name.log" 2>&1
  local rc=$?; echo "$name rc=$rc"; if [ $rc -ne 0 ]; then grep -m2 "error code\|exceeds" "$OUT/$name.log"; fi; if [ $rc -ge 124 ]; then exit $rc; fi; }
pass rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum
cd "$ROOT"; python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
for d in sorted(glob.glob(sys.argv[1]+'/*/')):
    agg=collections.defaultdict(list)
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rd_develop_batch' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('%-28s %.6g requests per frame (8-frame launches, %d dispatches)'%(k, sum(v)/len(v)/8, len(v)))
PY
