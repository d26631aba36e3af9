#!/bin/bash
# Round-3 diagnostics on one box: what bounds the narrow-surface kernels now?  (a) sensitivity to the VALU count
# (strict vs contracted arithmetic), (b) the fused histogram's share, (c) SQ / LDS counters of the u8 and f16 kernels.
set -u
TAG=${1:-r3exp}; OUT=$(pwd)/gpurun_out/$TAG; mkdir -p "$OUT"; ROOT=$(pwd); export TMPDIR=/tmp
pick='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["roofline"]["us_per_frame"], "us/frame  verified", d["verified"])'
for fmt in u8 f16; do
  for variant in "" "--no-hist" "--math contracted" "--math contracted --no-hist"; do
    timeout -k 10 300 python bench.py --format $fmt --ring 32 --no-cpu-baseline --no-alt-math --no-extra --steps 10 $variant 2>>"$OUT/err.txt" \
        | python -c "$pick" "$fmt [$variant]" | tee -a "$OUT/exp.txt"
    rc=${PIPESTATUS[0]}; if [ $rc -ge 124 ]; then echo "killed: stopping"; exit $rc; fi
  done
done
cd /tmp
BENCH="--steps 2 --warmup 1 --frames 32 --no-cpu-baseline --no-alt-math --no-extra"
pass() { local name=$1 ctrs=$2; shift 2
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc ${ctrs//,/ } --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?; echo "$name rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi; }
SQ1=SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_ACTIVE_INST_ANY,GRBM_GUI_ACTIVE
SQ2=SQ_WAVE_CYCLES,SQ_ACTIVE_INST_LDS,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_LDS,SQ_INSTS_SALU,GRBM_GUI_ACTIVE
SQ3=SQ_INSTS_VALU_TRANS_F32,SQ_INSTS_VALU_CVT,SQ_INSTS_VALU_FMA_F32,SQ_INSTS_VALU_MUL_F32,SQ_INSTS_VALU_ADD_F32,SQ_INSTS_VALU_INT32,SQ_INST_CYCLES_SALU,GRBM_GUI_ACTIVE
pass u8_SQ1 $SQ1 $BENCH --format u8
pass u8_SQ2 $SQ2 $BENCH --format u8
pass u8_SQ3 $SQ3 $BENCH --format u8
pass f16_SQ1 $SQ1 $BENCH --format f16
cd "$ROOT"; python3 - "$OUT" <<'PY'
import csv,glob,collections,sys
out=sys.argv[1]
for d in sorted(glob.glob(out+'/*/')):
    agg=collections.defaultdict(list); dur=[]
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rd_develop_batch' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(d+'**/*kernel_trace.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rd_develop_batch' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    n=max(1,len(dur))
    print(d.rstrip('/').split('/')[-1], 'launches %d, kernel us per 8-frame launch avg %.1f (%.2f per frame)'%(len(dur), sum(dur)/n, sum(dur)/n/8))
    for k,v in sorted(agg.items()): print('    %-28s %.5g per frame'%(k, sum(v)/len(v)/8))
PY
