#!/bin/bash
# Diagnostic PMC passes (one counter block per pass) on a short bench run.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc2; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
pass() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --frames 32 --no-cpu-baseline --no-alt-math > "$OUT/$name.log" 2>&1
  rc=$?; echo "$name rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi; }
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL
pass sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_CMD_FIFO_FULL
pass tcp TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcc TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum
pass tcc2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
pass ta TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
cd "$ROOT"; python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc2/*/')):
    agg=collections.defaultdict(list); dur=[]
    for f in glob.glob(d+'**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rd_develop_quads' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(d+'**/*kernel_trace.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'rd_develop_quads' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(d.split('/')[-2], 'kernel us avg %.1f'%(sum(dur)/max(1,len(dur))), {k:'%.4g'%(sum(v)/len(v)) for k,v in agg.items()})
PY
