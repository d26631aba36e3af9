#!/bin/bash
# round 4: how long do the eight row-band launches of one render_full_res_to_bytes call take, one by one?  (the kernel stats of the
# bench run showed an AVERAGE of 138 us per band launch against a minimum of 6.6 us)
set -u
OUT=$PWD/gpurun_out/r4bands; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $REPO/tools/bench_fullres.py > $OUT/run.log 2>&1
cd $REPO; grep -v amdgpu.ids $OUT/run.log | tail -3
python3 - <<'PY'
import csv, glob, statistics
f=glob.glob('/root/repo/gpurun_out/r4bands/trace/*/*kernel_trace.csv')[0]
ks=[(r['Kernel_Name'][:30], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f))]
ks=sorted(k for k in ks if 'rd_develop_quads' in k[0])
# group into calls of 8 consecutive launches (gap to the previous launch > 300 us starts a new call)
calls=[]; cur=[]
for k in ks:
    if cur and k[1]-cur[-1][2] > 300_000: calls.append(cur); cur=[]
    cur.append(k)
if cur: calls.append(cur)
calls=[c for c in calls if len(c)==8]
print(len(calls), "calls of 8 band launches")
for name, sel in (("pinned dst (calls 4..27)", calls[4:28]), ("pageable reused (32..55)", calls[32:56]), ("pageable fresh (60..83)", calls[60:84])):
    if not sel: continue
    per=[[ (c[i][2]-c[i][1])/1e3 for c in sel] for i in range(8)]
    gaps=[[ (c[i+1][1]-c[i][2])/1e3 for c in sel] for i in range(7)]
    print(name, "band durations us (median per band):", [round(statistics.median(p),1) for p in per], " gaps:", [round(statistics.median(g),1) for g in gaps], " all 8 bands span:", round(statistics.median([(c[7][2]-c[0][1])/1e3 for c in sel]),1))
mf=glob.glob('/root/repo/gpurun_out/r4bands/trace/*/*memory_copy_trace.csv')
if mf:
    ms=[r for r in csv.DictReader(open(mf[0]))]
    print("memory copies traced:", len(ms), list(ms[0].keys())[:8] if ms else None)
PY
