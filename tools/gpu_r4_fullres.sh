#!/bin/bash
# round 4: the box's streaming ceilings by probe-kernel shape, THP mode, and render_full_res_to_bytes A/B runs
set -u
OUT=gpurun_out/${1:-r4b}
mkdir -p "$OUT"
echo "THP enabled: $(cat /sys/kernel/mm/transparent_hugepage/enabled 2>&1)  defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag 2>&1)" | tee "$OUT/thp.txt"
uname -r | tee -a "$OUT/thp.txt"
timeout -k 10 300 ./tools/hbm_probe > "$OUT/hbm_probe.txt" 2>&1 || { echo "hbm_probe failed"; tail -5 "$OUT/hbm_probe.txt"; exit 1; }
sort -t: -k2 "$OUT/hbm_probe.txt" | grep copy | sort -k11 -n -r | head -6
grep fill "$OUT/hbm_probe.txt" | sort -k11 -n -r | head -6
grep -E "read|hipMem" "$OUT/hbm_probe.txt" | sort -k11 -n -r | head -6
: > "$OUT/fullres.txt"
for env in "X=1" "RD_DST_ADVISE=none" "RD_DST_ADVISE=populate" "RD_COPY_THREADS=0" "RD_COPY_THREADS=2" "RD_COPY_THREADS=8" "RD_RENDER_BANDS=1" "RD_COPY_CHUNK_MB=4" "RD_COPY_CHUNK_MB=96" "RD_ASSUME_PAGEABLE=1"; do
    env $env timeout -k 10 120 python tools/bench_fullres.py >> "$OUT/fullres.txt" 2>&1 || { echo "bench_fullres failed ($env)"; tail -5 "$OUT/fullres.txt"; exit 1; }
done
cat "$OUT/fullres.txt" | grep -v amdgpu.ids
