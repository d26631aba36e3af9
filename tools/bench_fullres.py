#!/usr/bin/env python3
"""render_full_res_to_bytes per call (PCIe-inclusive), by destination kind; RD_DST_ADVISE / RD_COPY_THREADS / RD_RENDER_BANDS /
RD_COPY_CHUNK_MB are read by librawdev at first use, so A/B runs are separate processes (tools/gpu_r4_fullres.sh)."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raweditor_amd as ra

W, H = 6016, 4016
WB = (2.0, 1.0, 1.5, 1.0); CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
rng = np.random.default_rng(0x52415745)
cfa = rng.integers(0, 4096, (H, W), dtype=np.uint16)
pipe = ra.RenderPipeline.new(1, cfa.reshape(-1), W, H, ra.EditParams.random(rng), WB, CM)
n = W * H * 4
pin = ra.PinnedBytes(n)
reused = np.zeros(n, np.uint8)


def run(make, iters=24):
    ms = []
    for _ in range(iters + 4):
        d = make()
        t0 = time.perf_counter()
        pipe.render_full_res_to_bytes(out=d)
        ms.append((time.perf_counter() - t0) * 1e3)
    ms = ms[4:]
    return statistics.median(ms), min(ms)


tag = " ".join(f"{k}={os.environ[k]}" for k in ("RD_DST_ADVISE", "RD_COPY_THREADS", "RD_RENDER_BANDS", "RD_COPY_CHUNK_MB", "RD_ASSUME_PAGEABLE") if k in os.environ) or "defaults"
for name, make in (("pinned", lambda: pin.array), ("pageable reused", lambda: reused), ("pageable fresh", lambda: None)):
    med, mn = run(make)
    print(f"[{tag}] {name:16s}: median {med:7.3f} ms  min {mn:7.3f} ms  {n / med / 1e6:6.1f} GB/s", flush=True)
