#!/usr/bin/env python3
"""BASELINE config 2: ONE 24 MP frame, f32 surface, default and randomised stacks, >= 50 warm iterations, median.
Each iteration = rd_render_device (one fused launch [+ histogram fold]) + device synchronise, host-timed, so
it includes launch + sync latency (what an interactive caller sees); the batch number of bench.py does not."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raweditor_amd as ra
from tests.gpu_util import DevBuf, sync

W, H = 6016, 4016
WB = (2.0, 1.0, 1.5, 1.0); CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
rng = np.random.default_rng(0x52415745)
cfa = rng.integers(0, 4096, (H, W), dtype=np.uint16)
out = DevBuf(W * H * 16); hist = DevBuf(768 * 4)
for label, params in (("default stack", ra.EditParams()), ("randomised stack", ra.EditParams.random(rng))):
    pipe = ra.RenderPipeline.new(1, cfa.reshape(-1), W, H, params, WB, CM)
    for with_hist in (False, True):
        for fmt_name, fmt in (("f32", ra.FMT_RGBA_F32), ("u8", ra.FMT_RGBA_U8)):
            ts = []
            for it in range(60):
                t0 = time.perf_counter()
                pipe.render_device(W, H, fmt, out.ptr, hist.ptr if with_hist else 0)
                sync()
                ts.append((time.perf_counter() - t0) * 1e6)
            ts = ts[10:]
            med = statistics.median(ts)
            print(f"{label:17s} {fmt_name:3s} hist={int(with_hist)}: median {med:7.1f} us  min {min(ts):7.1f} us  "
                  f"-> {W * H / med:9.0f} MP/s", flush=True)
    t0 = time.perf_counter(); n = 20
    for _ in range(n):
        pipe.render_to_bytes(); pipe.render_to_histogram_bytes()
    dt = (time.perf_counter() - t0) / n
    print(f"{label:17s} interactive frame (1280x854 preview + 128x85 histogram render, both read back): {dt * 1e3:.2f} ms "
          f"-> {1 / dt:.0f} fps", flush=True)
