#!/usr/bin/env python3
"""PCIe-inclusive export throughput (reported in DESIGN.md, never `value` of bench.py): 24 MP frames resident in
HBM -> fused develop to RGBA8 / RGB8 -> pinned host ring (rd_exporter_*), copy overlapped with the next kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raweditor_amd as ra
from tests.gpu_util import DevBuf

W, H = 6016, 4016
WB = (2.0, 1.0, 1.5, 1.0); CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
rng = np.random.default_rng(0x52415745)
ins = [DevBuf.from_array(rng.integers(0, 4096, (H, W), dtype=np.uint16)) for _ in range(8)]
for name, fmt in (("RGBA8", ra.FMT_RGBA_U8), ("RGB8", ra.FMT_RGB_U8), ("RGBA-f32", ra.FMT_RGBA_F32)):
    for slots in (2, 4):
        ex = ra.Exporter(0, W, H, fmt, n_slots=slots)
        frames = [ex.frame(ins[i % 8].ptr, ra.EditParams.random(np.random.default_rng([1, i])), WB, CM) for i in range(64)]
        for _ in ex.export(frames[:8]):
            pass
        t0 = time.perf_counter(); n = 0; checksum = 0
        for i, surf in ex.export(frames):
            checksum += int(surf[0, 0, 0]); n += 1
        dt = time.perf_counter() - t0
        mb = W * H * ra.BYTES_PER_PIXEL[fmt] / 1e6
        print(f"{name:9s} slots={slots}: {n / dt:7.1f} frames/s  {n * W * H / 1e6 / dt:9.0f} MP/s  "
              f"{n * mb / 1e3 / dt:6.1f} GB/s over PCIe  ({dt / n * 1e3:.2f} ms/frame)", flush=True)
        ex.close()
