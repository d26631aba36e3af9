#!/usr/bin/env python3
"""What does the fused histogram cost a single-frame f32 render, stack by stack?  (round 4: the first stack of
tools/bench_single_gap.py showed +20 us with the histogram, the default stack +1 us.)  Renders queued 16 deep."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raweditor_amd as ra

W, H = 6016, 4016
WB = (2.0, 1.0, 1.5, 1.0); CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0x52415745)
cfa = torch.randint(0, 4096, (H, W), generator=g, device=dev, dtype=torch.int16)
fmts = {"f32": (ra.FMT_RGBA_F32, 16), "u8": (ra.FMT_RGBA_U8, 4)}
out = torch.empty(H * W * 16, dtype=torch.uint8, device=dev)
hist = torch.zeros(768, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream(device=dev)


def queued(pipe, fmt, with_hist, K=16):
    q = []
    with torch.cuda.stream(stream):
        for rep in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            pipe.render_device(W, H, fmt, out.data_ptr(), hist.data_ptr() if with_hist else 0, stream.cuda_stream)
            e0.record(stream)
            for _ in range(K):
                pipe.render_device(W, H, fmt, out.data_ptr(), hist.data_ptr() if with_hist else 0, stream.cuda_stream)
            e1.record(stream)
            stream.synchronize()
            q.append(e0.elapsed_time(e1) * 1e3 / K)
    return statistics.median(q[1:])


stacks = [("default", ra.EditParams())] + [(f"seed {k}", ra.EditParams.random(np.random.default_rng([0x52415745, k]))) for k in range(12)]
for fname in ("f32", "u8"):
    fmt, bpp = fmts[fname]
    for label, p in stacks:
        pipe = ra.RenderPipeline.from_device(1, cfa.data_ptr(), W, H, p, WB, CM, device=0)
        a = queued(pipe, fmt, False); b = queued(pipe, fmt, True); a2 = queued(pipe, fmt, False); b2 = queued(pipe, fmt, True)
        h = hist.cpu().numpy().reshape(3, 256).astype(np.float64)
        ends = (h[:, 0] + h[:, 255]).sum() / h.sum()
        top = np.sort(h.reshape(-1))[::-1][:3].sum() / h.sum()
        print(f"{fname} {label:8s} exposure {p.exposure:+5.2f} contrast {p.contrast:+6.2f} blacks {p.blacks:.3f}: no hist {a:6.1f} / {a2:6.1f} us   hist {b:6.1f} / {b2:6.1f} us   "
              f"(+{b2 - a2:5.1f})   codes 0|255: {ends * 100:5.1f} %   3 fullest bins: {top * 100:5.1f} %", flush=True)
        pipe.close()
