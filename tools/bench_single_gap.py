#!/usr/bin/env python3
"""BASELINE configs[1], the enqueue anatomy: ONE 24 MP frame, f32 surface + fused histogram, rd_render_device + stream
synchronise per iteration; host latency, and the HIP-event time of the enqueue on its stream (the figure bench.py's
extra_configs.single_frame_f32 reports as enqueue_us).  RD_GRAPH=1 (read by librawdev at first use) sends develop + fold
as one two-node graph: run once per setting (tools/gpu_r4_single.sh), and under `rocprofv3 --kernel-trace --hip-trace` for
the timestamps of the gaps."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raweditor_amd as ra

W, H = 6016, 4016
WB = (2.0, 1.0, 1.5, 1.0); CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
ITERS = int(os.environ.get("ITERS", "60"))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0x52415745)
cfa = torch.randint(0, 4096, (H, W), generator=g, device=dev, dtype=torch.int16)
out = torch.empty(H * W * 16, dtype=torch.uint8, device=dev)
hist = torch.zeros(768, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream(device=dev)
tag = "RD_GRAPH=" + os.environ.get("RD_GRAPH", "0")
for label, p in (("randomised", ra.EditParams.random(np.random.default_rng([0x52415745, 0]))), ("default", ra.EditParams())):
    pipe = ra.RenderPipeline.from_device(1, cfa.data_ptr(), W, H, p, WB, CM, device=0)
    for with_hist in (True, False):
        host, ev = [], []
        with torch.cuda.stream(stream):
            for it in range(ITERS):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record(stream)
                pipe.render_device(W, H, ra.FMT_RGBA_F32, out.data_ptr(), hist.data_ptr() if with_hist else 0, stream.cuda_stream)
                e1.record(stream)
                stream.synchronize()
                host.append((time.perf_counter() - t0) * 1e6)
                ev.append(e0.elapsed_time(e1) * 1e3)
        host, ev = host[10:], ev[10:]
        ok = (not with_hist) or int(hist.sum().item()) == 3 * W * H
        # the same render queued K deep (no synchronise in between): the stream never idles, so (total / K) is what one render
        # costs the DEVICE -- develop + fold + whatever separates launches -- without the host's launch path in the window
        K = 16
        q = []
        with torch.cuda.stream(stream):
            for rep in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                pipe.render_device(W, H, ra.FMT_RGBA_F32, out.data_ptr(), hist.data_ptr() if with_hist else 0, stream.cuda_stream)
                e0.record(stream)
                for _ in range(K):
                    pipe.render_device(W, H, ra.FMT_RGBA_F32, out.data_ptr(), hist.data_ptr() if with_hist else 0, stream.cuda_stream)
                e1.record(stream)
                stream.synchronize()
                q.append(e0.elapsed_time(e1) * 1e3 / K)
        ev_s = sorted(ev)
        print(f"[{tag}] {label:10s} hist={int(with_hist)}: queued {K} deep: {statistics.median(q[1:]):6.1f} us per render on the device;  "
              f"one at a time, event times sorted: p10 {ev_s[len(ev_s) // 10]:.1f}  p50 {ev_s[len(ev_s) // 2]:.1f}  p90 {ev_s[len(ev_s) * 9 // 10]:.1f}", flush=True)
        print(f"[{tag}] {label:10s} hist={int(with_hist)}: host median {statistics.median(host):7.1f} us (min {min(host):6.1f})   "
              f"enqueue (HIP events) median {statistics.median(ev):7.1f} us (min {min(ev):6.1f})   hist ok: {ok}", flush=True)
    pipe.close()
