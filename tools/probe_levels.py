#!/usr/bin/env python3
"""round 5: the two levels a headline run lands on (profiles/r05_plane_stagger.txt, r05_f32_park_ab.txt: 74.3 or 75.5 us per frame
on a fast box, 79.9 or 81.2 on a slow one).  Is the level a property of the BUFFERS a run happened to get, or of the run?
One process: allocate the batch (256 planes + ring of 8), time six regions of 10 steps on it, free everything, allocate again
-- five allocations.  If the six regions of one allocation agree and the allocations differ, it is placement.
    python tools/probe_levels.py [allocations=5] [regions=6]"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import raweditor_amd as ra

W, H, F = 6016, 4016, 256
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
be = ra.BatchExporter(0, W, H, ra.FMT_RGBA_F32, True)
hist = torch.zeros(768, dtype=torch.int64, device=dev)
# third argument: what is allocated anew each time -- "both" (default), "ring" (the planes stay) or "planes" (the ring stays)
WHAT = sys.argv[3] if len(sys.argv) > 3 else "both"
cfas = ring = None
for a in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    if cfas is None or WHAT in ("both", "planes"):
        cfas = None; gc.collect(); torch.cuda.empty_cache()
        cfas, params = bench.make_batch(torch, np, ra, dev, W, H, F, 0, 1)
    if ring is None or WHAT in ("both", "ring"):
        ring = None; gc.collect(); torch.cuda.empty_cache()
        ring = bench.alloc_ring(torch, dev, 8, H * W * 16)
    arrays = [be.make_frames([c.data_ptr() for c in cfas], [ring[i % 8].data_ptr() for i in range(F)], v, bench.WB, bench.CM)
              for v in (params, bench.swapped_halves(params))]
    us = []
    with torch.cuda.stream(stream):
        k = 0
        for region in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
            for _ in range(2 if region else 4):
                be.develop(arrays[k % 2], stream=stream.cuda_stream); k += 1
            stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                be.develop(arrays[k % 2], stream=stream.cuda_stream); k += 1
                be.histogram(hist.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            stream.synchronize()
            us.append(e0.elapsed_time(e1) * 1e3 / (10 * F))
    ring_mod = [r.data_ptr() >> 21 for r in ring]
    print(f"[{WHAT}] allocation {a}: us per frame by region: " + " ".join(f"{x:.2f}" for x in us) +
          f"   ring starts (2 MiB units, first relative 0): {[m - ring_mod[0] for m in ring_mod]}  cfa[0] at {cfas[0].data_ptr():#x}", flush=True)
    del arrays
be.close()
