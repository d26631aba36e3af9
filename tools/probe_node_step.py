#!/usr/bin/env python3
"""Host-side wall time of every step of the one-process host (rd_node_batch_develop + rd_node_batch_histogram, which
synchronises), 256 x 24 MP.  Round 4 used it to find why `bench.py --host node` lost 8 % against the ranks host: one step in
~13 took 40 ms instead of 20 -- a generation-2 pass of Python's cyclic GC (20-30 ms in a process with torch imported) that
happened to fall into the timed loop; with one process per GPU the same pause hides behind queued work.  bench.py now
collects before and disables the GC inside its timed regions (quiet_gc).  The PROBE= variants rule out library calls."""
import ctypes as C, os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raweditor_amd as ra
from raweditor_amd import _lib
import bench
W, H, F = 6016, 4016, 256
dev = torch.device("cuda", 0)
probe = os.environ.get("PROBE", "none")
cfas, params = bench.make_batch(torch, np, ra, dev, W, H, F, 0, 1)
ring = [torch.empty(H * W * 16, dtype=torch.uint8, device=dev) for _ in range(8)]
torch.cuda.synchronize()
nb = ra.NodeBatch([0], W, H, ra.FMT_RGBA_F32, True)
if probe == "ident":
    a, b = C.create_string_buffer(64), C.create_string_buffer(128)
    _lib.lib().rd_device_identity(0, a, 64, b, 128)
if probe == "ident_nobuf":
    _lib.lib().rd_device_identity(0, None, 0, None, 0)
if probe == "count":
    n = C.c_int()
    _lib.lib().rd_device_count(C.byref(n))
if probe == "create":
    ra.BatchExporter(0, W, H, ra.FMT_RGBA_F32, True).close()
arrs = [ra.BatchExporter.make_frames([c.data_ptr() for c in cfas], [ring[i % 8].data_ptr() for i in range(F)], v, bench.WB, bench.CM)
        for v in (params, bench.swapped_halves(params))]
tt = []
for k in range(24):
    t0 = time.perf_counter()
    nb.develop(arrs[k % 2])
    nb.histogram()
    tt.append((time.perf_counter() - t0) * 1e3)
print(f"PROBE={probe:12s}: median {statistics.median(tt[2:]):7.3f} ms; steps over 21 ms: {[(i, round(x, 1)) for i, x in enumerate(tt) if x > 21]}", flush=True)
