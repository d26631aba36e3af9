#!/usr/bin/env python3
"""RenderPipeline::new (pipeline.rs:114-363) for a 24 MP frame: what the call costs from a pageable Vec<u16> and from a
page-locked plane, and the first preview after it (the image-open latency of main.rs:993-1006).

    python tools/bench_create.py [reps=10]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import raweditor_amd as ra

WB = (2.0, 1.0, 1.5, 1.0)
CM = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    h, w = 4016, 6016
    rng = np.random.default_rng(1)
    cfa = rng.integers(0, 4096, (h, w), dtype=np.uint16)
    pin = ra.PinnedBytes(h * w * 2)
    pin.array.view(np.uint16)[:] = cfa.reshape(-1)
    ra.RenderPipeline.new(0, cfa.reshape(-1), w, h, ra.EditParams(), WB, CM).close()       # warm: library, LUT, allocator
    for name, src in (("pageable", cfa.reshape(-1)), ("page-locked", pin.array.view(np.uint16))):
        t_new, t_prev, t_close = [], [], []
        for k in range(reps):
            t0 = time.perf_counter()
            p = ra.RenderPipeline.new(k, src, w, h, ra.EditParams(exposure=0.3), WB, CM)
            t1 = time.perf_counter()
            p.render_to_bytes()
            t2 = time.perf_counter()
            p.close()
            t3 = time.perf_counter()
            t_new.append(t1 - t0); t_prev.append(t2 - t1); t_close.append(t3 - t2)
        med = lambda v: sorted(v)[len(v) // 2] * 1e3
        print(f"{name:12s} source: new {med(t_new):7.3f} ms (min {min(t_new) * 1e3:.3f}), first render_to_bytes {med(t_prev):6.3f} ms, "
              f"close {med(t_close):6.3f} ms   [{h * w * 2 / 1e6:.1f} MB CFA, {reps} reps]")
    pin.free()


if __name__ == "__main__":
    main()
