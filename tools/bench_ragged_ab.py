#!/usr/bin/env python3
"""Time per PIXEL of a ragged frame width beside an aligned one, one surface format, one library (RAWDEV_LIB picks an A/B
build): 64-frame batches at 6016x4016 and at --width x --height alternate --rounds times; best of each.

    RAWDEV_LIB=tools/librawdev_r6rgbplain.so python tools/bench_ragged_ab.py --format rgb8
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raweditor_amd as ra

SEED = 0x52415745
WB = (2.0, 1.0, 1.5, 1.0)
CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", choices=["f32", "f16", "u8", "rgb8"], default="rgb8")
    ap.add_argument("--width", type=int, default=6000)
    ap.add_argument("--height", type=int, default=4000)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    fmt = {"f32": ra.FMT_RGBA_F32, "f16": ra.FMT_RGBA_F16, "u8": ra.FMT_RGBA_U8, "rgb8": ra.FMT_RGB_U8}[a.format]
    bpp = ra.BYTES_PER_PIXEL[fmt]
    ring_n = 8 if a.format == "f32" else 16
    stream = torch.cuda.Stream(device=dev)
    sets = {}
    for tag, (W, H) in (("aligned", (6016, 4016)), ("ragged", (a.width, a.height))):
        cfas, params = [], []
        for f in range(a.frames):
            g = torch.Generator(device=dev)
            g.manual_seed(SEED + f)
            cfas.append(torch.randint(0, 4096, (H, W), generator=g, device=dev, dtype=torch.int16))
            params.append(ra.EditParams.random(np.random.default_rng([SEED, f])))
        ring = [torch.empty(H * W * bpp, dtype=torch.uint8, device=dev) for _ in range(ring_n)]
        be = ra.BatchExporter(0, W, H, fmt, True)
        frames = be.make_frames([c.data_ptr() for c in cfas], [ring[i % ring_n].data_ptr() for i in range(a.frames)], params, WB, CM)
        sets[tag] = (W, H, cfas, ring, be, frames)
    hist = torch.zeros(768, dtype=torch.int64, device=dev)
    best = {}
    for rnd in range(a.rounds):
        for tag, (W, H, cfas, ring, be, frames) in sets.items():
            with torch.cuda.stream(stream):
                be.develop(frames, stream=stream.cuda_stream)
                stream.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(a.steps):
                    be.develop(frames, stream=stream.cuda_stream)
                    be.histogram(hist.data_ptr(), stream=stream.cuda_stream)
                e1.record(stream)
                stream.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (a.steps * a.frames)
            assert int(hist.sum().item()) == 3 * a.frames * W * H
            best[tag] = min(best.get(tag, 1e9), us)
    ns_a = best["aligned"] * 1e3 / (6016 * 4016)
    ns_r = best["ragged"] * 1e3 / (a.width * a.height)
    print(f"lib={os.environ.get('RAWDEV_LIB', 'product')} format={a.format} aligned 6016x4016 {best['aligned']:.2f} us/frame  "
          f"ragged {a.width}x{a.height} {best['ragged']:.2f} us/frame  ns/px {ns_a:.5f} vs {ns_r:.5f}  ratio {ns_r / ns_a:.4f}")


if __name__ == "__main__":
    main()
