#!/usr/bin/env python3
"""Interleaved A/B of rd_batch_develop configurations in ONE process (the shader clock of the box drifts by minutes,
so variants are only comparable when they alternate): per-frame launches vs multi-frame launches, frames per launch, ring depth.  The environment switches are read when a batch context is
created, so each variant gets its own context.

    python tools/bench_batch_ab.py [--format f32] [--frames 64] [--rounds 4] [--variants name,name,...]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import raweditor_amd as ra

SEED = 0x52415745
WB = (2.0, 1.0, 1.5, 1.0)
CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)

VARIANTS = {
    # name: (env, ring).  A launch never holds two frames that share a surface, so a ring of r caps the launch at r frames.
    "per_frame_r8": ({"RD_BATCH_PERSISTENT": "0"}, 8),
    "per_frame_r32": ({"RD_BATCH_PERSISTENT": "0"}, 32),
    "multi_r8": ({}, 8),                                     # the default: up to 8 frames per launch
    "multi_r32": ({}, 32),
    "multi_r32_cap1": ({"RD_BATCH_MAX_FRAMES": "1"}, 32),
    "multi_r32_cap2": ({"RD_BATCH_MAX_FRAMES": "2"}, 32),
    "multi_r32_cap4": ({"RD_BATCH_MAX_FRAMES": "4"}, 32),
    "multi_r32_cap6": ({"RD_BATCH_MAX_FRAMES": "6"}, 32),
    "multi_r32_cap12": ({"RD_BATCH_MAX_FRAMES": "12"}, 32),
    "multi_r32_cap16": ({"RD_BATCH_MAX_FRAMES": "16"}, 32),
    "multi_r32_cap32": ({"RD_BATCH_MAX_FRAMES": "32"}, 32),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", choices=["f32", "f16", "u8"], default="f32")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=4, help="passes over the batch per timing")
    ap.add_argument("--width", type=int, default=6016)
    ap.add_argument("--height", type=int, default=4016)
    ap.add_argument("--math", choices=["strict", "contracted"], default="strict")
    ap.add_argument("--variants", default="per_frame_r8,multi_r8,multi_r32_cap4,multi_r32_cap16,multi_r32_cap32")
    args = ap.parse_args()
    W, H, F = args.width, args.height, args.frames
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    fmt = {"f32": ra.FMT_RGBA_F32, "f16": ra.FMT_RGBA_F16, "u8": ra.FMT_RGBA_U8}[args.format]
    bpp = ra.BYTES_PER_PIXEL[fmt]
    cfas, params = [], []
    for f in range(F):
        g = torch.Generator(device=dev)
        g.manual_seed(SEED + f)
        cfas.append(torch.randint(0, 4096, (H, W), generator=g, device=dev, dtype=torch.int16))
        params.append(ra.EditParams.random(np.random.default_rng([SEED, f])))
    names = [n for n in args.variants.split(",") if n]
    max_ring = max(VARIANTS[n][1] for n in names)
    ring = [torch.empty(H * W * bpp, dtype=torch.uint8, device=dev) for _ in range(max_ring)]
    hist = torch.zeros(768, dtype=torch.int64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    math_mode = ra.MATH_CONTRACTED if args.math == "contracted" else ra.MATH_STRICT
    ctx = {}
    for n in names:
        env, r = VARIANTS[n]
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        be = ra.BatchExporter(0, W, H, fmt, True, math_mode=math_mode)
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        frames = be.make_frames([c.data_ptr() for c in cfas], [ring[i % r].data_ptr() for i in range(F)], params, WB, CM)
        ctx[n] = (be, frames)
    ref_hist = None
    res = {n: [] for n in names}
    with torch.cuda.stream(stream):
        for rnd in range(args.rounds + 1):
            for n in names:
                be, frames = ctx[n]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(args.reps):
                    be.develop(frames, stream=stream.cuda_stream)
                    be.histogram(hist.data_ptr(), stream=stream.cuda_stream)
                e1.record(stream)
                torch.cuda.synchronize()
                h = hist.clone()
                if ref_hist is None:
                    ref_hist = h
                assert torch.equal(h, ref_hist), f"{n}: histogram differs from the first variant's"
                if rnd:                                      # round 0 = warm-up
                    res[n].append(e0.elapsed_time(e1) * 1e3 / (args.reps * F))
    print(f"{args.format} {W}x{H}, {F} frames, {args.reps} passes per timing, {args.rounds} rounds; us per frame")
    for n in names:
        v = res[n]
        print(f"  {n:22s} mean {sum(v) / len(v):7.2f}  min {min(v):7.2f}  max {max(v):7.2f}   {VARIANTS[n][0]}")
    for be, _ in ctx.values():
        be.close()


if __name__ == "__main__":
    main()
