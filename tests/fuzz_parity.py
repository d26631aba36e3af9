#!/usr/bin/env python3
"""Differential fuzz of the develop path: random frames through librawdev.so (rd_render*, every surface format, fused
histogram, both kernels, both arithmetic modes) against the CPU oracle, bit for bit.  Part of tests/ (the oracle is the
checker); tests/test_gpu_parity.py::test_differential_fuzz_sample runs a sample of it, this file runs as many as asked.

The draws are built to land on the host-side case splits of rd_uniforms.h, not only in the middle of the UI ranges:
sliders left at their defaults in every combination (identity-step elision, the channel-separable path), the identity /
camera / random matrices, sliders far outside the UI ranges, tiny and huge levels sliders (the divide's three variants),
widths around the 128-pixel tile, 1- and 2-row frames, zeros and saturated samples, black levels.

    python -m tests.fuzz_parity [cases=2000] [seed=1]        (run on the GPU box; prints a summary line, exit 1 on a mismatch)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import raweditor_amd as ra
from oracle import ref_c as refc
from tests.helpers import CM_IDENTITY, CM_TEST, PARAM_NAMES, UI_RANGES

F = np.float32
DEFAULTS = {"exposure": 0.0, "contrast": 0.0, "highlights": 0.0, "shadows": 0.0, "whites": 1.0, "blacks": 0.0,
            "vibrance": 0.0, "saturation": 0.0, "temperature": 0.0, "tint": 0.0}


def draw_params(rng):
    kind = rng.integers(0, 10)
    p = {}
    keep = rng.uniform(0.1, 0.9)                                  # share of sliders left untouched in this draw
    for k in PARAM_NAMES:
        if rng.random() < keep:
            continue
        lo, hi = UI_RANGES[k]
        v = rng.uniform(lo, hi)
        if kind == 0:                                             # far outside the UI
            v *= 10.0 ** rng.integers(0, 4)
        elif kind == 1 and rng.random() < 0.5:                    # barely touched
            v = DEFAULTS[k] + (hi - lo) * 10.0 ** -rng.integers(3, 9) * rng.choice((-1.0, 1.0))
        p[k] = float(F(v))
    if kind == 2:                                                 # levels: tiny / cancelling / huge
        p["blacks"] = float(F(rng.choice((1e-20, -1e-30, 1e-15, 8.8e-16, 0.5, 0.49999997, -3.0, 0.3))))
        p["whites"] = float(F(rng.choice((1.0, 0.5001, 0.2999, 3e12, 1e12, 37.0, p["blacks"], p["blacks"] - 0.0001))))
    if kind == 3:                                                 # separable stacks: no channel-mixing slider
        for k in ("highlights", "shadows", "vibrance", "saturation"):
            p.pop(k, None)
    return p


def draw_case(rng):
    wsel = rng.integers(0, 6)
    w = int((1, 2, 6)[rng.integers(0, 3)] if wsel == 0 else 128 * rng.integers(1, 4) + (0, 0, 2, -2, 1)[rng.integers(0, 5)]
            if wsel < 3 else rng.integers(1, 400))
    h = int(rng.integers(1, 3) if rng.random() < 0.15 else rng.integers(1, 70))
    hi = 65536 if rng.random() < 0.3 else 4096
    cfa = rng.integers(0, hi, (h, w), dtype=np.uint16)
    if rng.random() < 0.5:
        cfa[rng.random((h, w)) < 0.1] = 0
        cfa[rng.random((h, w)) < 0.05] = hi - 1
    if rng.random() < 0.1:
        cfa[:] = rng.integers(0, hi)                               # flat frame
    msel = rng.integers(0, 4)
    cm = CM_IDENTITY if msel < 2 else CM_TEST if msel == 2 else tuple(float(F(x)) for x in rng.normal(0.3, 0.8, 9))
    wb = (2.0, 1.0, 1.5, 1.0) if rng.random() < 0.5 else tuple(float(F(x)) for x in rng.uniform(0.5, 3.0, 4))
    bl = 0 if rng.random() < 0.7 else int(rng.integers(1, 1200))
    return cfa, draw_params(rng), wb, cm, bl, int(rng.integers(0, 2))


def check(case, view=None):
    cfa, params, wb, cm, bl, math = case
    h, w = cfa.shape
    ep = ra.EditParams(**params)
    pipe = ra.RenderPipeline.new(1, cfa.reshape(-1), w, h, ep, wb, cm)
    if bl:
        pipe.set_black_level(bl)
    if math:
        pipe.set_math_mode(math)
    u = refc.make_uniforms(params, wb, cm, 1.0, 0.0, 0.0, bl, math)
    exp = refc.render_f32(cfa, u, None, None, nthreads=8)
    u8 = refc.pack_u8(exp)
    hist_exp = refc.histogram(u8)
    bad = []
    got, hist = pipe.render(None, None, ra.FMT_RGBA_F32, with_histogram=True)
    if not np.array_equal(got.view(np.uint32), exp.view(np.uint32)):
        bad.append("f32")
    if not np.array_equal(hist, hist_exp):
        bad.append("hist(f32)")
    got8, hist8 = pipe.render(None, None, ra.FMT_RGBA_U8, with_histogram=True)
    if not np.array_equal(got8, u8):
        bad.append("u8")
    if not np.array_equal(hist8, hist_exp):
        bad.append("hist(u8)")
    got16, hist16 = pipe.render(None, None, ra.FMT_RGBA_F16, with_histogram=True)
    if not np.array_equal(got16.view(np.uint16), refc.pack_f16(exp).view(np.uint16)):
        bad.append("f16")
    if not np.array_equal(hist16, hist_exp):
        bad.append("hist(f16)")
    rgb = pipe.render(fmt=ra.FMT_RGB_U8)
    if not np.array_equal(rgb, u8[..., :3]):
        bad.append("rgb8")
    os.environ["RD_FORCE_MAP"] = "1"                             # the general (one pixel per lane) kernel
    try:
        gm = pipe.render()
    finally:
        os.environ.pop("RD_FORCE_MAP", None)
    if not np.array_equal(gm.view(np.uint32), exp.view(np.uint32)):
        bad.append("f32(map)")
    flags = ra.elided_steps(ep, wb, cm, math)
    if view is not None:                                         # the point-sampling map: target size, zoom, pan (shaders.rs:23-60)
        tw, th, zoom, px_, py_ = view
        pipe.update_uniforms_with_zoom(ep, zoom, px_, py_)
        uz = refc.make_uniforms(params, wb, cm, zoom, px_, py_, bl, math)
        ez = refc.render_f32(cfa, uz, tw, th, nthreads=8)
        gz, hz = pipe.render(tw, th, ra.FMT_RGBA_F32, with_histogram=True)
        if not np.array_equal(gz.view(np.uint32), ez.view(np.uint32)):
            bad.append(f"f32(view {view})")
        if not np.array_equal(hz, refc.histogram(refc.pack_u8(ez))):
            bad.append(f"hist(view {view})")
        g8 = pipe.render(tw, th, ra.FMT_RGBA_U8)
        if not np.array_equal(g8, refc.pack_u8(ez)):
            bad.append(f"u8(view {view})")
    pipe.close()
    return bad, flags


def check_batch(rng):
    """One rd_batch_develop call over 1-12 heterogeneous frames (own stack, white balance, matrix and black level each), a
    random surface format, row bands, frames-per-launch cap and arithmetic mode, run twice; every surface and the
    accumulated histogram against the oracle.  Returns (mismatch descriptions, frames)."""
    from raweditor_amd._lib import RdFrame
    from tests.gpu_util import DevBuf, sync
    # batch: any width since round 6 -- even ones around the tile, multiples of the tile, and odd ones (whole quads by the export
    # kernel + rd_develop_lastcol; 1 = no quad at all)
    sel = rng.random()
    w = int(2 * rng.integers(1, 200)) if sel < 0.45 else int(128 * rng.integers(1, 9)) if sel < 0.7 else int(2 * rng.integers(0, 200) + 1)
    h = int(rng.integers(1, 60))
    n = int(rng.integers(1, 13))
    math = int(rng.integers(0, 2))
    fmt, dtype, ch, pack = [(ra.FMT_RGBA_F32, np.float32, 4, None), (ra.FMT_RGBA_U8, np.uint8, 4, refc.pack_u8),
                            (ra.FMT_RGBA_F16, np.uint16, 4, refc.pack_f16),
                            (ra.FMT_RGB_U8, np.uint8, 3, lambda e: np.ascontiguousarray(refc.pack_u8(e)[..., :3]))][rng.integers(0, 4)]
    if fmt == ra.FMT_RGB_U8 and w < 128:                          # the RGB8 batch needs one whole tile per row
        fmt, dtype, ch, pack = ra.FMT_RGBA_U8, np.uint8, 4, refc.pack_u8
    bands = int(rng.integers(1, 4))
    cap = rng.choice(("", "1", "2", "3", "8"))
    cases = []
    for _ in range(n):
        cfa, params, wb, cm, bl, _m = draw_case(rng)
        hi = 65536 if rng.random() < 0.3 else 4096
        cases.append((rng.integers(0, hi, (h, w), dtype=np.uint16), params, wb, cm, bl))
    exp, exp_hist = [], np.zeros(768, np.uint64)
    for cfa, params, wb, cm, bl in cases:
        e = refc.render_f32(cfa, refc.make_uniforms(params, wb, cm, 1.0, 0.0, 0.0, bl, math), None, None, nthreads=8)
        exp_hist += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
        exp.append(e if pack is None else pack(e))
    if cap:
        os.environ["RD_BATCH_MAX_FRAMES"] = cap
    try:
        be = ra.BatchExporter(0, w, h, fmt, True, math_mode=math)
    finally:
        os.environ.pop("RD_BATCH_MAX_FRAMES", None)
    d_in = [DevBuf.from_array(c[0]) for c in cases]
    d_out = [DevBuf(h * w * ra.BYTES_PER_PIXEL[fmt]) for _ in range(n)]
    d_hist = DevBuf(768 * 8)
    frames = (RdFrame * n)()
    for i, (cfa, params, wb, cm, bl) in enumerate(cases):
        frames[i].cfa_dev = d_in[i].ptr
        frames[i].out_dev = d_out[i].ptr
        frames[i].params = ra.EditParams(**params).to_c()
        frames[i].wb_multipliers[:] = [float(x) for x in wb]
        frames[i].color_matrix[:] = [float(x) for x in cm]
        frames[i].black_level = bl
    bad = []
    for rep in range(2):
        be.develop(frames, row_bands=bands)
        be.histogram(d_hist.ptr)
        sync()
        for i in range(n):
            got = d_out[i].to_array(dtype, (h, w, ch))
            e = exp[i]
            same = np.array_equal(got.view(np.uint32), e.view(np.uint32)) if dtype is np.float32 else \
                np.array_equal(got, e.view(dtype).reshape(got.shape))
            if not same:
                bad.append(f"frame {i} pass {rep}")
        if not np.array_equal(d_hist.to_array(np.uint64, (768,)), exp_hist):
            bad.append(f"histogram pass {rep}")
    be.close()
    if bad:
        bad.append(f"[{w}x{h}, {n} frames, fmt {fmt}, bands {bands}, cap {cap!r}, math {math}]")
    return bad, n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng([0x52415745, seed])
    t0 = time.perf_counter()
    fails, px, sep, fix, views = 0, 0, 0, 0, 0
    for i in range(n):
        case = draw_case(rng)
        view = None
        if rng.random() < 0.5:
            view = (int(rng.integers(1, 300)), int(rng.integers(1, 200)), float(F(rng.choice((1.0, rng.uniform(0.3, 8.0))))),
                    float(F(rng.uniform(-0.7, 0.7))), float(F(rng.uniform(-0.7, 0.7))))
            if rng.random() < 0.03:                                  # zoom = 0: infinite coordinates, NaN at the centre of an odd target
                view = (2 * int(rng.integers(0, 40)) + 1, 2 * int(rng.integers(0, 30)) + 1, 0.0, float(rng.choice((0.0, 0.25))), 0.0)
            elif rng.random() < 0.1:                                 # tex_coords of exactly 0.0 and 1.0: zoom 1/2, an even target, no pan
                view = (2 * int(rng.integers(1, 150)), 2 * int(rng.integers(1, 100)), 0.5, 0.0, 0.0)   # (pixel_coords = W: parity on W)
            views += 1
        bad, flags = check(case, view)
        px += case[0].size
        sep += (flags & 0x7a) == 0x7a
        fix += bool(flags & 128)
        if bad:
            fails += 1
            print(f"case {i}: MISMATCH {bad}: {case[0].shape} params={case[1]} wb={case[2]} cm={case[3]} bl={case[4]} math={case[5]}",
                  flush=True)
        if i % 250 == 249:
            print(f"{i + 1} cases, {fails} mismatching", flush=True)
    nb, bfails, bframes = max(1, n // 20), 0, 0
    for i in range(nb):
        bad, k = check_batch(rng)
        bframes += k
        if bad:
            bfails += 1
            print(f"batch call {i}: MISMATCH {bad}", flush=True)
    print(f"fuzz_parity seed {seed}: {nb} rd_batch_develop calls ({bframes} heterogeneous frames, random format / row bands / frames per "
          f"launch / arithmetic mode, two passes each; surfaces and the accumulated histogram against the oracle): {bfails} mismatching calls")
    fails_total = fails + bfails
    print(f"fuzz_parity seed {seed}: {n} random frames ({px} pixels; {sep} channel-separable stacks, {fix} with the one-correction "
          f"divide, {n - fix} on the other divide paths), 8 renders each (f32 / RGBA8 / f16 + histograms, RGB8, map kernel), {views} of them also as a zoomed / panned "
          f"view of random size (f32 + histogram, RGBA8), against the oracle: {fails} mismatching frames, {time.perf_counter() - t0:.0f} s")
    return 1 if fails_total else 0


if __name__ == "__main__":
    sys.exit(main())
