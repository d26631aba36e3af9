#!/usr/bin/env python3
"""Differential fuzz: the reference's shader text, executed (oracle/wgsl_eval.py), against the C oracle -- build container only.

Random small frames (1 x 1 ... 9 x 12, 12- and 16-bit CFA values, flat and saturated fields among them), random targets, random
zoom / pan (views that reach the border, tex_coords of exactly 0.0 and 1.0 included), slider stacks of mild edits, inside the UI ranges,
far outside them and at degenerate points (whites == blacks, contrast = -100, saturation = -100, exposure +-20), random matrices
and white balance.  Every frame is evaluated from the text with both pow flavours and compared BIT FOR BIT with
oracle/develop_ref.c in the matching pow mode; any difference is printed with its inputs and the run fails.

    python tools/wgsl_fuzz.py --frames 2000 [--seed N] > profiles/r06_wgsl_fuzz.txt
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_c, wgsl_eval as we, wgsl_render as wr  # noqa: E402
from tests.helpers import MILD_SPAN, PARAM_NAMES, UI_RANGES  # noqa: E402
from tools.make_wgsl_golden import LOWERINGS, shader_source  # noqa: E402

POW_MODE = {"f32_pinned": ref_c.POW_PINNED, "f32_libm": ref_c.POW_LIBM}


def f32(x):
    return float(np.float32(x))


def draw_params(rng):
    kind = 5 if rng.random() < 0.4 else rng.integers(0, 5)
    p = {}
    for k in PARAM_NAMES:
        lo, hi = UI_RANGES[k]
        if kind == 5:                                    # mild edits: most pixels stay strictly inside (0, 1), where rounding shows
            v = (1.0 if k == "whites" else 0.0) + rng.uniform(-1.0, 1.0) * MILD_SPAN.get(k, 0.3)
        elif kind == 0:                                  # inside the UI
            v = rng.uniform(lo, hi)
        elif kind == 1:                                  # far outside it
            v = rng.uniform(lo - 5 * (hi - lo), hi + 5 * (hi - lo))
        elif kind == 2:                                  # mostly neutral, a few sliders on
            v = rng.uniform(lo, hi) if rng.random() < 0.3 else (1.0 if k == "whites" else 0.0)
        elif kind == 3:                                  # range ends
            v = (lo, hi)[int(rng.integers(0, 2))]
        else:                                            # degenerate points
            v = {"exposure": rng.choice([-20.0, 20.0, 0.0]), "contrast": rng.choice([-100.0, 100.0, 0.0]),
                 "saturation": rng.choice([-100.0, 100.0]), "vibrance": rng.choice([-3.0, 3.0, 1.0]),
                 "highlights": rng.choice([-4.0, 4.0]), "shadows": rng.choice([-4.0, 4.0])}.get(k, rng.uniform(lo, hi))
        p[k] = f32(v)
    if kind == 4 and rng.random() < 0.5:
        p["blacks"] = p["whites"]                        # the levels denominator is the 0.0001 alone
    return p


def draw_frame(rng):
    h, w = int(rng.integers(1, 10)), int(rng.integers(1, 13))
    style = rng.integers(0, 5)
    if style == 0:
        cfa = rng.integers(0, 4096, (h, w))
    elif style == 1:
        cfa = rng.integers(0, 65536, (h, w))
    elif style == 2:
        cfa = np.full((h, w), rng.choice([0, 1, 4095, 4096, 65535]))
    elif style == 3:
        cfa = rng.choice([0, 4095, 65535], (h, w))
    else:
        cfa = rng.integers(0, 64, (h, w))
    return cfa.astype(np.uint16)


def draw_view(rng, h, w):
    r = rng.random()
    if r < 0.4:
        return w, h, 1.0, (0.0, 0.0)
    tw, th = int(rng.integers(1, 9)), int(rng.integers(1, 7))
    if r < 0.5:
        return tw, th, 1.0, (0.0, 0.0)
    if r < 0.6:                                          # tex_coords hit 0.0 and 1.0 exactly: zoom 1/2, even target
        return 2 * int(rng.integers(1, 4)), 2 * int(rng.integers(1, 3)), 0.5, (0.0, 0.0)
    if r < 0.64:                                         # zoom = 0: infinite coordinates (black) and, at the centre of an odd target, NaN
        return 2 * int(rng.integers(0, 4)) + 1, 2 * int(rng.integers(0, 3)) + 1, 0.0, (f32(rng.choice([0.0, 0.25])), 0.0)
    zoom = f32(rng.choice([0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 3.0, 8.0]) if rng.random() < 0.5 else rng.uniform(0.2, 6.0))
    pan = (f32(rng.uniform(-0.6, 0.6)), f32(rng.uniform(-0.6, 0.6))) if rng.random() < 0.7 else (0.5, -0.5)
    return tw, th, zoom, pan


def vector_run(src, frames, seed):
    """Larger frames (up to 96 x 128, views up to 160 x 120) through the array-at-a-time evaluator (oracle/wgsl_vec.py), pinned pow
    flavour only (numpy's binary64 power is not the C library's, so the other flavour stays with the scalar evaluator)."""
    rng = np.random.default_rng(seed)
    t0 = time.time()
    px = oob = nans = interior = 0
    for n in range(frames):
        h, w = int(rng.integers(1, 97)), int(rng.integers(1, 129))
        cfa = draw_frame(rng)
        cfa = np.resize(cfa, (h, w)) if rng.random() < 0.2 else rng.integers(0, int(rng.choice([64, 4096, 65536])), (h, w)).astype(np.uint16)
        tw, th, zoom, pan = draw_view(rng, h, w)
        if (tw, th) != (w, h) and rng.random() < 0.5:
            tw, th = int(rng.integers(1, 161)), int(rng.integers(1, 121))
        params = draw_params(rng)
        wb = [f32(x) for x in rng.uniform(0.5, 3.0, 4)]
        cm = [f32(x) for x in (rng.uniform(-1.0, 2.0, 9) if rng.random() < 0.7 else np.eye(3).reshape(-1))]
        block = wr.uniform_block(params, wb, cm, zoom, pan[0], pan[1])
        got, mod = wr.render_rows(src, cfa, block, we.Lowering(pow=wr.pow_pinned_lanes), 0, th, tw, th)
        o = ref_c.render_f32(cfa, ref_c.make_uniforms(params, wb, cm, zoom, pan[0], pan[1]), tw, th)
        if not np.array_equal(o.view(np.uint32), got.view(np.uint32)):
            bad = np.argwhere((o.view(np.uint32) != got.view(np.uint32)).any(axis=2))
            print(f"MISMATCH frame {n}: {len(bad)} pixel(s), first at (row, col) = {tuple(bad[0])}; frame {w}x{h}, target {tw}x{th}, zoom {zoom}, "
                  f"pan {pan}\n  params {params}\n  wb {wb} cm {cm}\n  oracle {o[tuple(bad[0])].tolist()} evaluated {got[tuple(bad[0])].tolist()}")
            raise SystemExit(1)
        px += tw * th
        oob += mod.texture.oob_loads
        nans += mod.nan_to_int
        interior += int(((got[..., :3] > 0) & (got[..., :3] < 1)).any(axis=2).sum())
        if (n + 1) % 2000 == 0:
            print(f"# {n + 1} frames, {px} pixels, {time.time() - t0:.0f} s", flush=True)
    print(f"wgsl_fuzz --vector: seed {seed:#x}: {frames} frames, {px} output pixels, evaluated shader text (array-at-a-time evaluator, pinned pow) == "
          f"C oracle bit for bit on all of them; {oob} out-of-bounds centre loads, {nans} NaN coordinates converted, {interior} pixels with a "
          f"channel strictly inside (0, 1); {time.time() - t0:.0f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=0x57475346)
    ap.add_argument("--vector", action="store_true", help="larger frames through oracle/wgsl_vec.py (pinned pow only)")
    a = ap.parse_args()
    src = shader_source()
    if a.vector:
        return vector_run(src, a.frames, a.seed)
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    px = oob = nan_px = border_px = interior = 0
    for n in range(a.frames):
        cfa = draw_frame(rng)
        h, w = cfa.shape
        tw, th, zoom, pan = draw_view(rng, h, w)
        params = draw_params(rng)
        wb = [f32(x) for x in rng.uniform(0.5, 3.0, 4)]
        cm = [f32(x) for x in (rng.uniform(-1.0, 2.0, 9) if rng.random() < 0.7 else np.eye(3).reshape(-1))]
        block = wr.uniform_block(params, wb, cm, zoom, pan[0], pan[1])
        u = ref_c.make_uniforms(params, wb, cm, zoom, pan[0], pan[1])
        for flavour, make in LOWERINGS.items():
            r = wr.render(src, cfa, block, tw, th, lowering=make())
            o = ref_c.render_f32(cfa, u, tw, th, pow_mode=POW_MODE[flavour])
            if not np.array_equal(o.view(np.uint32), r["rgba"].view(np.uint32)):
                bad = np.argwhere((o.view(np.uint32) != r["rgba"].view(np.uint32)).any(axis=2))
                print(f"MISMATCH frame {n} {flavour}: {len(bad)} pixel(s), first at (row, col) = {tuple(bad[0])}")
                print("  cfa", cfa.tolist(), "\n  target", (tw, th), "zoom", zoom, "pan", pan, "\n  params", params, "\n  wb", wb, "cm", cm)
                print("  oracle", o[tuple(bad[0])].tolist(), "evaluated", r["rgba"][tuple(bad[0])].tolist())
                raise SystemExit(1)
        px += tw * th
        oob += r["oob_loads"]
        border_px += int(((r["tex"] < 0) | (r["tex"] > 1)).any(axis=2).sum())
        nan_px += int((r["rgba"][..., :3] == 0).all(axis=2).sum())
        interior += int(((r["rgba"][..., :3] > 0) & (r["rgba"][..., :3] < 1)).any(axis=2).sum())
        if (n + 1) % 250 == 0:
            print(f"# {n + 1} frames, {px} pixels, {time.time() - t0:.0f} s", flush=True)
    print(f"wgsl_fuzz: seed {a.seed:#x}: {a.frames} frames, {px} output pixels x 2 pow flavours, evaluated shader text == C oracle "
          f"bit for bit on all of them; {border_px} pixels outside the image, {oob} out-of-bounds centre loads (tex_coords == 1.0), "
          f"{nan_px} pixels that came out (0, 0, 0), {interior} with a channel strictly inside (0, 1); {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
