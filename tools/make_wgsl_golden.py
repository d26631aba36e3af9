#!/usr/bin/env python3
"""Generate tests/golden/wgsl_golden.npz: vectors made by EXECUTING the reference's own shader text.

The reference's develop path is the WGSL string constant PASSTHROUGH_SHADER of /root/reference/src/gpu/shaders.rs:14-267.
This script reads that string where it lies (it is never copied into this repository), runs its `vs_main` and `fs_main`
through the WGSL evaluator of oracle/wgsl_eval.py -- one fragment per output pixel, the way the render pass of
gpu/pipeline.rs:567-590 draws the full-screen triangle -- and writes inputs and outputs as a fixture.  The oracle
(oracle/develop_ref.c, oracle/develop_np.py) and the HIP path must reproduce these vectors BIT FOR BIT
(tests/test_wgsl_pin_cpu.py, tests/test_gpu_parity.py).  Unlike tests/golden/develop_golden.npz, nothing in these outputs
comes from the oracle's restatement of the shader: operation order, constants, the demosaic selection table, the
abstract-float constant folding and every type conversion are whatever the reference's text says.

What WGSL leaves to the implementation is fixed by the `Lowering` recorded in the fixture (oracle/wgsl_eval.py): two flavours
of pow --
    f32_libm     pow = binary64 pow of the C library rounded once (independent of this repository; the oracle's REF_POW_LIBM)
    f32_pinned   pow = exp2(y * log2 x) on the polynomial pair this repository pinned (oracle/develop_np.py: pow_pinned),
                 the flavour the product computes
-- dot products and matrix * vector summed left to right, mix = x (1 - a) + y a, min / max / clamp with IEEE minNum / maxNum
on NaN, textureLoad clamped at the border, and the rasteriser model "pixel_centre_f32" of oracle/wgsl_render.py.  Every case
is also drawn with the independent "barycentric_f64" rasteriser model; the script refuses to write a fixture in which the two
models disagree on any pixel.

Run from the repo root, in the build container (needs /root/reference):  python tools/make_wgsl_golden.py
"""
import hashlib
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import develop_np as dn, wgsl_eval as we, wgsl_render as wr  # noqa: E402
from tests.helpers import (CM_IDENTITY, CM_TEST, PARAM_NAMES, UI_RANGES, WB_DAYLIGHT, mild_params, random_cfa,  # noqa: E402
                           random_params)

SHADER_FILE = "/root/reference/src/gpu/shaders.rs"
SHADER_CONST = "PASSTHROUGH_SHADER"
SEED = 0x57475348


def shader_source():
    with open(SHADER_FILE, encoding="utf-8") as f:
        return we.extract_rust_raw_string(f.read(), SHADER_CONST)


def pow_pinned_scalar(x, y):
    return dn.pow_pinned(np.array([x], np.float32), y)[0]


LOWERINGS = {
    "f32_libm": lambda: we.Lowering(pow=we.pow_f64_rounded),
    "f32_pinned": lambda: we.Lowering(pow=pow_pinned_scalar),
}


def range_end_params(which):
    """Every slider at one end of its UI range: which = 0 (all low), 1 (all high), 2 / 3 (alternating)."""
    out = {}
    for n, k in enumerate(PARAM_NAMES):
        lo, hi = UI_RANGES[k]
        pick = {0: 0, 1: 1, 2: n & 1, 3: (n & 1) ^ 1}[which]
        out[k] = float(np.float32((lo, hi)[pick]))
    return out


def case_rng(name):
    """Every case draws from its own stream, keyed by its name: adding a case never changes another one's vectors."""
    return np.random.default_rng([SEED, zlib.crc32(name.encode())])


def spec():
    R = "random"
    s = [
        dict(name="default_8x12", h=8, w=12, params={}, wb=(1, 1, 1, 1), cm=CM_IDENTITY),
        dict(name="random_12x16", h=12, w=16, params=R, wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="random_odd_7x9", h=7, w=9, params=R, wb=WB_DAYLIGHT, cm=CM_IDENTITY),
        dict(name="one_pixel", h=1, w=1, params=R, wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="u16_full_range_8x10", h=8, w=10, params=R, wb=WB_DAYLIGHT, cm=CM_TEST, hi=65536),
        dict(name="preview_zoom_pan", h=12, w=16, params=R, wb=WB_DAYLIGHT, cm=CM_TEST,
             tw=11, th=7, zoom=1.75, pan=(0.125, -0.0625)),
        dict(name="zoomed_out_border", h=10, w=14, params={}, wb=WB_DAYLIGHT, cm=CM_IDENTITY,
             tw=20, th=12, zoom=0.5, pan=(0.0, 0.0)),
        # tex_coords reach exactly 1.0 in the second column and row: pixel_coords = (W, H), one past the texture
        dict(name="tex_coord_one", h=6, w=8, params=R, wb=WB_DAYLIGHT, cm=CM_TEST, tw=2, th=2, zoom=0.5, pan=(0.0, 0.0)),
        dict(name="downsampled_preview", h=12, w=18, params=R, wb=WB_DAYLIGHT, cm=CM_TEST, tw=7, th=5),
        dict(name="random_32x48", h=32, w=48, params=R, wb=WB_DAYLIGHT, cm=CM_TEST),
        # widths the export kernel takes (W >= 128): one whole 128-pixel tile + a ragged tail (four frames of that size, each with
        # its own CFA and stack: a multi-frame launch of the batch entry is checked on DIFFERENT frames); an ODD width (last
        # column by the second kernel, shifted-window stores on the f32 surface); W % 4 == 2 with the shifted tiling's last tile
        # owning one quad
        dict(name="export_tile_5x134", h=5, w=134, params=R, wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="export_tile_5x134_b", h=5, w=134, params=R, wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="export_tile_5x134_c", h=5, w=134, params=R, wb=WB_DAYLIGHT, cm=CM_IDENTITY),
        dict(name="export_tile_5x134_d", h=5, w=134, params={}, wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="export_odd_4x131", h=4, w=131, params=R, wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="export_shift_3x250", h=3, w=250, params=R, wb=WB_DAYLIGHT, cm=CM_IDENTITY),
    ]
    for which in range(4):
        s.append(dict(name=f"range_ends_{which}", h=6, w=8, params=range_end_params(which), wb=WB_DAYLIGHT, cm=CM_TEST))
    for n in range(12):
        s.append(dict(name=f"random_stack_{n}", h=6, w=10, params=R, wb=WB_DAYLIGHT, cm=CM_TEST))
    # mild edits (tests/helpers.py: mild_params): most output values strictly inside (0, 1), where every rounding of the stack
    # shows in the result (the draws over the whole UI ranges above saturate two thirds of theirs)
    M = "mild"
    s += [dict(name=f"mild_stack_{n}", h=6, w=10, params=M, wb=WB_DAYLIGHT, cm=CM_TEST) for n in range(8)]
    s += [
        dict(name="mild_preview_zoom_pan", h=12, w=16, params=M, wb=WB_DAYLIGHT, cm=CM_TEST, tw=11, th=7, zoom=1.75,
             pan=(0.125, -0.0625)),
        dict(name="mild_tex_coord_one", h=6, w=8, params=M, wb=(1, 1, 1, 1), cm=CM_IDENTITY, tw=4, th=4, zoom=0.5, pan=(0.0, 0.0)),
        dict(name="mild_export_tile_5x134", h=5, w=134, params=M, wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="mild_export_odd_4x131", h=4, w=131, params=M, wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="mild_export_odd_4x131_b", h=4, w=131, params=M, wb=WB_DAYLIGHT, cm=CM_IDENTITY),
        dict(name="mild_export_shift_3x250", h=3, w=250, params=M, wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="mild_export_two_tiles_3x262", h=3, w=262, params=M, wb=WB_DAYLIGHT, cm=CM_TEST),
    ]
    # zoom = 0 (no UI state, but the uniform block can hold it): vs_main divides by it -- tex_coords are infinities, black, except
    # where the numerator is 0 too: the centre column / row of an odd target gets a NaN, which passes the shader's bounds test
    # (every comparison is false) and converts to pixel 0 (i32 of a NaN: the lowering's choice, counted in `nan_to_int`).  The
    # barycentric rasteriser model has no meaning here (the vertices lie at infinity): `bary=False`.
    s += [
        dict(name="zoom_zero_5x3", h=6, w=8, params=M, wb=WB_DAYLIGHT, cm=CM_TEST, tw=5, th=3, zoom=0.0, pan=(0.0, 0.0), bary=False),
        dict(name="zoom_zero_pan_5x5", h=6, w=8, params=M, wb=WB_DAYLIGHT, cm=CM_TEST, tw=5, th=5, zoom=0.0, pan=(0.25, 0.0),
             bary=False),
    ]
    return s


def main():
    src = shader_source()
    out, cases = {}, []
    t0 = time.time()
    for s in spec():
        rng = case_rng(s["name"])
        if s["params"] == "random":
            s["params"] = random_params(rng)
        elif s["params"] == "mild":
            s["params"] = mild_params(rng)
        cfa = random_cfa(rng, s["h"], s["w"], s.get("hi", 4096))
        zoom, pan = s.get("zoom", 1.0), s.get("pan", (0.0, 0.0))
        tw, th = s.get("tw", s["w"]), s.get("th", s["h"])
        block = wr.uniform_block(s["params"], s["wb"], s["cm"], zoom, pan[0], pan[1])
        n = s["name"]
        info = {}
        for flavour, make in LOWERINGS.items():
            r = wr.render(src, cfa, block, tw, th, lowering=make(), raster="pixel_centre_f32")
            out[f"{n}/{flavour}"] = r["rgba"]
            info = dict(oob_loads=r["oob_loads"], nan_to_int=r["nan_to_int"], tex_ulp_between_raster_models=None)
            if s.get("bary", True):
                b = wr.render(src, cfa, block, tw, th, lowering=make(), raster="barycentric_f64")
                differ = int((r["rgba"].view(np.uint32) != b["rgba"].view(np.uint32)).any(axis=2).sum())
                if differ:
                    raise SystemExit(f"{n}/{flavour}: the two rasteriser models disagree on {differ} pixel(s); choose another case")
                info["tex_ulp_between_raster_models"] = int(np.abs(r["tex"].view(np.int32).astype(np.int64)
                                                                   - b["tex"].view(np.int32).astype(np.int64)).max())
        out[f"{n}/cfa"] = cfa
        out[f"{n}/tex"] = r["tex"]
        cases.append(dict(name=n, params=s["params"], wb=list(map(float, s["wb"])), cm=list(map(float, s["cm"])), zoom=zoom,
                          pan=list(pan), tw=tw, th=th, **info))
        print(f"{n}: {tw}x{th} from {s['w']}x{s['h']}  {info}  ({time.time() - t0:.1f} s)", flush=True)
    meta = dict(
        shader_file="src/gpu/shaders.rs", shader_const=SHADER_CONST, shader_sha256=hashlib.sha256(src.encode()).hexdigest(),
        shader_lines=src.count("\n"), evaluator="oracle/wgsl_eval.py + oracle/wgsl_render.py", raster="pixel_centre_f32",
        lowering=dict(dot="left_to_right", matrix_times_vector="dot(row, v), left to right", mix="x*(1-a)+y*a",
                      nan_minmax="other_operand", texture_oob="clamp", nan_to_int=0,
                      pow=dict(f32_libm="binary64 pow rounded once", f32_pinned="oracle/develop_np.py: pow_pinned")),
        cases=cases)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "wgsl_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(cases), "cases, shader sha256", meta["shader_sha256"][:16])


if __name__ == "__main__":
    main()
