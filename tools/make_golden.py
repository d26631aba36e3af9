#!/usr/bin/env python3
"""Generate tests/golden/develop_golden.npz: small input/expected-output vectors for the develop path.

The reference (Rust + WGSL on wgpu) cannot run in this environment and ships no fixtures for
src/gpu/, so these vectors come from the CPU oracle (oracle/develop_ref.c), cross-checked here
against the independent numpy twin (oracle/develop_np.py) before they are written.  They pin the
oracle against regressions and give the GPU tests a reference that needs no oracle build.
Run from the repo root:  python tools/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import develop_np as dn, ref_c  # noqa: E402
from tests.helpers import CM_IDENTITY, CM_TEST, WB_DAYLIGHT, random_cfa, random_params  # noqa: E402

SEED = 0x52415745


def main():
    rng = np.random.default_rng(SEED)
    cases = []
    spec = [
        dict(name="default_16x24", h=16, w=24, params=None, wb=(1, 1, 1, 1), cm=CM_IDENTITY),
        dict(name="random_16x24", h=16, w=24, params="random", wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="random_odd_9x13", h=9, w=13, params="random", wb=WB_DAYLIGHT, cm=CM_IDENTITY),
        dict(name="preview_zoom_pan", h=16, w=24, params="random", wb=WB_DAYLIGHT, cm=CM_TEST,
             tw=11, th=7, zoom=1.75, pan=(0.125, -0.0625)),
        dict(name="zoomed_out_border", h=10, w=14, params=None, wb=WB_DAYLIGHT, cm=CM_IDENTITY,
             tw=20, th=12, zoom=0.5, pan=(0.0, 0.0)),
        dict(name="u16_full_range", h=12, w=16, params="random", wb=WB_DAYLIGHT, cm=CM_TEST, hi=65536),
        dict(name="black_level_64", h=12, w=16, params="random", wb=WB_DAYLIGHT, cm=CM_IDENTITY, bl=64),
        dict(name="random_64x96", h=64, w=96, params="random", wb=WB_DAYLIGHT, cm=CM_TEST),
        # round 5 (appended: the random stream of the cases above is unchanged): a width that is not a multiple of the export
        # kernel's 128-pixel tile -- 134 = one whole tile + a last tile pulled back over 61 of its quads -- and an odd height
        dict(name="ragged_11x134", h=11, w=134, params="random", wb=WB_DAYLIGHT, cm=CM_TEST),
        # round 6 (appended likewise): an ODD width of more than one tile -- the export kernel takes its 65 whole quads, a second
        # kernel the last column, and the f32 surface stores through shifted windows (W % 4 != 0) -- and an even width with
        # W % 4 == 2 whose W / 2 = 62 T - 61 (the shifted-window tiling's last tile owns a single quad)
        dict(name="odd_7x131", h=7, w=131, params="random", wb=WB_DAYLIGHT, cm=CM_TEST),
        dict(name="shift_6x250", h=6, w=250, params="random", wb=WB_DAYLIGHT, cm=CM_IDENTITY),
    ]
    out = {}
    for s in spec:
        cfa = random_cfa(rng, s["h"], s["w"], s.get("hi", 4096))
        params = random_params(rng) if s["params"] == "random" else {}
        zoom, pan, bl = s.get("zoom", 1.0), s.get("pan", (0.0, 0.0)), s.get("bl", 0)
        tw, th = s.get("tw", s["w"]), s.get("th", s["h"])
        u = ref_c.make_uniforms(params, s["wb"], s["cm"], zoom, pan[0], pan[1], bl)
        f32 = ref_c.render_f32(cfa, u, tw, th)
        twin = dn.render_f32(cfa, dn.Uniforms(**params, wb=tuple(s["wb"]), cm=tuple(s["cm"]), zoom=zoom,
                                              pan_x=pan[0], pan_y=pan[1], black_level=bl), tw, th)
        assert np.array_equal(f32.view(np.uint32), twin.view(np.uint32)), s["name"]
        u8 = ref_c.pack_u8(f32)
        n = s["name"]
        out[n + "/cfa"] = cfa
        out[n + "/f32"] = f32
        out[n + "/u8"] = u8
        out[n + "/f16"] = ref_c.pack_f16(f32).view(np.uint16)
        out[n + "/hist"] = ref_c.histogram(u8)
        cases.append(dict(name=n, params=params, wb=list(map(float, s["wb"])), cm=list(map(float, s["cm"])),
                          zoom=zoom, pan=list(pan), black_level=bl, tw=tw, th=th))
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "develop_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
