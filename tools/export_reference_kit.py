#!/usr/bin/env python3
"""Write tests/golden/reference_kit/: the full-resolution cases of tests/golden/develop_golden.npz as plain files a Rust
`#[test]` inside the reference can read with std + serde_json alone (the reference-side parity kit).

    cases.json            [{name, width, height, params{10 sliders}, wb[4], cm[9]}]   (EditParams' own serde field names)
    <name>.cfa.u16le      width*height little-endian u16, row-major: RenderPipeline::new's raw_data
    <name>.rgba8          width*height*4 bytes: what render_full_res_to_bytes must return, to <= 1 code per channel

Only cases the reference itself can express are exported: target = frame size, zoom 1 / pan 0, no black level.  The expected
bytes are THIS repo's pinned evaluation (oracle/develop_ref.c); a real wgpu driver may differ by one code where 255*gamma
lands within its pow()'s error of a half-integer (the seven self-pinned choices).
tests/test_golden_cpu.py checks the kit against the .npz, so it cannot drift.  Run from the repo root.
"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("exposure", "contrast", "highlights", "shadows", "whites", "blacks", "vibrance", "saturation", "temperature", "tint")
DEFAULTS = dict.fromkeys(FIELDS, 0.0)
DEFAULTS["whites"] = 1.0


def kit_cases(z):
    cases = json.loads(bytes(z["cases_json"]).decode())
    for c in cases:
        cfa = z[c["name"] + "/cfa"]
        h, w = cfa.shape
        if c["zoom"] != 1.0 or c["pan"] != [0.0, 0.0] or c["black_level"] != 0 or (c["tw"], c["th"]) != (w, h):
            continue
        params = dict(DEFAULTS)
        params.update(c["params"])
        yield {"name": c["name"], "width": w, "height": h, "params": {f: float(np.float32(params[f])) for f in FIELDS},
               "wb": c["wb"], "cm": c["cm"]}, cfa, z[c["name"] + "/u8"]


def main():
    z = np.load(os.path.join(ROOT, "tests", "golden", "develop_golden.npz"))
    out = os.path.join(ROOT, "tests", "golden", "reference_kit")
    os.makedirs(out, exist_ok=True)
    meta = []
    for m, cfa, u8 in kit_cases(z):
        cfa.astype("<u2").tofile(os.path.join(out, m["name"] + ".cfa.u16le"))
        np.ascontiguousarray(u8, dtype=np.uint8).tofile(os.path.join(out, m["name"] + ".rgba8"))
        meta.append(m)
    with open(os.path.join(out, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(f"{len(meta)} cases -> {out}")


if __name__ == "__main__":
    main()
