#!/usr/bin/env python3
"""How far the implementation-defined part of WGSL can move the reference's OWN text -- build container only.

Evaluates the shader string of /root/reference/src/gpu/shaders.rs (oracle/wgsl_eval.py) on the inputs of
tests/golden/wgsl_golden.npz under the lowering this repository pinned and under alternatives a driver could pick (pow as a
binary64 pow rounded once; mix as x + (y - x) a; both), packs each result to RGBA8 with the pinned UNORM rule and reports, against
the pinned lowering: the share of colour bytes that differ, the largest difference in codes, and the largest f32 difference in
units of the last place.  It measures the room DESIGN.md section 2 talks about on the text itself, not on a restatement.

    python tools/wgsl_lowering_room.py > profiles/r06_wgsl_lowering_room.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_c, wgsl_eval as we, wgsl_render as wr  # noqa: E402
from tests.helpers import ulp_diff  # noqa: E402
from tests.test_wgsl_pin_cpu import CASES  # noqa: E402
from tools.make_wgsl_golden import pow_pinned_scalar, shader_source  # noqa: E402

ALTERNATIVES = {
    "pow = binary64 pow rounded once": lambda: we.Lowering(pow=we.pow_f64_rounded),
    "mix = x + (y - x) a": lambda: we.Lowering(pow=pow_pinned_scalar, mix_form="x+(y-x)*a"),
    "both": lambda: we.Lowering(pow=we.pow_f64_rounded, mix_form="x+(y-x)*a"),
}


def main():
    src = shader_source()
    print("# lowering                          colour bytes   differing   share      max code diff   max f32 ulp   cases with a difference")
    for label, make in ALTERNATIVES.items():
        total = differing = worst = worst_ulp = ncases = 0
        for c in CASES:
            block = wr.uniform_block(c["params"], c["wb"], c["cm"], c["zoom"], *c["pan"])
            alt = wr.render(src, c["cfa"], block, c["tw"], c["th"], lowering=make())["rgba"]
            a8, p8 = ref_c.pack_u8(alt)[..., :3].astype(np.int32), ref_c.pack_u8(c["f32_pinned"])[..., :3].astype(np.int32)
            d = np.abs(a8 - p8)
            total += d.size
            differing += int((d != 0).sum())
            worst = max(worst, int(d.max()))
            worst_ulp = max(worst_ulp, ulp_diff(alt[..., :3], c["f32_pinned"][..., :3]))
            ncases += bool((d != 0).any())
        print(f"{label:36s}{total:12d}{differing:12d}{100.0 * differing / total:10.4f} %{worst:14d}{worst_ulp:14d}{ncases:10d} of {len(CASES)}")


if __name__ == "__main__":
    main()
