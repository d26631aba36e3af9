#!/usr/bin/env python3
"""How far the implementation-defined part of WGSL can move the reference's OWN text -- build container only.

Evaluates the shader string of /root/reference/src/gpu/shaders.rs (oracle/wgsl_eval.py) on the inputs of
tests/golden/wgsl_golden.npz under the lowering this repository pinned and under alternatives a driver could pick (pow as a
binary64 pow rounded once; mix as x + (y - x) a; both; division as a multiplication by the reciprocal; multiplications fused into the additions that consume them the way an
LLVM-style compiler orders it), packs each result to RGBA8 with the pinned UNORM rule and reports, against
the pinned lowering: the share of colour bytes that differ, the largest difference in codes, and the largest f32 difference in
units of the last place.  It measures the room DESIGN.md section 2 talks about on the text itself, not on a restatement.

    python tools/wgsl_lowering_room.py --frames 400 > profiles/r06_wgsl_lowering_room.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_c, wgsl_eval as we, wgsl_render as wr  # noqa: E402
from tests.helpers import CM_TEST, WB_DAYLIGHT, mild_params, random_cfa, random_params, ulp_diff  # noqa: E402
from tests.test_wgsl_pin_cpu import CASES  # noqa: E402
from tools.make_wgsl_golden import pow_pinned_scalar, shader_source  # noqa: E402

ALTERNATIVES = {
    "pow = binary64 pow rounded once": lambda: we.Lowering(pow=we.pow_f64_rounded),
    "mix = x + (y - x) a": lambda: we.Lowering(pow=pow_pinned_scalar, mix_form="x+(y-x)*a"),
    "both": lambda: we.Lowering(pow=we.pow_f64_rounded, mix_form="x+(y-x)*a"),
    "contraction (LLVM operand order)": lambda: we.Lowering(pow=pow_pinned_scalar, contraction="fuse"),
    "contraction + mix x + (y - x) a": lambda: we.Lowering(pow=pow_pinned_scalar, contraction="fuse", mix_form="x+(y-x)*a"),
    "contraction + both": lambda: we.Lowering(pow=we.pow_f64_rounded, contraction="fuse", mix_form="x+(y-x)*a"),
    "division = x * (1 / y)": lambda: we.Lowering(pow=pow_pinned_scalar, division="reciprocal"),
    "all four": lambda: we.Lowering(pow=we.pow_f64_rounded, contraction="fuse", mix_form="x+(y-x)*a", division="reciprocal"),
}


def larger_sample(src, frames):
    """`frames` random 16 x 24 frames (mild and whole-UI-range stacks alternating), the most driver-like lowering of the text
    against the pinned one (the C oracle, which reproduces the pinned evaluation bit for bit: tests/test_wgsl_pin_cpu.py)."""
    rng = np.random.default_rng(0x4C4F5752)
    total = differing = worst = one = 0
    hist = {}
    for n in range(frames):
        cfa = random_cfa(rng, 16, 24)
        params = mild_params(rng) if n % 4 else random_params(rng)
        block = wr.uniform_block(params, WB_DAYLIGHT, CM_TEST)
        alt = wr.render(src, cfa, block, lowering=ALTERNATIVES["all four"]())["rgba"]
        pinned = ref_c.render_f32(cfa, ref_c.make_uniforms(params, WB_DAYLIGHT, CM_TEST))
        d = np.abs(ref_c.pack_u8(alt)[..., :3].astype(np.int32) - ref_c.pack_u8(pinned)[..., :3].astype(np.int32))
        total += d.size
        differing += int((d != 0).sum())
        for v in d[d != 0].tolist():
            hist[v] = hist.get(v, 0) + 1
        worst = max(worst, int(d.max()))
        if (n + 1) % 50 == 0:
            print(f"# {n + 1} frames, {total} colour bytes, {differing} differ", flush=True)
    print(f"larger sample, all four (pow, mix, contraction, division) against the pinned lowering: {frames} frames of 16 x 24, {total} colour bytes, "
          f"{differing} differ ({100.0 * differing / total:.4f} %), by codes: {dict(sorted(hist.items()))}, largest {worst}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=0, help="also run the larger random sample (about 0.8 s per frame)")
    a = ap.parse_args()
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
    print("# the same evaluations of the text with contraction, against this repository's OPT-IN contracted arithmetic mode")
    print("# (oracle/develop_ref.c: colour_stack_contracted -- its own choice of which products to fuse + a reciprocal multiply for the")
    print("# levels division; the default, strict mode is what the fixture pins):")
    for label in ("contraction (LLVM operand order)",):
        total = differing = worst = 0
        for c in CASES:
            block = wr.uniform_block(c["params"], c["wb"], c["cm"], c["zoom"], *c["pan"])
            alt = wr.render(src, c["cfa"], block, c["tw"], c["th"], lowering=ALTERNATIVES[label]())["rgba"]
            u = ref_c.make_uniforms(c["params"], c["wb"], c["cm"], c["zoom"], *c["pan"], math_mode=ref_c.MATH_CONTRACTED)
            mine = ref_c.render_f32(c["cfa"], u, c["tw"], c["th"])
            d = np.abs(ref_c.pack_u8(alt)[..., :3].astype(np.int32) - ref_c.pack_u8(mine)[..., :3].astype(np.int32))
            total += d.size
            differing += int((d != 0).sum())
            worst = max(worst, int(d.max()))
        print(f"{label:36s}{total:12d}{differing:12d}{100.0 * differing / total:10.4f} %{worst:14d}")
    if a.frames:
        larger_sample(src, a.frames)


if __name__ == "__main__":
    main()
