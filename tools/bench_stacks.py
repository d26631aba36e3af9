"""Per-launch time of the export path (256 x 24 MP, surface f32 | f16 | u8 = argv[1], + histogram) for slider stacks with more or fewer
untouched sliders: shows what the exact identity-step elision (rd_uniforms.h: RD_EL_*) buys on realistic edits.
RD_NO_ELIDE=1 in the environment switches the elision off for an A/B in a second process."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raweditor_amd as ra
from tests.gpu_util import DevBuf, sync

W, H = 6016, 4016
WB = (2.0, 1.0, 1.5, 1.0)
IDENT = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)
CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
NF, NIN, NOUT = 256, 32, 8
FMT_NAME = sys.argv[1] if len(sys.argv) > 1 else "f32"


def main():
    rng = np.random.default_rng(1)
    ins = [DevBuf.from_array(rng.integers(0, 4096, (H, W), dtype=np.uint16)) for _ in range(NIN)]
    fmt = {"f32": ra.FMT_RGBA_F32, "f16": ra.FMT_RGBA_F16, "u8": ra.FMT_RGBA_U8}[FMT_NAME]
    outs = [DevBuf(H * W * ra.BYTES_PER_PIXEL[fmt]) for _ in range(NOUT)]
    print(f"surface {FMT_NAME}, fused histogram on, {NF} frames of {W}x{H}")
    stacks = {
        "all sliders default, identity matrix": (lambda i: ra.EditParams(), IDENT),
        "exposure + contrast + whites/blacks, identity matrix": (lambda i: ra.EditParams(exposure=0.7, contrast=5.0, whites=1.05, blacks=0.02), IDENT),
        "exposure + contrast + vibrance + saturation, identity": (lambda i: ra.EditParams(exposure=0.5, contrast=8.0, vibrance=0.4, saturation=15.0), IDENT),
        "exposure + highlights + shadows + temperature, identity": (lambda i: ra.EditParams(exposure=-0.4, highlights=-0.5, shadows=0.3, temperature=0.2), IDENT),
        "all ten sliders randomised, identity matrix": (lambda i: ra.EditParams.random(np.random.default_rng([7, i])), IDENT),
        "all ten sliders randomised, camera matrix (bench.py)": (lambda i: ra.EditParams.random(np.random.default_rng([7, i])), CM),
    }
    for name, (mk, cm) in stacks.items():
        be = ra.BatchExporter(0, W, H, fmt, True)
        fr = be.make_frames([ins[i % NIN].ptr for i in range(NF)], [outs[i % NOUT].ptr for i in range(NF)], [mk(i) for i in range(NF)], WB, cm)
        be.develop(fr); sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); be.develop(fr); sync(); ts.append((time.perf_counter() - t0) / NF * 1e6)
        be.close()
        med = sorted(ts)[len(ts) // 2]
        print(f"{name:58s}: {med:6.1f} us per frame  {W * H / med:9.0f} MP/s", flush=True)


if __name__ == "__main__":
    main()
