"""The RGBA-f16 export path's slow region: lanes with 0 < x < 2^-16 (linear values below 1.5e-5: an 8-bit code under 2) take the
pinned evaluation instead of the threshold tables (rd_f16_lut_lookup).  How much of a frame must sit there before it shows, and
what the worst case costs: 24 MP frames whose CFA samples are 1 (of 4095) in a growing share of pixels and ~mid-grey elsewhere,
developed five stops down (exposure -5: a sample of 1 becomes 7.6e-6 .. 1.5e-5) and as shot."""
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
NF, NIN, NOUT = 64, 4, 8


def main():
    rng = np.random.default_rng(3)
    outs = [DevBuf(H * W * 8) for _ in range(NOUT)]
    print(f"surface f16, fused histogram on, {NF} frames of {W}x{H}, identity matrix")
    for share in (0.0, 0.001, 0.01, 0.1, 1.0):
        ins = []
        for _ in range(NIN):
            cfa = rng.integers(1500, 2600, (H, W), dtype=np.uint16)
            cfa[rng.random((H, W)) < share] = 1
            ins.append(DevBuf.from_array(cfa))
        for name, p in (("exposure -5", ra.EditParams(exposure=-5.0)), ("as shot", ra.EditParams())):
            be = ra.BatchExporter(0, W, H, ra.FMT_RGBA_F16, True)
            fr = be.make_frames([ins[i % NIN].ptr for i in range(NF)], [outs[i % NOUT].ptr for i in range(NF)], [p] * NF, WB, IDENT)
            be.develop(fr); sync()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); be.develop(fr); sync(); ts.append((time.perf_counter() - t0) / NF * 1e6)
            be.close()
            med = sorted(ts)[len(ts) // 2]
            print(f"{share * 100:6.1f} % of the samples are 1, {name:12s}: {med:6.1f} us per frame", flush=True)
        for b in ins:
            b.free()


if __name__ == "__main__":
    main()
