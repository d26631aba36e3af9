"""Soak test of the ticket-scheduled export kernel: tens of thousands of frames back to back on one batch context
(multi-frame launches by default; RD_BATCH_PERSISTENT=0 for one launch per frame / row band);
every tile of every launch must be processed exactly once, which the accumulated u64 histogram shows (it must be
exactly reps x the histogram of one pass) together with the bytes of the last pass (equal to the first pass)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raweditor_amd as ra
from tests.gpu_util import DevBuf, sync

# python tools/soak.py [reps=2000] [width=6016] [height=4016]   (round 5: e.g. 6000 4000 -- every row pair ends in an overlapped tile)
W = int(sys.argv[2]) if len(sys.argv) > 2 else 6016
H = int(sys.argv[3]) if len(sys.argv) > 3 else 4016
WB = (2.0, 1.0, 1.5, 1.0)
CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
NF = 8


def main():
    rng = np.random.default_rng(0x52415745)
    ins = [DevBuf.from_array(rng.integers(0, 4096, (H, W), dtype=np.uint16)) for _ in range(NF)]
    for fmt, name, bands in ((ra.FMT_RGBA_F32, "f32", 1), (ra.FMT_RGBA_U8, "u8", 3)) + (((ra.FMT_RGB_U8, "rgb8", 1),) if W % 128 else ()):
        bpp = ra.BYTES_PER_PIXEL[fmt]
        outs = [DevBuf(H * W * bpp) for _ in range(NF)]
        hist = DevBuf(768 * 8)
        be = ra.BatchExporter(0, W, H, fmt, True)
        # even frames: all ten sliders + camera matrix (the general path); odd frames: a usual edit with the identity matrix
        # (the channel-separable path) -- a launch switches paths at every frame boundary
        ps = [ra.EditParams.random(np.random.default_rng([7, i])) if i % 2 == 0 else
              ra.EditParams(exposure=0.3 * i, contrast=5.0, whites=1.05, blacks=0.02, temperature=0.1) for i in range(NF)]
        fr = be.make_frames([b.ptr for b in ins], [b.ptr for b in outs], ps, WB, CM)
        for i in range(1, NF, 2):
            fr[i].color_matrix[:] = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]
        be.develop(fr, row_bands=bands); be.histogram(hist.ptr); sync()
        h1 = hist.to_array(np.uint64, (768,)).copy()
        first = [o.to_array(np.uint8, (H * W * bpp,)).copy() for o in outs[:2]]
        assert int(h1[:256].sum()) == NF * W * H, "one pass does not count every pixel once"
        t0 = time.perf_counter()
        for r in range(REPS):
            be.develop(fr, row_bands=bands)
            if r % 500 == 499:
                sync(); print(f"{name}: {r + 1} passes", flush=True)
        be.histogram(hist.ptr); sync()
        dt = time.perf_counter() - t0
        hn = hist.to_array(np.uint64, (768,))
        ok_h = np.array_equal(hn, h1 * np.uint64(REPS))
        ok_b = all(np.array_equal(o.to_array(np.uint8, (H * W * bpp,)), f) for o, f in zip(outs[:2], first))
        lpc = be.last_launch_count()
        print(f"{name}: {REPS} passes over {NF} frames = {REPS * NF} frames in {REPS * lpc} launches ({lpc} per pass), {dt:.2f} s "
              f"({dt / (REPS * NF) * 1e6:.1f} us per frame): histogram {'exact' if ok_h else 'MISMATCH'}, "
              f"surfaces {'identical' if ok_b else 'DIFFER'}", flush=True)
        be.close()
        if not (ok_h and ok_b):
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
