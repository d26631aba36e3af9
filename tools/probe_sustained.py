"""Probe: per-launch time of the product export kernel through the C ABI (no torch), sustained,
as a function of (#distinct frames, fixed vs randomised stacks, histogram on/off)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raweditor_amd as ra
from tests.gpu_util import DevBuf, sync

W, H = 6016, 4016
WB = (2.0, 1.0, 1.5, 1.0); CM = (1.6, -0.4, -0.2, -0.3, 1.5, -0.2, 0.0, -0.5, 1.5)
rng = np.random.default_rng(1)
NMAX = 64
ins = [DevBuf.from_array(rng.integers(0, 4096, (H, W), dtype=np.uint16)) for _ in range(NMAX)]
outs = [DevBuf(H * W * 16) for _ in range(8)]
fixed = ra.EditParams(exposure=0.7, contrast=3.0, highlights=-0.3, shadows=0.4, whites=1.1, blacks=0.02,
                      vibrance=0.3, saturation=20.0, temperature=0.2, tint=-0.1)

def run(nframes, nin, nout, randomised, hist, reps=6):
    be = ra.BatchExporter(0, W, H, ra.FMT_RGBA_F32, hist)
    ps = [ra.EditParams.random(np.random.default_rng([7, i])) if randomised else fixed for i in range(nframes)]
    fr = be.make_frames([ins[i % nin].ptr for i in range(nframes)], [outs[i % nout].ptr for i in range(nframes)], ps, WB, CM)
    be.develop(fr); sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); be.develop(fr); sync(); ts.append((time.perf_counter() - t0) / nframes * 1e6)
    be.close()
    print(f"frames {nframes:4d} nin {nin:3d} nout {nout} random={int(randomised)} hist={int(hist)}: "
          + " ".join(f"{t:6.1f}" for t in ts) + " us/launch", flush=True)

run(256, 8, 4, False, True)
run(256, 8, 4, True, True)
run(256, 64, 8, False, True)
run(256, 64, 8, True, True)
run(256, 64, 8, True, False)
run(1024, 64, 8, True, True, reps=3)
for k, v in (("exposure", 4.0), ("exposure", -4.0), ("blacks", 0.2), ("contrast", 10.0), ("saturation", -100.0)):
    fixed2 = ra.EditParams(**{k: v}); fixed_saved = fixed; fixed = fixed2
    print(k, v, end=": "); run(256, 64, 8, False, True, reps=3); fixed = fixed_saved
