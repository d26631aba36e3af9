#!/usr/bin/env python3
"""Generate tests/golden/wgsl_fullsize.json: checksums of WHOLE frames made by executing the reference's shader text.

tests/golden/wgsl_golden.npz holds small frames value by value.  This script does the same thing -- the WGSL string of
/root/reference/src/gpu/shaders.rs:14-267 read where it lies and run through this repository's WGSL evaluator -- at
BASELINE.json's full sizes: 24 MP frames at the aligned width (6016 x 4016), a ragged one (6000 x 4000), an odd one
(6001 x 4001) and one 100 MP frame (11648 x 8736, the shape of BASELINE config 5).  A frame's f32 surface is 386 MB ... 1.6 GB,
so what is committed is its SHA-256, a CRC-32 per band of rows (to localise a difference), the SHA-256 of the RGBA8 and binary16
surfaces derived from it by the pinned pack rules (oracle/develop_ref.c: ref_pack_u8 / ref_pack_f16 -- fixed-function in the
reference, not shader text) and the 3 x 256 histogram of the RGBA8 bytes.  The inputs are not stored either: a frame's CFA plane
and slider stack come from numpy's PCG64 stream seeded with the frame's name (numpy's stream guarantee makes them the same on
every machine; the plane's own SHA-256 is stored so that a test can tell a different input from a different result).

The fragments are evaluated a band of rows at a time by oracle/wgsl_vec.py (the evaluator of wgsl_eval.py over arrays of
fragments; bit-identical to it on every case of wgsl_golden.npz), eight bands in parallel, with the pinned lowering
(tools/make_wgsl_golden.py: f32_pinned).  Both oracles and the HIP path must reproduce every checksum
(tests/test_wgsl_pin_cpu.py, tests/test_gpu_wgsl_pin.py).

Run from the repo root, in the build container (needs /root/reference; about ten minutes on 8 cores):
    python tools/make_wgsl_fullsize.py
"""
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_c, wgsl_eval as we, wgsl_render as wr  # noqa: E402
from tests.helpers import CM_IDENTITY, CM_TEST, WB_DAYLIGHT, mild_params, random_params  # noqa: E402
from tools.make_wgsl_golden import SEED, shader_source  # noqa: E402

BAND_ROWS = 128
FRAMES = [
    dict(name="full_6016x4016_mild", w=6016, h=4016, params="mild", cm=CM_TEST),
    dict(name="full_6016x4016_random", w=6016, h=4016, params="random", cm=CM_TEST),
    dict(name="full_6000x4000_mild", w=6000, h=4000, params="mild", cm=CM_TEST),
    dict(name="full_6000x4000_random", w=6000, h=4000, params="random", cm=CM_IDENTITY),
    dict(name="full_6001x4001_mild", w=6001, h=4001, params="mild", cm=CM_TEST),
    dict(name="full_6001x4001_random", w=6001, h=4001, params="random", cm=CM_TEST),
    dict(name="full_11648x8736_mild", w=11648, h=8736, params="mild", cm=CM_TEST),
]


def frame_inputs(spec):
    """-> (cfa u16 (h, w), params dict): the frame's own PCG64 stream (seeded with SEED and the CRC-32 of its name), the CFA plane
    first, then the sliders."""
    rng = np.random.default_rng([SEED, zlib.crc32(spec["name"].encode())])
    cfa = rng.integers(0, 4096, (spec["h"], spec["w"]), dtype=np.uint16)
    params = mild_params(rng) if spec["params"] == "mild" else random_params(rng)
    return cfa, params


_state = {}


def _band(job):
    name, r0, r1 = job
    if _state.get("name") != name:
        spec = next(f for f in FRAMES if f["name"] == name)
        cfa, params = frame_inputs(spec)
        _state.clear()
        _state.update(name=name, cfa=cfa, module=None, src=shader_source(),
                      block=wr.uniform_block(params, WB_DAYLIGHT, spec["cm"]))
    rgba, _state["module"] = wr.render_rows(_state["src"], _state["cfa"], _state["block"], we.Lowering(pow=wr.pow_pinned_lanes),
                                            r0, r1, module=_state["module"])
    return r0, rgba


def checksums(bands_in_order, w, h):
    """Consume (row0, rgba band) in row order -> the frame's record (without its inputs)."""
    sha32, sha8, sha16 = hashlib.sha256(), hashlib.sha256(), hashlib.sha256()
    crcs, hist = [], np.zeros((3, 256), np.int64)
    for r0, rgba in bands_in_order:
        b = np.ascontiguousarray(rgba, np.float32)
        sha32.update(b.tobytes())
        crcs.append(zlib.crc32(b.tobytes()))
        u8 = ref_c.pack_u8(b)
        sha8.update(u8.tobytes())
        sha16.update(ref_c.pack_f16(b).tobytes())
        for c in range(3):
            hist[c] += np.bincount(u8[..., c].reshape(-1), minlength=256)
    return dict(sha256_f32=sha32.hexdigest(), crc32_f32_bands=crcs, sha256_rgba8=sha8.hexdigest(), sha256_f16=sha16.hexdigest(),
                histogram=hist.reshape(-1).tolist())


def main():
    src = shader_source()
    out = dict(shader_file="src/gpu/shaders.rs", shader_sha256=hashlib.sha256(src.encode()).hexdigest(), seed=SEED, band_rows=BAND_ROWS,
               wb=list(map(float, WB_DAYLIGHT)), lowering="f32_pinned (tools/make_wgsl_golden.py)", raster="pixel_centre_f32",
               evaluator="oracle/wgsl_vec.py", frames=[])
    t0 = time.time()
    with mp.Pool(8) as pool:
        for spec in FRAMES:
            cfa, params = frame_inputs(spec)
            jobs = [(spec["name"], r, min(r + BAND_ROWS, spec["h"])) for r in range(0, spec["h"], BAND_ROWS)]
            # chunks of consecutive bands per worker keep the frame's inputs resident there; results come back in row order
            rec = checksums(pool.imap(_band, jobs, chunksize=1), spec["w"], spec["h"])
            rec = dict(name=spec["name"], w=spec["w"], h=spec["h"], params=params, cm=list(map(float, spec["cm"])),
                       sha256_cfa=hashlib.sha256(cfa.tobytes()).hexdigest(), **rec)
            out["frames"].append(rec)
            print(f"{spec['name']}: f32 {rec['sha256_f32'][:16]}  rgba8 {rec['sha256_rgba8'][:16]}  ({time.time() - t0:.0f} s)", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "wgsl_fullsize.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
