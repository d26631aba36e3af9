"""-m gpu: the headline workload through its OWN entry point, at its own size.

BASELINE.json's metric is quoted on `rd_batch_develop` + `rd_batch_histogram` over 6016 x 4016 frames
(configs[2]); bench.py times exactly that.  What those entry points add on top of the single-frame render
(tests/test_gpu_parity.py::test_full_size_24mp): tickets that run across frames, the u64 slab accumulation, the
fixed grid, and an output ring that wraps.  This test runs that path on 10 distinct 24 MP frames with randomised
stacks and the non-identity matrix, a ring of 8 surfaces (so it wraps), two consecutive develop() passes, and checks
  * every frame's surface on sampled row bands against the oracle, bit for bit (rows 0-5, an odd/even pair in the
    middle, the last 6 -- the output layout of pipeline.rs:526-606: tightly packed rows, row 0 on top);
  * the accumulated u64 histogram against np.bincount of the 8-bit pack of the downloaded surfaces, summed.
"""
import numpy as np
import pytest

from tests.gpu_util import DevBuf, sync
from tests.helpers import CM_TEST, WB_DAYLIGHT, random_cfa, random_params

pytestmark = pytest.mark.gpu

W, H = 6016, 4016
N_FRAMES, RING = 10, 8


def _bands(h):
    mid = (h // 2) | 1                                  # odd row, then the even one after it
    return ((0, 6), (mid, mid + 2), (h - 6, h))


def _check_frame(refc, surface, cfa, u, tag):
    for r0, r1 in _bands(H):
        exp = refc.render_band(cfa, u, r0, r1)
        assert np.array_equal(surface[r0:r1].view(np.uint32), exp.view(np.uint32)), f"{tag}: rows {r0}..{r1}"


def _bincount3(refc, surface):
    q = refc.pack_u8(surface).reshape(-1, 4)
    return np.concatenate([np.bincount(q[:, c], minlength=256) for c in range(3)]).astype(np.uint64)


def test_batch_develop_at_the_headline_size(gpu_lib, refc):
    ra = gpu_lib
    rng = np.random.default_rng([0x52415745, 24])
    cfas = [random_cfa(rng, H, W) for _ in range(N_FRAMES)]
    params = [random_params(rng) for _ in range(N_FRAMES)]
    us = [refc.make_uniforms(p, WB_DAYLIGHT, CM_TEST) for p in params]
    d_in = [DevBuf.from_array(c) for c in cfas]
    ring = [DevBuf(H * W * 16) for _ in range(RING)]
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, W, H, ra.FMT_RGBA_F32, True)

    def frames_for(order):
        return be.make_frames([d_in[i].ptr for i in order], [ring[k % RING].ptr for k in range(len(order))],
                              [ra.EditParams(**params[i]) for i in order], WB_DAYLIGHT, CM_TEST)

    per_frame_hist = {}

    def check_pass(order, tag):
        # slot k % RING holds the LAST frame written to it
        last = {}
        for k, i in enumerate(order):
            last[k % RING] = i
        for slot, i in sorted(last.items()):
            surf = ring[slot].to_array(np.float32, (H, W, 4))
            _check_frame(refc, surf, cfas[i], us[i], f"{tag} frame {i} (slot {slot})")
            assert np.all(surf[..., 3] == 1.0)
            if i not in per_frame_hist:
                per_frame_hist[i] = _bincount3(refc, surf)
            del surf

    # pass A: frames 0..9 -> slots 0..7,0,1 (frames 8, 9 overwrite 0, 1); pass B: reversed order, so frames 1, 0
    # are the ones that survive in slots 0, 1.  Two develop() calls back to back, then ONE histogram().
    order_a = list(range(N_FRAMES))
    order_b = order_a[::-1]
    fa, fb = frames_for(order_a), frames_for(order_b)
    be.develop(fa)
    sync()
    check_pass(order_a, "pass A")
    be.develop(fb)
    be.histogram(d_hist.ptr)
    sync()
    check_pass(order_b, "pass B")
    assert sorted(per_frame_hist) == order_a             # every frame's surface was downloaded and checked once
    total = sum(per_frame_hist.values())
    got = d_hist.to_array(np.uint64, (768,))
    assert int(got.sum()) == 2 * 3 * N_FRAMES * W * H
    assert np.array_equal(got, np.uint64(2) * total), "u64 histogram != 2 x sum of per-frame bincounts"

    # a third pass after the fold: the accumulator restarted from zero, the tickets from their reset state
    be.develop(fa)
    be.histogram(d_hist.ptr)
    sync()
    assert np.array_equal(d_hist.to_array(np.uint64, (768,)), total)
    surf = ring[2].to_array(np.float32, (H, W, 4))
    _check_frame(refc, surf, cfas[2], us[2], "pass C frame 2")
    be.close()
