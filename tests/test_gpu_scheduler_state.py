"""-m gpu: the per-stream scheduler state of a pipeline (ticket counters + histogram slab; rawdev.hip rd_scratch).

  * renders WITH a fused histogram enqueued from two threads on two streams at once give exact histograms (round 1
    shared one slab across streams and documented "one stream at a time");
  * counters left in an arbitrary state (simulated through the test hook -- a real GPU fault is never provoked) are
    re-zeroed on the stream before the next launch;
  * a pipeline used with many streams keeps state for at most 16 of them and stays correct.
"""
import ctypes as C
import threading

import numpy as np
import pytest

from raweditor_amd import _lib
from raweditor_amd._lib import check
from tests.gpu_util import DevBuf
from tests.helpers import CM_TEST, WB_DAYLIGHT, random_cfa, random_params

pytestmark = pytest.mark.gpu

H, W = 2008, 3008          # 24 120 tiles against 8 192 resident waves: most tiles are dealt by ticket


def _stream():
    s = C.c_void_p()
    check(_lib.lib().rd_stream_create(0, C.byref(s)))
    return s.value


def _pipe(ra, refc, seed):
    rng = np.random.default_rng([0x52415745, seed])
    cfa = random_cfa(rng, H, W)
    params = random_params(rng)
    pipe = ra.RenderPipeline.new(seed, cfa.reshape(-1), W, H, ra.EditParams(**params), WB_DAYLIGHT, CM_TEST)
    u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST)
    exp = refc.render_f32(cfa, u, nthreads=8)
    return pipe, exp, refc.histogram(refc.pack_u8(exp)).reshape(-1)


def test_histogram_renders_on_two_streams_from_two_threads(gpu_lib, refc):
    ra = gpu_lib
    pipe, exp, exp_hist = _pipe(ra, refc, 11)
    streams = [_stream(), _stream()]
    outs = [DevBuf(H * W * 16), DevBuf(H * W * 4)]
    fmts = [ra.FMT_RGBA_F32, ra.FMT_RGBA_U8]
    hists = [DevBuf(768 * 4), DevBuf(768 * 4)]
    bad = []

    def work(k):
        for it in range(12):
            pipe.render_device(W, H, fmts[k], outs[k].ptr, hists[k].ptr, streams[k])
            if it % 4 == 3:
                check(_lib.lib().rd_stream_synchronize(0, C.c_void_p(streams[k])))
                if not np.array_equal(hists[k].to_array(np.uint32, (768,)), exp_hist):
                    bad.append((k, it))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad, bad
    assert np.array_equal(outs[0].to_array(np.float32, (H, W, 4)).view(np.uint32), exp.view(np.uint32))
    assert np.array_equal(outs[1].to_array(np.uint8, (H, W, 4)), refc.pack_u8(exp))
    # the synchronous entry point (the pipeline's own stream) still agrees while the other streams' state exists
    got, hist = pipe.render(fmt=ra.FMT_RGBA_F32, with_histogram=True)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)) and np.array_equal(hist.reshape(-1), exp_hist)
    assert _lib.lib().rd_debug_scheduler_entries(pipe._h) == 3
    for s in streams:
        check(_lib.lib().rd_stream_destroy(0, C.c_void_p(s)))
    pipe.close()


def test_stale_ticket_counters_are_rezeroed(gpu_lib, refc):
    ra = gpu_lib
    pipe, exp, exp_hist = _pipe(ra, refc, 12)
    s = _stream()
    out, hist = DevBuf(H * W * 16), DevBuf(768 * 4)
    assert _lib.lib().rd_debug_poison_scheduler(pipe._h, C.c_void_p(s)) == -1          # no state for that stream yet
    for target in (s, None):                                   # a caller stream, then the pipeline's own stream
        for rep in range(3):
            if target is None:
                got, h = pipe.render(fmt=ra.FMT_RGBA_F32, with_histogram=True)
                assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)) and np.array_equal(h.reshape(-1), exp_hist)
            else:
                pipe.render_device(W, H, ra.FMT_RGBA_F32, out.ptr, hist.ptr, target)
                check(_lib.lib().rd_stream_synchronize(0, C.c_void_p(target)))
                assert np.array_equal(out.to_array(np.float32, (H, W, 4)).view(np.uint32), exp.view(np.uint32)), rep
                assert np.array_equal(hist.to_array(np.uint32, (768,)), exp_hist)
            # garbage in the counters, flagged the way a failed launch / synchronisation flags them
            check(_lib.lib().rd_debug_poison_scheduler(pipe._h, C.c_void_p(target) if target else None))
    check(_lib.lib().rd_stream_destroy(0, C.c_void_p(s)))
    pipe.close()


def test_scheduler_state_is_bounded(gpu_lib, refc):
    ra = gpu_lib
    pipe, exp, exp_hist = _pipe(ra, refc, 13)
    out, hist = DevBuf(H * W * 4), DevBuf(768 * 4)
    exp8 = refc.pack_u8(exp)
    streams = [_stream() for _ in range(40)]
    for i, s in enumerate(streams):
        pipe.render_device(W, H, ra.FMT_RGBA_U8, out.ptr, hist.ptr, s)
        check(_lib.lib().rd_stream_synchronize(0, C.c_void_p(s)))
        if i % 13 == 0 or i == len(streams) - 1:
            assert np.array_equal(out.to_array(np.uint8, (H, W, 4)), exp8), i
            assert np.array_equal(hist.to_array(np.uint32, (768,)), exp_hist), i
        assert _lib.lib().rd_debug_scheduler_entries(pipe._h) <= 16
    assert _lib.lib().rd_debug_scheduler_entries(pipe._h) == 16
    # an early stream again: its state was handed on long ago and is set up afresh
    pipe.render_device(W, H, ra.FMT_RGBA_U8, out.ptr, hist.ptr, streams[0])
    check(_lib.lib().rd_stream_synchronize(0, C.c_void_p(streams[0])))
    assert np.array_equal(hist.to_array(np.uint32, (768,)), exp_hist)
    for s in streams:
        check(_lib.lib().rd_stream_destroy(0, C.c_void_p(s)))
    pipe.close()
