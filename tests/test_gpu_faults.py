"""-m gpu: C++ exceptions injected at the places that really allocate or start threads (rd_debug_inject_fault, rawdev.h).

tests/test_abi_nothrow_cpu.py proves the wrapper at the entry points; here the fault fires in the middle of the work -- a
vector that grows with device memory in hand, the per-stream scheduler state, the descriptor array of a multi-frame launch,
the k-th worker thread of a node batch, a job running ON a worker thread -- and the checks are: a status and a message come
back (the process does not terminate), nothing is left half-built, and the SAME handle gives the oracle's bits on the next
call.  Reference contract: RenderPipeline::new returns Result<Self, String> (pipeline.rs:122, :156, :169).
"""
import ctypes as C

import numpy as np
import pytest

from raweditor_amd import _lib
from tests.gpu_util import DevBuf, sync
from tests.helpers import CM_TEST, WB_DAYLIGHT, random_cfa, random_params

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def disarm():
    yield
    _lib.inject_fault(None, 0)


def _free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    _lib.check(_lib.lib().rd_device_memory(0, C.byref(f), C.byref(t)))
    return f.value


def _frame(rng, h=70, w=384):
    cfa = random_cfa(rng, h, w)
    return cfa, random_params(rng)


def test_pipeline_create_and_render_survive_faults(gpu_lib, refc):
    ra = gpu_lib
    rng = np.random.default_rng(51)
    cfa, p = _frame(rng)
    h, w = cfa.shape
    exp = refc.render_f32(cfa, refc.make_uniforms(p, WB_DAYLIGHT, CM_TEST))
    ra.RenderPipeline.new(1, cfa.reshape(-1), w, h, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST).close()    # first use: tables, runtime pools
    sync()
    free0 = _free_bytes()
    for kind, code in ((_lib.FAULT_BAD_ALLOC, _lib.RD_ERR_OOM), (_lib.FAULT_RUNTIME, _lib.RD_ERR_INTERNAL)):
        _lib.inject_fault("pipeline.lanes", kind)                 # the lane vector's reserve, with the pipeline object built
        with pytest.raises(ra.RawdevError) as ei:
            ra.RenderPipeline.new(1, cfa.reshape(-1), w, h, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST)
        assert ei.value.code == code and "rd_pipeline_create" in ei.value.message
    pipe = ra.RenderPipeline.new(1, cfa.reshape(-1), w, h, ra.EditParams(**p), WB_DAYLIGHT, CM_TEST)
    _lib.inject_fault("scratch.entry", _lib.FAULT_BAD_ALLOC)      # first render on this stream: its scheduler state
    with pytest.raises(ra.RawdevError) as ei:
        pipe.render(fmt=ra.FMT_RGBA_F32)
    assert ei.value.code == _lib.RD_ERR_OOM and "rd_render" in ei.value.message
    got, hist = pipe.render(fmt=ra.FMT_RGBA_F32, with_histogram=True)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    assert np.array_equal(hist, refc.histogram(refc.pack_u8(exp)))
    _lib.inject_fault("rd_render", _lib.FAULT_FOREIGN)            # the nested entry (rd_render under rd_render_full_res_to_bytes)
    with pytest.raises(ra.RawdevError) as ei:
        pipe.render_full_res_to_bytes()
    assert ei.value.code == _lib.RD_ERR_INTERNAL
    assert np.array_equal(pipe.render_full_res_to_bytes().reshape(h, w, 4), refc.pack_u8(exp))
    pipe.close()
    sync()
    assert _free_bytes() >= free0 - (8 << 20), "device memory leaked by the failed creates"


def test_batch_develop_survives_a_fault_in_the_descriptor_array(gpu_lib, refc):
    ra = gpu_lib
    rng = np.random.default_rng(52)
    h, w, n = 66, 256, 5
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * 4) for _ in range(n)]
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_U8, True)
    frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    for kind, code in ((_lib.FAULT_BAD_ALLOC, _lib.RD_ERR_OOM), (_lib.FAULT_FOREIGN, _lib.RD_ERR_INTERNAL)):
        _lib.inject_fault("batch.descs", kind)
        with pytest.raises(ra.RawdevError) as ei:
            be.develop(frames)
        assert ei.value.code == code and ei.value.message.startswith("rd_batch_develop:")
    be.develop(frames)
    be.histogram(d_hist.ptr)
    sync()
    exp_hist = np.zeros(768, np.uint64)
    for c, p, o in zip(cfas, params, d_out):
        e = refc.pack_u8(refc.render_f32(c, refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)))
        exp_hist += refc.histogram(e).reshape(-1).astype(np.uint64)
        assert np.array_equal(o.to_array(np.uint8, (h, w, 4)), e)
    assert np.array_equal(d_hist.to_array(np.uint64, (768,)), exp_hist), "a failed call must not have counted anything"
    be.close()
    _lib.inject_fault("exporter.slots", _lib.FAULT_BAD_ALLOC)
    with pytest.raises(ra.RawdevError) as ei:
        ra.Exporter(0, w, h, ra.FMT_RGB_U8)
    assert ei.value.code == _lib.RD_ERR_OOM
    ra.Exporter(0, w, h, ra.FMT_RGB_U8).close()


def test_node_batch_thread_start_failure_is_a_status(gpu_lib, refc, monkeypatch):
    """Round 4's hazard: the k-th std::thread cannot be started.  It was std::terminate (joinable threads destroyed by the
    unwinding vector) on every call; now the workers are started once, in create, and a failure joins what was started."""
    ra = gpu_lib
    monkeypatch.setenv("RD_NODE_REDUCE", "host")                  # device 0 four times: the one-GPU rehearsal
    rng = np.random.default_rng(53)
    h, w, n = 34, 256, 9
    for k in range(4):                                            # the first, second, third, fourth worker fails to start
        _lib.inject_fault("node.thread", _lib.FAULT_THREAD_START, after=k)
        with pytest.raises(ra.RawdevError) as ei:
            ra.NodeBatch([0, 0, 0, 0], w, h, ra.FMT_RGBA_U8, True)
        assert ei.value.code == _lib.RD_ERR_INTERNAL
        assert f"({k + 1} of 4)" in ei.value.message and "thread" in ei.value.message, ei.value.message
    nb = ra.NodeBatch([0, 0, 0, 0], w, h, ra.FMT_RGBA_U8, True)
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * 4) for _ in range(n)]
    frames = ra.BatchExporter.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    # a fault in the dealing (host vectors), then one INSIDE a worker's job: both are statuses, both leave the handle usable
    _lib.inject_fault("node.share", _lib.FAULT_BAD_ALLOC)
    with pytest.raises(ra.RawdevError) as ei:
        nb.develop(frames)
    assert ei.value.code == _lib.RD_ERR_OOM and ei.value.message.startswith("rd_node_batch_develop:")
    _lib.inject_fault("batch.descs", _lib.FAULT_RUNTIME, after=2)     # the third device's rd_batch_develop, on its worker thread
    with pytest.raises(ra.RawdevError) as ei:
        nb.develop(frames)
    assert ei.value.code == _lib.RD_ERR_INTERNAL and "device 0: rd_batch_develop:" in ei.value.message, ei.value.message
    nb.synchronize()
    nb.histogram()                                                # drop what the three healthy devices counted in the failed call
    for _ in range(3):                                            # the workers are still there, call after call
        nb.develop(frames)
        got_hist = nb.histogram().reshape(-1)
        exp_hist = np.zeros(768, np.uint64)
        for c, p, o in zip(cfas, params, d_out):
            e = refc.pack_u8(refc.render_f32(c, refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)))
            exp_hist += refc.histogram(e).reshape(-1).astype(np.uint64)
            assert np.array_equal(o.to_array(np.uint8, (h, w, 4)), e)
        assert np.array_equal(got_hist, exp_hist)
    nb.close()
