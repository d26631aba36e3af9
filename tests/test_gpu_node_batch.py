"""-m gpu: the node-level batch entry of the C ABI (rd_node_batch_*, SURVEY.md section 8b "Batch" / 8e).

On the one-GPU box: N = 1 through the node entry equals rd_batch bit for bit (no communicator); N = 2 and 4 are rehearsed
with device 0 listed several times and the histograms folded on the host (RD_NODE_REDUCE=host) -- dealing, one rd_batch +
stream + host thread per entry and the fold are the code a real node runs; and the RCCL leg itself (librccl.so loaded
with dlopen, ncclCommInitAll, grouped ncclAllReduce of 768 x u64, in place) runs with a one-rank communicator
(RD_NODE_REDUCE=rccl).  N > 1 on distinct devices is not measurable here.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.gpu_util import DevBuf, sync
from tests.helpers import CM_TEST, WB_DAYLIGHT, random_cfa, random_params

pytestmark = pytest.mark.gpu


def _inputs(ra, refc, h, w, n, seed):
    rng = np.random.default_rng([0x52415745, seed])
    cfas = [random_cfa(rng, h, w) for _ in range(n)]
    params = [ra.EditParams(**random_params(rng)) for _ in range(n)]
    exp = []
    hist = np.zeros(768, np.uint64)
    for c, p in zip(cfas, params):
        u = refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)
        e = refc.render_f32(c, u)
        exp.append(e)
        hist += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
    return cfas, params, exp, hist


def _run_node(ra, refc, devices, h, w, n, fmt, seed=1):
    cfas, params, exp, exp_hist = _inputs(ra, refc, h, w, n, seed)
    bpp = ra.BYTES_PER_PIXEL[fmt]
    nb = ra.NodeBatch(devices, w, h, fmt, True)
    d_in = [DevBuf.from_array(c, device=nb.device_of(i)) for i, c in enumerate(cfas)]
    d_out = [DevBuf(h * w * bpp, device=nb.device_of(i)) for i in range(n)]
    frames = ra.BatchExporter.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    for rep in range(2):                                   # second pass: accumulators were reset by histogram()
        nb.develop(frames)
        got_hist = nb.histogram().reshape(-1)
        assert np.array_equal(got_hist, exp_hist), f"pass {rep}"
        for e, o in zip(exp, d_out):
            if fmt == ra.FMT_RGBA_F32:
                assert np.array_equal(o.to_array(np.float32, (h, w, 4)).view(np.uint32), e.view(np.uint32))
            else:
                assert np.array_equal(o.to_array(np.uint8, (h, w, 4)), refc.pack_u8(e))
    nb.develop(frames)                                     # two calls before one histogram(): counts add up
    m = n // 2 if n > 1 else n
    first = ra.BatchExporter.make_frames([b.ptr for b in d_in[:m]], [b.ptr for b in d_out[:m]], params[:m], WB_DAYLIGHT, CM_TEST)
    nb.develop(first)
    nb.synchronize()
    half = np.zeros(768, np.uint64)
    for e in exp[:m]:
        half += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
    assert np.array_equal(nb.histogram().reshape(-1), exp_hist + half)
    nb.close()
    return exp_hist


def test_node_batch_one_device_equals_rd_batch(gpu_lib, refc):
    ra = gpu_lib
    h, w, n = 260, 384, 7
    exp_hist = _run_node(ra, refc, [0], h, w, n, ra.FMT_RGBA_F32)
    # the same frames through rd_batch: same histogram (surfaces were compared with the oracle in both)
    cfas, params, exp, _ = _inputs(ra, refc, h, w, n, 1)
    d_in = [DevBuf.from_array(c) for c in cfas]
    d_out = [DevBuf(h * w * 16) for _ in range(n)]
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F32, True)
    frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
    be.develop(frames)
    be.histogram(d_hist.ptr)
    sync()
    assert np.array_equal(d_hist.to_array(np.uint64, (768,)), exp_hist)
    be.close()


@pytest.mark.parametrize("n_dev", [2, 4])
def test_node_batch_rehearsal_on_one_gpu(gpu_lib, refc, monkeypatch, n_dev):
    ra = gpu_lib
    with pytest.raises(ra.RawdevError):                    # a device listed twice needs the host fold
        ra.NodeBatch([0] * n_dev, 64, 34, ra.FMT_RGBA_U8)
    monkeypatch.setenv("RD_NODE_REDUCE", "host")
    _run_node(ra, refc, [0] * n_dev, 260, 384, 9, ra.FMT_RGBA_U8, seed=n_dev)
    _run_node(ra, refc, [0] * n_dev, 34, 48, 3, ra.FMT_RGBA_F32, seed=10 + n_dev)    # fewer frames than 2 x devices


def test_node_batch_rccl_leg_with_one_rank(gpu_lib, refc, monkeypatch):
    """librccl.so through dlopen, ncclCommInitAll(1 device), grouped in-place ncclAllReduce(768, ncclUint64, ncclSum)."""
    ra = gpu_lib
    monkeypatch.setenv("RD_NODE_REDUCE", "rccl")
    _run_node(ra, refc, [0], 260, 384, 5, ra.FMT_RGBA_F32, seed=3)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from tests.cpp.build_standin import build_rccl_standin  # noqa: E402  (no pytest in there: __graft_entry__.build() uses it too)


def test_node_batch_rccl_branch_with_several_ranks_through_the_standin(gpu_lib):
    """The RCCL branch of rd_node_batch (ncclCommInitAll, the grouped in-place ncclAllReduce loop, 'every device holds the
    sum') with n = 2 and 4 ranks on ONE GPU, through a host-memory stand-in for librccl (tests/cpp/rccl_standin.cpp).
    A fresh process: librawdev binds its RCCL library once.  Test infrastructure -- says nothing about RCCL / xGMI.
    Naming the stand-in alone (RAWDEV_RCCL_LIB without RD_NODE_REDUCE=standin) must NOT relax the one-rank-per-device rule."""
    ra = gpu_lib
    env0 = dict(os.environ, RAWDEV_RCCL_LIB=build_rccl_standin())
    env0.pop("RD_NODE_REDUCE", None)
    probe = ("import raweditor_amd as ra\n"
             "try:\n    ra.NodeBatch([0, 0], 64, 34, ra.FMT_RGBA_U8)\n    print('accepted')\n"
             "except ra.RawdevError as e:\n    print('refused', e.code)\n")
    out0 = subprocess.run([sys.executable, "-c", probe], env=env0, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert "refused" in out0.stdout, out0.stdout + out0.stderr
    env = dict(os.environ, RAWDEV_RCCL_LIB=build_rccl_standin(), RD_NODE_REDUCE="standin")   # the explicit opt-in
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "node_rccl_standin_run.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "rccl stand-in ok" in out.stdout


def test_node_batch_rejects_bad_arguments(gpu_lib):
    ra = gpu_lib
    with pytest.raises(ra.RawdevError):
        ra.NodeBatch([], 8, 8, ra.FMT_RGBA_F32)
    with pytest.raises(ra.RawdevError):
        ra.NodeBatch([99], 8, 8, ra.FMT_RGBA_F32)          # no such device
    nb = ra.NodeBatch([0], 8, 8, ra.FMT_RGBA_F32, with_histogram=False)
    with pytest.raises(ra.RawdevError):
        nb.histogram()
    nb.close()


def test_histogram_in_two_halves(gpu_lib, refc, monkeypatch):
    """rd_node_batch_histogram_enqueue / _fetch (round 5): the fold, the reduction and the read-back are enqueued behind the
    develop call, the next develop call is enqueued behind them without a drain, and fetch hands out the sum of EVERY interval
    enqueued since the last fetch (ABI 5: round 5 returned the last interval and dropped the others -- ADVICE round 5);
    N = 1 (no exchange) and the one-GPU rehearsal of N = 3 with the host fold."""
    ra = gpu_lib
    h, w, n = 130, 256, 6
    for devices, env in (([0], None), ([0, 0, 0], "host")):
        if env:
            monkeypatch.setenv("RD_NODE_REDUCE", env)
        cfas, params, exp, exp_hist = _inputs(ra, refc, h, w, n, 7)
        nb = ra.NodeBatch(devices, w, h, ra.FMT_RGBA_U8, True)
        d_in = [DevBuf.from_array(c) for c in cfas]
        d_out = [DevBuf(h * w * 4) for _ in range(n)]
        frames = ra.BatchExporter.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
        half = ra.BatchExporter.make_frames([b.ptr for b in d_in[:3]], [b.ptr for b in d_out[:3]], params[:3], WB_DAYLIGHT, CM_TEST)
        with pytest.raises(ra.RawdevError):
            nb.histogram_fetch()                                  # nothing enqueued yet
        for _ in range(3):                                        # three steps queued back to back, no synchronise in between
            nb.develop(frames)
            nb.histogram_enqueue()
        nb.develop(half)
        nb.histogram_enqueue()
        got = nb.histogram_fetch().reshape(-1)                    # four intervals: three whole batches + the three frames of `half`
        exp_half = np.zeros(768, np.uint64)
        for e in exp[:3]:
            exp_half += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
        assert np.array_equal(got, 3 * exp_hist.reshape(-1).astype(np.uint64) + exp_half), devices
        with pytest.raises(ra.RawdevError):
            nb.histogram_fetch()                                  # nothing enqueued since that fetch
        nb.develop(half)                                          # the running sums started again with the fetch
        nb.histogram_enqueue()
        assert np.array_equal(nb.histogram_fetch().reshape(-1), exp_half), devices
        nb.develop(frames)
        assert np.array_equal(nb.histogram().reshape(-1), exp_hist), devices     # the one-call form still synchronises and resets
        nb.synchronize()
        for e, o in zip(exp, d_out):
            assert np.array_equal(o.to_array(np.uint8, (h, w, 4)), refc.pack_u8(e))
        nb.close()
