"""Run by tests/test_gpu_node_batch.py in a FRESH process (librawdev binds its RCCL library once per process):
the RCCL branch of rd_node_batch_* with n > 1 ranks on a one-GPU box, through tests/cpp/librccl_standin.so.

Environment set by the caller: RAWDEV_RCCL_LIB=<the stand-in>, RD_NODE_REDUCE=standin (the explicit opt-in).  Checks, for N = 2 and 4 with device 0
listed N times: the communicator path is taken (reduce kind "rccl all-reduce"), the global histogram equals the oracle's
sum, EVERY rank's device buffer holds that sum after the grouped in-place all-reduce, surfaces are bit-identical to the
oracle, and the stand-in saw exactly N all-reduce calls per histogram() inside ONE group.  This executes librawdev's call
sequence; it proves nothing about RCCL or xGMI.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import raweditor_amd as ra  # noqa: E402
from oracle import ref_c as refc  # noqa: E402
from raweditor_amd import _lib  # noqa: E402
from tests.gpu_util import DevBuf  # noqa: E402
from tests.helpers import CM_TEST, WB_DAYLIGHT, random_cfa, random_params  # noqa: E402


def stats(standin):
    a, b, c = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
    standin.rawdev_rccl_standin_stats(C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def main():
    path = os.environ["RAWDEV_RCCL_LIB"]
    _lib.lib()                                                   # librawdev (and the HIP runtime it shares with the stand-in) first
    standin = C.CDLL(path)
    h, w = 132, 256
    total_calls = total_groups = 0
    for n_dev, n_frames, fmt in ((2, 7, ra.FMT_RGBA_F32), (4, 9, ra.FMT_RGBA_U8), (4, 3, ra.FMT_RGBA_F32)):
        rng = np.random.default_rng([0x52415745, 77, n_dev, n_frames])
        cfas = [random_cfa(rng, h, w) for _ in range(n_frames)]
        params = [ra.EditParams(**random_params(rng)) for _ in range(n_frames)]
        exp, exp_hist = [], np.zeros(768, np.uint64)
        for c, p in zip(cfas, params):
            u = refc.make_uniforms({f: getattr(p, f) for f in ra.FIELDS}, WB_DAYLIGHT, CM_TEST)
            e = refc.render_f32(c, u)
            exp.append(e)
            exp_hist += refc.histogram(refc.pack_u8(e)).reshape(-1).astype(np.uint64)
        nb = ra.NodeBatch([0] * n_dev, w, h, fmt, True)          # device 0 listed N times: accepted only with RD_NODE_REDUCE=standin
        assert nb.reduce_kind() == "rccl all-reduce", nb.reduce_kind()
        bpp = ra.BYTES_PER_PIXEL[fmt]
        d_in = [DevBuf.from_array(c) for c in cfas]
        d_out = [DevBuf(h * w * bpp) for _ in range(n_frames)]
        frames = ra.BatchExporter.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], params, WB_DAYLIGHT, CM_TEST)
        for rep in range(2):
            nb.develop(frames)
            got = nb.histogram().reshape(-1)
            assert np.array_equal(got, exp_hist), f"N={n_dev} pass {rep}: global histogram differs from the oracle sum"
            for d in range(n_dev):                                # every rank's buffer holds the sum
                part = np.zeros(768, np.uint64)
                _lib.check(_lib.lib().rd_debug_node_histogram_of(nb._h, d, part.ctypes.data_as(C.c_void_p)))
                assert np.array_equal(part, exp_hist), f"N={n_dev} pass {rep}: rank {d}'s buffer does not hold the sum"
            for e, o in zip(exp, d_out):
                if fmt == ra.FMT_RGBA_F32:
                    assert np.array_equal(o.to_array(np.float32, (h, w, 4)).view(np.uint32), e.view(np.uint32))
                else:
                    assert np.array_equal(o.to_array(np.uint8, (h, w, 4)), refc.pack_u8(e))
            total_calls += n_dev
            total_groups += 1
            calls, groups, max_ranks = stats(standin)
            assert (calls, groups) == (total_calls, total_groups), (calls, groups, total_calls, total_groups)
            assert max_ranks >= n_dev
        nb.close()
    print(f"rccl stand-in ok: {total_groups} grouped all-reduces, {total_calls} rank calls, up to {stats(standin)[2]} ranks")


if __name__ == "__main__":
    main()
