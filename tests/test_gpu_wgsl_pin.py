"""-m gpu: the HIP path against vectors made by EXECUTING the reference's shader text (tests/golden/wgsl_golden.npz, see
tools/make_wgsl_golden.py and oracle/wgsl_eval.py).  Nothing on the expected side of these comparisons comes from the oracle's
restatement of the shader: the f32 surface must equal, bit for bit, what /root/reference/src/gpu/shaders.rs:14-267 evaluates to
under the lowering the fixture records (the pinned pow pair, left-to-right dot, mix = x (1 - a) + y a, minNum / maxNum, clamped
loads, pixel-centre rasterisation).  The narrow surfaces are derived from those f32 values by the pack rules of DESIGN.md
section 2 (UNORM8 and binary16 conversion are fixed-function in the reference, not shader text), so they are checked through
the oracle's pack functions applied to the evaluated vectors.

Every case runs through the RenderPipeline mirror with the kernel the library picks and with the general map kernel forced;
the full-resolution cases whose width the export kernel takes (W >= 128: one tile + a ragged tail, an odd width, the
shifted-window tiling) also run through the batch entry points, in a multi-frame launch."""
import numpy as np
import pytest

from tests.gpu_util import DevBuf, sync
from tests.test_gpu_parity import _null, force_map, make_pipe
from tests.test_wgsl_pin_cpu import CASES, IDS

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("case", CASES, ids=IDS)
@pytest.mark.parametrize("kernel", ["auto", "map"])
def test_pipeline_reproduces_the_evaluated_shader(gpu_lib, refc, case, kernel):
    ra = gpu_lib
    exp = case["f32_pinned"]
    pipe = make_pipe(ra, case["cfa"], case["params"], case["wb"], case["cm"], case["zoom"], case["pan"])
    with (force_map() if kernel == "map" else _null()):
        got, hist = pipe.render(case["tw"], case["th"], ra.FMT_RGBA_F32, with_histogram=True)
        got8 = pipe.render(case["tw"], case["th"], ra.FMT_RGBA_U8)
        got16 = pipe.render(case["tw"], case["th"], ra.FMT_RGBA_F16)
    assert np.array_equal(bits(got), bits(exp))
    exp8 = refc.pack_u8(exp)
    assert np.array_equal(got8, exp8)
    assert np.array_equal(got16.view(np.uint16), refc.pack_f16(exp).view(np.uint16))
    assert np.array_equal(hist, refc.histogram(exp8))
    pipe.close()


EXPORT_CASES = [c for c in CASES if c["zoom"] == 1.0 and c["pan"] == [0.0, 0.0] and c["tw"] == c["cfa"].shape[1]
                and c["th"] == c["cfa"].shape[0] and c["cfa"].shape[1] >= 128]


def test_fixture_holds_export_kernel_widths():
    assert {c["name"] for c in EXPORT_CASES} >= {"export_tile_5x134", "export_odd_4x131", "export_shift_3x250"}


@pytest.mark.parametrize("case", EXPORT_CASES, ids=[c["name"] for c in EXPORT_CASES])
def test_batch_entry_reproduces_the_evaluated_shader(gpu_lib, refc, case):
    """Three frames per launch (the same CFA and stack in each: the fixture holds one frame per case), every surface format."""
    ra = gpu_lib
    h, w = case["cfa"].shape
    exp = case["f32_pinned"]
    exp8 = refc.pack_u8(exp)
    n = 3
    for fmt, dt, ch, want in ((ra.FMT_RGBA_F32, np.uint32, 4, bits(exp)),
                              (ra.FMT_RGBA_F16, np.uint16, 4, refc.pack_f16(exp).view(np.uint16)),
                              (ra.FMT_RGBA_U8, np.uint8, 4, exp8), (ra.FMT_RGB_U8, np.uint8, 3, exp8[..., :3])):
        d_in = [DevBuf.from_array(case["cfa"]) for _ in range(n)]
        d_out = [DevBuf(h * w * ra.BYTES_PER_PIXEL[fmt]) for _ in range(n)]
        d_hist = DevBuf(768 * 8)
        be = ra.BatchExporter(0, w, h, fmt, True)
        frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], [ra.EditParams(**case["params"])] * n,
                                case["wb"], case["cm"])
        be.develop(frames)
        be.histogram(d_hist.ptr)
        sync()
        for o in d_out:
            assert np.array_equal(o.to_array(dt, (h, w, ch)), want), (case["name"], fmt)
        assert np.array_equal(d_hist.to_array(np.uint64, (768,)), refc.histogram(exp8).reshape(-1).astype(np.uint64) * n)
        be.close()
        for b in d_in + d_out + [d_hist]:
            b.free()
