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
    h, w = case["cfa"].shape
    if (case["tw"], case["th"], case["zoom"], case["pan"]) == (w, h, 1.0, [0.0, 0.0]) and kernel == "auto":
        # the reference's own export call (pipeline.rs:526-605)
        assert np.array_equal(pipe.render_full_res_to_bytes().reshape(h, w, 4), exp8)
    pipe.close()


EXPORT_CASES = [c for c in CASES if c["zoom"] == 1.0 and c["pan"] == [0.0, 0.0] and c["tw"] == c["cfa"].shape[1]
                and c["th"] == c["cfa"].shape[0] and c["cfa"].shape[1] >= 128]


def _groups():
    """Export cases grouped by frame size: one multi-frame launch per group."""
    by = {}
    for c in EXPORT_CASES:
        by.setdefault(c["cfa"].shape, []).append(c)
    return sorted(by.items())


def test_fixture_holds_export_kernel_widths():
    assert {c["name"] for c in EXPORT_CASES} >= {"export_tile_5x134", "export_odd_4x131", "export_shift_3x250"}
    assert max(len(g) for _, g in _groups()) >= 4                     # different frames of one size, for one launch


@pytest.mark.parametrize("shape,group", _groups(), ids=[f"{s[1]}x{s[0]}" for s, _ in _groups()])
def test_batch_entry_reproduces_the_evaluated_shader(gpu_lib, refc, shape, group):
    """One multi-frame launch per frame size and surface format: every frame of the group (its own CFA and slider stack; white
    balance and matrix are per frame too) plus a repeat of the first one, so a group of one still launches two frames."""
    ra = gpu_lib
    h, w = shape
    cases = group + group[:1]
    n = len(cases)
    exp = [c["f32_pinned"] for c in cases]
    exp8 = [refc.pack_u8(e) for e in exp]
    want_hist = sum(refc.histogram(e).reshape(-1).astype(np.uint64) for e in exp8)
    for fmt, dt, ch, want in ((ra.FMT_RGBA_F32, np.uint32, 4, [bits(e) for e in exp]),
                              (ra.FMT_RGBA_F16, np.uint16, 4, [refc.pack_f16(e).view(np.uint16) for e in exp]),
                              (ra.FMT_RGBA_U8, np.uint8, 4, exp8), (ra.FMT_RGB_U8, np.uint8, 3, [e[..., :3] for e in exp8])):
        d_in = [DevBuf.from_array(c["cfa"]) for c in cases]
        d_out = [DevBuf(h * w * ra.BYTES_PER_PIXEL[fmt]) for _ in range(n)]
        d_hist = DevBuf(768 * 8)
        be = ra.BatchExporter(0, w, h, fmt, True)
        frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], [ra.EditParams(**c["params"]) for c in cases],
                                cases[0]["wb"], cases[0]["cm"])
        for f, c in zip(frames, cases):                               # rd_frame carries its own white balance and matrix
            f.wb_multipliers[:] = c["wb"]
            f.color_matrix[:] = c["cm"]
        be.develop(frames)
        be.histogram(d_hist.ptr)
        sync()
        for c, o, e in zip(cases, d_out, want):
            assert np.array_equal(o.to_array(dt, (h, w, ch)), e), (c["name"], fmt)
        assert np.array_equal(d_hist.to_array(np.uint64, (768,)), want_hist), fmt
        be.close()
        for b in d_in + d_out + [d_hist]:
            b.free()
