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


# ---- whole frames: BASELINE's sizes against the checksums of the evaluated shader text ----------------------------------------
from tests.test_wgsl_pin_cpu import FULL, fullsize_check, fullsize_inputs  # noqa: E402


@pytest.mark.parametrize("fr", FULL["frames"], ids=[f["name"] for f in FULL["frames"]])
def test_pipeline_reproduces_the_evaluated_shader_at_full_size(gpu_lib, fr):
    """24 MP (aligned, ragged and odd width) and 100 MP frames through the RenderPipeline mirror: the f32 surface (band CRCs, then
    SHA-256), the fused histogram, the RGBA8 bytes of render_full_res_to_bytes and the binary16 surface against
    tests/golden/wgsl_fullsize.json -- checksums of what the reference's shader text evaluates to (tools/make_wgsl_fullsize.py)."""
    ra = gpu_lib
    cfa = fullsize_inputs(fr)
    pipe = make_pipe(ra, cfa, fr["params"], FULL["wb"], fr["cm"])
    got, hist = pipe.render(fmt=ra.FMT_RGBA_F32, with_histogram=True)
    fullsize_check(fr, f32=got, hist=hist)
    del got
    fullsize_check(fr, rgba8=pipe.render_full_res_to_bytes().reshape(fr["h"], fr["w"], 4))
    f16, hist16 = pipe.render(fmt=ra.FMT_RGBA_F16, with_histogram=True)
    fullsize_check(fr, f16=f16, hist=hist16)
    pipe.close()


def _full_groups():
    by = {}
    for fr in FULL["frames"]:
        by.setdefault((fr["w"], fr["h"]), []).append(fr)
    return sorted(by.items())


@pytest.mark.parametrize("size,group", _full_groups(), ids=[f"{s[0]}x{s[1]}" for s, _ in _full_groups()])
@pytest.mark.parametrize("bands", [1, 8])
def test_batch_entry_reproduces_the_evaluated_shader_at_full_size(gpu_lib, size, group, bands):
    """The batch entry on the same frames: the frames of one size in one call -- multi-frame launches (bands = 1, the default)
    and BASELINE config 5's wording, eight row-band launches per frame -- f32 for the 24 MP sizes, binary16 for the 100 MP
    frame (config 5's surface), RGBA8 and RGB8 for all, with the accumulated histogram."""
    ra = gpu_lib
    w, h = size
    cfas = [fullsize_inputs(fr) for fr in group]
    d_in = [DevBuf.from_array(c) for c in cfas]
    want_hist = sum(np.asarray(fr["histogram"], np.uint64) for fr in group)
    wide = ra.FMT_RGBA_F16 if w * h > 30_000_000 else ra.FMT_RGBA_F32
    for fmt in (wide, ra.FMT_RGBA_U8, ra.FMT_RGB_U8):
        bpp = ra.BYTES_PER_PIXEL[fmt]
        d_out = [DevBuf(h * w * bpp) for _ in group]
        d_hist = DevBuf(768 * 8)
        be = ra.BatchExporter(0, w, h, fmt, True)
        frames = be.make_frames([b.ptr for b in d_in], [b.ptr for b in d_out], [ra.EditParams(**fr["params"]) for fr in group],
                                FULL["wb"], group[0]["cm"])
        for f, fr in zip(frames, group):
            f.color_matrix[:] = fr["cm"]
        be.develop(frames, row_bands=bands)
        be.histogram(d_hist.ptr)
        sync()
        for fr, o in zip(group, d_out):
            if fmt == ra.FMT_RGBA_F32:
                fullsize_check(fr, f32=o.to_array(np.float32, (h, w, 4)))
            elif fmt == ra.FMT_RGBA_F16:
                fullsize_check(fr, f16=o.to_array(np.uint16, (h, w, 4)))
            elif fmt == ra.FMT_RGBA_U8:
                fullsize_check(fr, rgba8=o.to_array(np.uint8, (h, w, 4)))
            else:                                                    # RGB8 = the RGBA8 bytes without their alpha
                rgb = o.to_array(np.uint8, (h, w, 3))
                rgba = np.full((h, w, 4), 255, np.uint8)
                rgba[..., :3] = rgb
                fullsize_check(fr, rgba8=rgba)
        assert np.array_equal(d_hist.to_array(np.uint64, (768,)), want_hist), (size, fmt)
        be.close()
        for b in d_out + [d_hist]:
            b.free()
    for b in d_in:
        b.free()
