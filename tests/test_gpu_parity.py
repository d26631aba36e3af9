"""-m gpu: the HIP path (through the C ABI, via the RenderPipeline mirror) against the CPU oracle.

Bar (BASELINE.json north_star): f32 surface within 1 ULP per channel; integer / byte / index work
(CFA indexing, RGBA8 pack, f16 pack, histogram) bit-exact.  Because both sides implement the same
pinned operation order we additionally assert that the f32 surface is bit-identical.
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from tests.golden_util import load_golden
from tests.helpers import (CM_IDENTITY, CM_TEST, UI_RANGES, WB_DAYLIGHT, expected_taps, random_cfa,
                           random_params, ulp_diff)

pytestmark = pytest.mark.gpu
F = np.float32
ULP_TOL = 1   # north_star: "<= 1 ULP fp32 per channel"


def make_pipe(ra, cfa, params=None, wb=(1, 1, 1, 1), cm=CM_IDENTITY, zoom=None, pan=(0.0, 0.0), bl=0, image_id=7,
              math=0):
    h, w = cfa.shape
    ep = ra.EditParams(**(params or {}))
    p = ra.RenderPipeline.new(image_id, cfa.reshape(-1), w, h, ep, wb, cm)
    if zoom is not None:
        p.update_uniforms_with_zoom(ep, zoom, pan[0], pan[1])
    if bl:
        p.set_black_level(bl)
    if math:
        p.set_math_mode(math)
    return p


def oracle(refc, cfa, params=None, wb=(1, 1, 1, 1), cm=CM_IDENTITY, tw=None, th=None, zoom=1.0, pan=(0.0, 0.0), bl=0,
           math=0):
    u = refc.make_uniforms(params, wb, cm, zoom, pan[0], pan[1], bl, math)
    return refc.render_f32(cfa, u, tw, th, nthreads=8)


class force_map:
    """Route a full-resolution render through rd_develop_map instead of rd_develop_quads."""

    def __enter__(self):
        os.environ["RD_FORCE_MAP"] = "1"

    def __exit__(self, *a):
        os.environ.pop("RD_FORCE_MAP", None)


def check_all_surfaces(ra, refc, pipe, exp, tw=None, th=None):
    """f32 (<=1 ULP, and bit-identical), f16 / u8 / histogram bit-exact, for one pipeline state."""
    got, hist = pipe.render(tw, th, ra.FMT_RGBA_F32, with_histogram=True)
    assert got.shape == exp.shape
    assert ulp_diff(got, exp) <= ULP_TOL
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), "f32 surface not bit-identical"
    u8 = refc.pack_u8(exp)
    assert np.array_equal(hist, refc.histogram(u8))
    got8, hist8 = pipe.render(tw, th, ra.FMT_RGBA_U8, with_histogram=True)
    assert np.array_equal(got8, u8) and np.array_equal(hist8, hist)
    got16 = pipe.render(tw, th, ra.FMT_RGBA_F16)
    assert np.array_equal(got16.view(np.uint16), refc.pack_f16(exp).view(np.uint16))


# ---- committed golden vectors ----------------------------------------------------------------------------
GOLD = load_golden()


@pytest.mark.parametrize("case", GOLD, ids=[c["name"] for c in GOLD])
@pytest.mark.parametrize("kernel", ["auto", "map"])
def test_golden_vectors(gpu_lib, case, kernel):
    ra = gpu_lib
    pipe = make_pipe(ra, case["cfa"], case["params"], case["wb"], case["cm"], case["zoom"], case["pan"],
                     case["black_level"])
    ctx = force_map() if kernel == "map" else _null()
    with ctx:
        got, hist = pipe.render(case["tw"], case["th"], ra.FMT_RGBA_F32, with_histogram=True)
        got8 = pipe.render(case["tw"], case["th"], ra.FMT_RGBA_U8)
        got16 = pipe.render(case["tw"], case["th"], ra.FMT_RGBA_F16)
    assert ulp_diff(got, case["f32"]) <= ULP_TOL
    assert np.array_equal(got.view(np.uint32), case["f32"].view(np.uint32))
    assert np.array_equal(got8, case["u8"])
    assert np.array_equal(got16.view(np.uint16), case["f16"])
    assert np.array_equal(hist, case["hist"])


class _null:
    def __enter__(self):
        pass

    def __exit__(self, *a):
        pass


# ---- analytic KATs on the GPU -----------------------------------------------------------------------------
@pytest.mark.parametrize("level,rgba8", [(2048, 186), (0, 0), (4095, 255)])
def test_kat_flat_fields(gpu_lib, level, rgba8):
    ra = gpu_lib
    cfa = np.full((6, 8), level, np.uint16)
    pipe = make_pipe(ra, cfa)
    u8 = pipe.render_full_res_to_bytes().reshape(6, 8, 4)
    assert np.all(u8[..., :3] == rgba8) and np.all(u8[..., 3] == 255)


@pytest.mark.parametrize("h,w", [(4, 4), (5, 7), (1, 1), (1, 6), (7, 1), (2, 2), (3, 8), (6, 6), (8, 2)])
def test_kat_demosaic_taps(gpu_lib, refc, h, w):
    """Every tap and every edge clamp of shaders.rs:104-169, against the hand-written table."""
    ra = gpu_lib
    cfa = (np.arange(h * w, dtype=np.uint32) * 16 % 4096 + 3).astype(np.uint16).reshape(h, w)
    taps = expected_taps(cfa)
    L = refc.lib()

    def transfer(raw):
        v = F(raw) * F(1.0 / 4096.0)
        v = F(v - F(0.0)) / F(F(1.0) - F(0.0) + F(0.0001))
        return F(min(max(L.ref_powf(float(v), float(F(1.0 / 2.2)), 0), 0.0), 1.0))

    exp = np.vectorize(transfer)(taps).astype(F)
    pipe = make_pipe(ra, cfa)
    got = pipe.render()
    assert np.array_equal(got[..., :3].view(np.uint32), exp.view(np.uint32))
    assert np.all(got[..., 3] == 1.0)
    with force_map():
        assert np.array_equal(pipe.render().view(np.uint32), got.view(np.uint32))


def test_kat_matrix_transpose_and_negative_gamma(gpu_lib, refc):
    ra = gpu_lib
    cfa = random_cfa(np.random.default_rng(3), 8, 8)
    cm = [0.0] * 9
    cm[1] = 1.0
    check_all_surfaces(ra, refc, make_pipe(ra, cfa, None, (1, 1, 1, 1), cm), oracle(refc, cfa, None, (1, 1, 1, 1), cm))
    dark = np.full((4, 4), 100, np.uint16)
    got = make_pipe(ra, dark, {"blacks": 0.1}).render()
    assert np.all(got[..., :3] == 0.0) and not np.isnan(got).any()      # NaN -> clamp -> 0


# ---- random parity: sizes (even/odd/tiny), slider stacks, matrices, 16-bit range ---------------------------
SIZES = [(2, 2), (2, 4), (3, 5), (16, 24), (17, 24), (16, 25), (31, 33), (64, 96), (130, 258), (257, 514), (1, 64), (64, 2)]


@pytest.mark.parametrize("math", [0, 1], ids=["strict", "contracted"])
@pytest.mark.parametrize("h,w", SIZES)
def test_random_parity(gpu_lib, refc, h, w, math):
    ra = gpu_lib
    rng = np.random.default_rng([0x52415745, h, w])
    for trial in range(4):
        cfa = random_cfa(rng, h, w, 65536 if trial == 3 else 4096)
        params = random_params(rng) if trial else None
        cm = CM_TEST if trial % 2 else CM_IDENTITY
        pipe = make_pipe(ra, cfa, params, WB_DAYLIGHT, cm, math=math)
        check_all_surfaces(ra, refc, pipe, oracle(refc, cfa, params, WB_DAYLIGHT, cm, math=math))
        with force_map():                     # the general kernel in the same arithmetic
            got = pipe.render()
        assert np.array_equal(got.view(np.uint32), oracle(refc, cfa, params, WB_DAYLIGHT, cm, math=math).view(np.uint32))
        pipe.close()


def test_contracted_mode_edges_and_preview(gpu_lib, refc):
    """RD_MATH_CONTRACTED: range ends, degenerate stacks, zoom/pan map, 128 x 128k tiles (W % 128 == 0)."""
    ra = gpu_lib
    rng = np.random.default_rng(77)
    cfa = random_cfa(rng, 40, 256)
    for name, (lo, hi) in UI_RANGES.items():
        for v in (lo, hi):
            p = {name: float(F(v))}
            pipe = make_pipe(ra, cfa, p, WB_DAYLIGHT, CM_TEST, math=1)
            assert np.array_equal(pipe.render().view(np.uint32), oracle(refc, cfa, p, WB_DAYLIGHT, CM_TEST, math=1).view(np.uint32))
    for p in ({"exposure": 200.0}, {"whites": 0.0, "blacks": 0.0001}, {"whites": 0.2, "blacks": 0.2}, {"contrast": 500.0}):
        pipe = make_pipe(ra, cfa, p, WB_DAYLIGHT, CM_TEST, math=1)
        got = pipe.render()
        assert not np.isnan(got).any()
        assert np.array_equal(got.view(np.uint32), oracle(refc, cfa, p, WB_DAYLIGHT, CM_TEST, math=1).view(np.uint32))
    p = random_params(rng)
    pipe = make_pipe(ra, cfa, p, WB_DAYLIGHT, CM_TEST, zoom=1.7, pan=(0.05, -0.1), math=1)
    exp = oracle(refc, cfa, p, WB_DAYLIGHT, CM_TEST, tw=100, th=30, zoom=1.7, pan=(0.05, -0.1), math=1)
    check_all_surfaces(ra, refc, pipe, exp, 100, 30)


@pytest.mark.parametrize("name", list(UI_RANGES))
def test_each_slider_at_range_ends(gpu_lib, refc, name):
    ra = gpu_lib
    cfa = random_cfa(np.random.default_rng(11), 24, 32)
    for v in UI_RANGES[name]:
        p = {name: float(F(v))}
        pipe = make_pipe(ra, cfa, p, WB_DAYLIGHT, CM_TEST)
        check_all_surfaces(ra, refc, pipe, oracle(refc, cfa, p, WB_DAYLIGHT, CM_TEST))


def test_out_of_range_and_degenerate_params(gpu_lib, refc):
    """Beyond the UI ranges: huge exposure, whites == blacks (denominator 1e-4), negative / tiny wb,
    saturation -100 (greyscale) -- NaN/inf handling must agree with the oracle."""
    ra = gpu_lib
    cfa = random_cfa(np.random.default_rng(5), 20, 28, 65536)
    stacks = [
        ({"exposure": 20.0}, WB_DAYLIGHT), ({"exposure": -40.0}, WB_DAYLIGHT), ({"exposure": 200.0}, WB_DAYLIGHT),
        ({"whites": 0.2, "blacks": 0.2}, WB_DAYLIGHT), ({"whites": 0.0, "blacks": 0.5}, WB_DAYLIGHT),
        ({"whites": 0.0, "blacks": 0.0001}, WB_DAYLIGHT),          # denominator exactly 0: generic divide path
        ({"saturation": -100.0, "vibrance": 1.0}, WB_DAYLIGHT), ({"contrast": 500.0}, WB_DAYLIGHT),
        ({"temperature": 5.0, "tint": -5.0}, WB_DAYLIGHT), ({}, (-1.0, 1.0, 1e-38, 1.0)), ({}, (1e-42, 1e30, 0.0, 1.0)),
        ({"highlights": 50.0, "shadows": -50.0}, WB_DAYLIGHT),
    ]
    for params, wb in stacks:
        pipe = make_pipe(ra, cfa, params, wb, CM_TEST)
        got = pipe.render()
        exp = oracle(refc, cfa, params, wb, CM_TEST)
        assert not np.isnan(got).any()
        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (params, wb)


def test_identity_step_elision_is_exact(gpu_lib, refc):
    """The export kernel skips steps of the stack that are exact identities for a frame's uniforms (rd_uniforms.h RD_EL_*);
    the oracle never skips anything.  Untouched sliders in every combination with the identity matrix, zeros in the CFA
    (signed-zero paths), and the cases where the skip must NOT be taken because an intermediate can overflow
    (0 * inf is NaN in the literal evaluation): all bit-identical, on the export kernel and on the general one."""
    ra = gpu_lib
    rng = np.random.default_rng(23)
    cfa = random_cfa(rng, 36, 256, 65536)
    cfa[::5, ::3] = 0
    full = {"exposure": 0.7, "contrast": 5.0, "highlights": -0.4, "shadows": 0.3, "whites": 1.05, "blacks": 0.02,
            "vibrance": 0.4, "saturation": 25.0, "temperature": 0.25, "tint": -0.15}
    stacks = [({}, WB_DAYLIGHT, CM_IDENTITY), ({}, (1, 1, 1, 1), CM_IDENTITY), (full, WB_DAYLIGHT, CM_IDENTITY)]
    for drop in (("highlights",), ("shadows",), ("vibrance",), ("saturation",), ("temperature", "tint"), ("exposure",),
                 ("highlights", "shadows"), ("vibrance", "saturation"), ("exposure", "temperature", "tint", "vibrance")):
        p = {k: v for k, v in full.items() if k not in drop}
        stacks.append((p, WB_DAYLIGHT, CM_IDENTITY))
        stacks.append((p, WB_DAYLIGHT, CM_TEST))
    # overflow guards: the flags must stay clear, inf * 0 -> NaN -> 0 exactly as in the literal evaluation
    stacks += [({"exposure": 126.0}, WB_DAYLIGHT, CM_IDENTITY), ({"exposure": 200.0}, WB_DAYLIGHT, CM_IDENTITY),
               ({"exposure": 110.0, "contrast": 40.0}, WB_DAYLIGHT, CM_IDENTITY),
               ({}, (1e38, 1e38, 1e38, 1.0), CM_IDENTITY), ({"whites": 0.0, "blacks": 0.0001}, WB_DAYLIGHT, CM_IDENTITY),
               ({"whites": 0.2, "blacks": 0.2, "exposure": 90.0}, WB_DAYLIGHT, CM_IDENTITY),
               ({"exposure": 100.0}, WB_DAYLIGHT, (1e30, 0, 0, 0, 1e30, 0, 0, 0, 1e30))]
    # the levels divide: sliders for which the one-correction quotient is not proven (RD_EL_FIX clear: the two-correction
    # sequence with v_div_fixup runs), at its limits (|blacks| around 2^-50, den around 2^40), and numerators that cancel
    # to zero or to a few ulps (contrast -100 maps every pixel to 0.5; blacks = 0.5 then gives 0 / den)
    stacks += [({"blacks": 1e-20}, WB_DAYLIGHT, CM_IDENTITY), ({"blacks": -1e-30, "whites": 3e12}, WB_DAYLIGHT, CM_IDENTITY),
               ({"blacks": 1e-15}, WB_DAYLIGHT, CM_IDENTITY), ({"blacks": 8.8e-16}, WB_DAYLIGHT, CM_IDENTITY),
               ({"whites": 1.0e12}, WB_DAYLIGHT, CM_IDENTITY), ({"whites": 1.2e12}, WB_DAYLIGHT, CM_IDENTITY),
               ({"contrast": -100.0, "blacks": 0.5}, WB_DAYLIGHT, CM_IDENTITY),
               ({"contrast": -100.0, "blacks": 0.49999997}, WB_DAYLIGHT, CM_IDENTITY),
               ({"contrast": -99.99999, "blacks": 0.5, "whites": 0.5001}, WB_DAYLIGHT, CM_IDENTITY),
               ({"blacks": 0.3, "whites": 0.2999}, WB_DAYLIGHT, CM_TEST), ({"blacks": -3.0, "whites": 37.0}, WB_DAYLIGHT, CM_TEST)]
    for math in (0, 1):
        for params, wb, cm in stacks:
            exp = oracle(refc, cfa, params, wb, cm, math=math)
            pipe = make_pipe(ra, cfa, params, wb, cm, math=math)
            got, hist = pipe.render(None, None, ra.FMT_RGBA_F32, with_histogram=True)
            assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), (math, params, wb, cm)
            assert np.array_equal(hist, refc.histogram(refc.pack_u8(exp)))
            with force_map():
                got_map = pipe.render()
            assert np.array_equal(got_map.view(np.uint32), exp.view(np.uint32)), ("map", math, params)


def test_channel_separable_stacks(gpu_lib, refc):
    """Stacks whose channel-mixing steps are all identities (RD_EL_SEPARABLE: identity matrix, highlights = shadows =
    vibrance = saturation = 0 -- the usual edit) take rd_colour_separable: five values per 2x2 block instead of nine.
    Every surface format and the fused histogram must stay bit-identical to the oracle, which never skips anything: both
    arithmetic modes, full and ragged tiles (256 / 130 / 6 columns), 1- and 2-row frames, a black level, zeros and
    saturated samples in the CFA, levels sliders on both divide paths; and a stack one slider away from separable."""
    ra = gpu_lib
    rng = np.random.default_rng(77)
    sep = [{}, {"exposure": 0.7, "contrast": 5.0, "whites": 1.05, "blacks": 0.02}, {"exposure": -1.3, "temperature": 0.25, "tint": -0.15},
           {"contrast": -100.0, "blacks": 0.5}, {"blacks": 1e-20, "whites": 0.9}, {"temperature": -1.0, "tint": 1.0, "contrast": 100.0},
           {"exposure": 5.0, "blacks": -0.2, "whites": 0.4}]
    for s in sep:
        assert ra.elided_steps(ra.EditParams(**s), WB_DAYLIGHT, CM_IDENTITY, 0) & 0x7a == 0x7a, s     # MAT | HL | SH | SAT | VIB
    almost = [{"exposure": 0.7, "vibrance": 1e-6}, {"contrast": 5.0, "saturation": 1e-4}, {"highlights": 1e-7}, {"shadows": -1e-7}]
    for (h, w) in ((36, 256), (5, 130), (2, 6), (1, 256), (7, 384)):
        cfa = random_cfa(rng, h, w, 65536)
        cfa[::3, ::5] = 0
        cfa[1::4, 2::7] = 65535
        for math in (0, 1):
            for params in sep + almost:
                for bl in (0, 600):
                    if bl and params is not sep[1]:
                        continue
                    pipe = make_pipe(ra, cfa, params, WB_DAYLIGHT, CM_IDENTITY, bl=bl, math=math)
                    exp = oracle(refc, cfa, params, WB_DAYLIGHT, CM_IDENTITY, bl=bl, math=math)
                    check_all_surfaces(ra, refc, pipe, exp)
                    rgb, hist = pipe.render(fmt=ra.FMT_RGB_U8, with_histogram=True)      # export kernel from 128 px up, else the map kernel
                    u8 = refc.pack_u8(exp)
                    assert np.array_equal(rgb, u8[..., :3]) and np.array_equal(hist, refc.histogram(u8)), (params, math)
    # the camera matrix switches the path off again
    cfa = random_cfa(rng, 12, 256, 65536)
    pipe = make_pipe(ra, cfa, sep[1], WB_DAYLIGHT, CM_TEST)
    check_all_surfaces(ra, refc, pipe, oracle(refc, cfa, sep[1], WB_DAYLIGHT, CM_TEST))


def test_black_level_extension(gpu_lib, refc):
    ra = gpu_lib
    cfa = random_cfa(np.random.default_rng(9), 18, 26)
    for bl in (1, 64, 600, 5000):
        pipe = make_pipe(ra, cfa, None, WB_DAYLIGHT, CM_IDENTITY, bl=bl)
        check_all_surfaces(ra, refc, pipe, oracle(refc, cfa, None, WB_DAYLIGHT, CM_IDENTITY, bl=bl))


# ---- the reference's RenderPipeline surface ----------------------------------------------------------------
def test_render_pipeline_surface(gpu_lib, refc):
    """new / fields / dimensions / update_uniforms[_with_zoom] / the three renders / calculate_histogram
    (pipeline.rs:114-736) on a 300x200 frame; preview 300x200 (w < 1280), histogram 128x85."""
    ra = gpu_lib
    rng = np.random.default_rng(21)
    h, w = 200, 300
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = make_pipe(ra, cfa, params, WB_DAYLIGHT, CM_IDENTITY, image_id=42)
    assert pipe.dimensions() == (w, h) and pipe.image_id == 42
    assert (pipe.preview_width, pipe.preview_height, pipe.histogram_width, pipe.histogram_height) == refc.derived_dims(w, h)
    full = pipe.render_full_res_to_bytes()
    assert full.dtype == np.uint8 and full.size == w * h * 4
    assert np.array_equal(full.reshape(h, w, 4), refc.pack_u8(oracle(refc, cfa, params, WB_DAYLIGHT)))
    hb = pipe.render_to_histogram_bytes()
    exp_h = refc.pack_u8(oracle(refc, cfa, params, WB_DAYLIGHT, tw=pipe.histogram_width, th=pipe.histogram_height))
    assert np.array_equal(hb.reshape(exp_h.shape), exp_h)
    hist = pipe.calculate_histogram(hb)
    assert hist.shape == (3, 256) and np.array_equal(hist, refc.histogram(exp_h))
    assert hist.sum() == 3 * pipe.histogram_width * pipe.histogram_height
    # slider drag + zoom/pan (main.rs:1515): preview with zoom 2, pan
    p2 = ra.EditParams(**random_params(rng))
    pipe.update_uniforms_with_zoom(p2, 2.0, 0.1, -0.05)
    prev = pipe.render_to_bytes()
    p2d = {f: getattr(p2, f) for f in ra.FIELDS}
    exp_p = refc.pack_u8(oracle(refc, cfa, p2d, WB_DAYLIGHT, tw=pipe.preview_width, th=pipe.preview_height,
                                zoom=2.0, pan=(0.1, -0.05)))
    assert np.array_equal(prev.reshape(exp_p.shape), exp_p)
    # reference quirk: export re-uses whatever zoom/pan the last view() wrote (main.rs:1515 vs :1754)
    full_z = pipe.render_full_res_to_bytes()
    exp_z = refc.pack_u8(oracle(refc, cfa, p2d, WB_DAYLIGHT, zoom=2.0, pan=(0.1, -0.05)))
    assert np.array_equal(full_z.reshape(exp_z.shape), exp_z)
    # update_uniforms resets zoom/pan to 1/0/0 (pipeline.rs:367-369)
    pipe.update_uniforms(p2)
    assert np.array_equal(pipe.render_full_res_to_bytes().reshape(h, w, 4), refc.pack_u8(oracle(refc, cfa, p2d, WB_DAYLIGHT)))
    # zoomed-out: out-of-bounds fragments are (0,0,0,255) and count in bin 0
    pipe.update_uniforms_with_zoom(p2, 0.5, 0.0, 0.0)
    hb = pipe.render_to_histogram_bytes().reshape(pipe.histogram_height, pipe.histogram_width, 4)
    assert tuple(hb[0, 0]) == (0, 0, 0, 255)
    exp_h = refc.pack_u8(oracle(refc, cfa, p2d, WB_DAYLIGHT, tw=pipe.histogram_width, th=pipe.histogram_height, zoom=0.5))
    assert np.array_equal(hb, exp_h)
    assert np.array_equal(pipe.calculate_histogram(hb), refc.histogram(exp_h))


def test_preview_dims_for_a_wide_frame(gpu_lib, refc):
    ra = gpu_lib
    h, w = 100, 1504            # w > 1280: preview is 1280 x trunc(1280/aspect)
    cfa = random_cfa(np.random.default_rng(2), h, w)
    pipe = make_pipe(ra, cfa, None, WB_DAYLIGHT)
    assert (pipe.preview_width, pipe.preview_height) == (1280, refc.derived_dims(w, h)[1])
    prev = pipe.render_to_bytes()
    exp = refc.pack_u8(oracle(refc, cfa, None, WB_DAYLIGHT, tw=pipe.preview_width, th=pipe.preview_height))
    assert np.array_equal(prev.reshape(exp.shape), exp)


def test_calculate_histogram_edge_inputs(gpu_lib):
    ra = gpu_lib
    pipe = make_pipe(ra, np.zeros((2, 2), np.uint16))
    assert pipe.calculate_histogram(np.zeros(0, np.uint8)).sum() == 0
    flat = np.tile(np.array([7, 7, 7, 255], np.uint8), 100000)     # every pixel in one bin
    h = pipe.calculate_histogram(flat)
    assert np.all(h[:, 7] == 100000) and h.sum() == 300000
    rnd = np.random.default_rng(1).integers(0, 256, 4 * 77777 + 3, dtype=np.uint8)   # ragged tail ignored
    h = pipe.calculate_histogram(rnd)
    px = rnd[: 4 * 77777].reshape(-1, 4)
    assert all(np.array_equal(h[c], np.bincount(px[:, c], minlength=256)) for c in range(3))


def test_errors_and_threads(gpu_lib):
    ra = gpu_lib
    from raweditor_amd import _lib
    import ctypes as C
    cfa = random_cfa(np.random.default_rng(4), 64, 64)
    pipe = make_pipe(ra, cfa, None, WB_DAYLIGHT)
    small = np.empty(10, np.uint8)
    assert _lib.lib().rd_render_full_res_to_bytes(pipe._h, small.ctypes.data_as(C.c_void_p), small.size) == -1
    assert b"dst_len" in _lib.lib().rd_last_error()
    with pytest.raises(ra.RawdevError):
        pipe.render(0, 5)
    with pytest.raises(ra.RawdevError):
        pipe.render(4, 4, fmt=9)
    # Arc<RenderPipeline> is shared by the UI thread and the export thread (main.rs:1054, :1749)
    ref8 = pipe.render_full_res_to_bytes()
    bad = []

    def work():
        for _ in range(10):
            if not np.array_equal(pipe.render_full_res_to_bytes(), ref8):
                bad.append(1)
            pipe.render_to_histogram_bytes()

    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


# ---- full-size frames (BASELINE.json: 6016 x 4016): properties + sampled oracle rows ---------------------------
def test_full_size_24mp(gpu_lib, refc):
    ra = gpu_lib
    h, w = 4016, 6016
    rng = np.random.default_rng(0x52415745)
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = make_pipe(ra, cfa, params, WB_DAYLIGHT, CM_TEST)
    got, hist = pipe.render(fmt=ra.FMT_RGBA_F32, with_histogram=True)
    # (1) oracle on sampled row bands: first/last rows, both parities, the middle
    u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST)
    for r0, r1 in ((0, 6), (1001, 1007), (2006, 2012), (h - 6, h)):
        exp = refc.render_band(cfa, u, r0, r1)
        assert ulp_diff(got[r0:r1], exp) <= ULP_TOL
        assert np.array_equal(got[r0:r1].view(np.uint32), exp.view(np.uint32))
    # (2) histogram: checksum of checksums + independent numpy bincount of the U8 surface
    got8, hist8 = pipe.render(fmt=ra.FMT_RGBA_U8, with_histogram=True)
    assert hist.sum() == 3 * h * w and np.array_equal(hist, hist8)
    flat8 = got8.reshape(-1, 4)
    assert all(np.array_equal(hist[c], np.bincount(flat8[:, c], minlength=256)) for c in range(3))
    assert np.all(flat8[:, 3] == 255)
    assert np.array_equal(got8, refc.pack_u8(got))            # u8 surface == pack of the f32 surface
    assert np.array_equal(pipe.calculate_histogram(got8), hist)
    del got8, flat8
    # (3) linearity: doubling wb, or doubling every CFA sample, is bit-identical to one more stop of exposure
    cfa_half = (cfa >> 1).astype(np.uint16)
    p0 = dict(params, exposure=float(F(min(params["exposure"], 3.0))))
    p1 = dict(p0, exposure=float(F(p0["exposure"] + 1.0)))
    ref_lin = make_pipe(ra, cfa_half, p1, WB_DAYLIGHT, CM_TEST).render()
    assert np.array_equal(make_pipe(ra, cfa_half, p0, tuple(2 * x for x in WB_DAYLIGHT), CM_TEST).render().view(np.uint32),
                          ref_lin.view(np.uint32))
    assert np.array_equal(make_pipe(ra, (cfa_half * 2).astype(np.uint16), p0, WB_DAYLIGHT, CM_TEST).render().view(np.uint32),
                          ref_lin.view(np.uint32))
    del ref_lin
    # (4) the export kernel and the general map kernel agree on every pixel of the full frame
    with force_map():
        got_map = pipe.render(fmt=ra.FMT_RGBA_F32)
    assert np.array_equal(got_map.view(np.uint32), got.view(np.uint32))


# ---- BASELINE config 5 shape: 100 MP frame (11648 x 8736), f16 surface, tiled multi-launch (row bands) ----
def test_full_size_100mp_f16_row_bands(gpu_lib, refc):
    """wgpu's 8192-px texture limit (pipeline.rs:164) would reject this frame; linear buffers do not care.
    One pipeline render (single launch) and the batch path in 8 row bands must both match the oracle on
    sampled bands (first / last rows, a band seam, the middle) and agree with each other everywhere."""
    from tests.gpu_util import DevBuf, sync
    ra = gpu_lib
    h, w = 8736, 11648
    rng = np.random.default_rng(100)
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST)
    d_in = DevBuf.from_array(cfa)
    d_out = DevBuf(h * w * 8)
    d_hist = DevBuf(768 * 8)
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F16, True)
    fr = be.make_frames([d_in.ptr], [d_out.ptr], [ra.EditParams(**params)], WB_DAYLIGHT, CM_TEST)
    be.develop(fr, row_bands=8)
    be.histogram(d_hist.ptr)
    sync()
    got = d_out.to_array(np.uint16, (h, w, 4))
    units = h // 2 + 1
    seam = 2 * (units * 3 // 8) - 1                      # first row of band 3
    for r0, r1 in ((0, 4), (seam - 3, seam + 3), (h // 2, h // 2 + 4), (h - 4, h)):
        exp = refc.pack_f16(refc.render_band(cfa, u, r0, r1)).view(np.uint16)
        assert np.array_equal(got[r0:r1], exp), (r0, r1)
    hist = d_hist.to_array(np.uint64, (768,))
    assert int(hist.sum()) == 3 * h * w
    be.develop(fr, row_bands=1)                          # one launch: same surface, same histogram
    be.histogram(d_hist.ptr)
    sync()
    assert np.array_equal(d_out.to_array(np.uint16, (h, w, 4)), got)
    assert np.array_equal(d_hist.to_array(np.uint64, (768,)), hist)
    be.close()


def test_surface_beyond_4_gib(gpu_lib, refc):
    """Maximum sizes: a 320 MP frame (20096 x 16000: a stitched panorama; 643 MB of CFA) whose f32 surface is 5.1 GB, so pixel
    and byte offsets pass 2^32 -- rendered into device memory by the export kernel (one launch, 1.26 M tiles) and by the map
    kernel, row bands from both sides of the 4 GiB line against the oracle; the histogram counts every pixel.  Then the
    host read-back of a surface above 4 GiB (band-pipelined, page-locked destination) on the rows around the line."""
    from tests.gpu_util import DevBuf
    ra = gpu_lib
    h, w = 16000, 20096
    free, total = C.c_size_t(), C.c_size_t()
    ra._lib.check(ra._lib.lib().rd_device_memory(0, C.byref(free), C.byref(total)))
    if free.value < 14 * (1 << 30):
        pytest.skip("needs 14 GiB of free HBM")
    rng = np.random.default_rng(0x3434)
    cfa = random_cfa(rng, h, w)
    params = random_params(rng)
    pipe = make_pipe(ra, cfa, params, WB_DAYLIGHT, CM_TEST)
    u = refc.make_uniforms(params, WB_DAYLIGHT, CM_TEST)
    row_bytes = w * 16
    line = (1 << 32) // row_bytes                                # the row that holds byte 2^32
    bands = ((0, 4), (line - 3, line + 5), (h // 2 + 1, h // 2 + 5), (h - 4, h))
    exp = {b: refc.render_band(cfa, u, b[0], b[1]) for b in bands}
    surf = DevBuf(h * row_bytes)
    hist_dev = DevBuf(768 * 4)
    assert surf.nbytes > (1 << 32)
    for kernel in ("quads", "map"):
        ctx = force_map() if kernel == "map" else None
        if ctx:
            ctx.__enter__()
        try:
            pipe.render_device(w, h, ra.FMT_RGBA_F32, surf.ptr, hist_dev.ptr)
            ra._lib.check(ra._lib.lib().rd_device_synchronize(0))
        finally:
            if ctx:
                ctx.__exit__(None, None, None)
        for r0, r1 in bands:
            got = np.empty((r1 - r0, w, 4), np.float32)
            ra._lib.check(ra._lib.lib().rd_memcpy_d2h(0, got.ctypes.data_as(C.c_void_p), C.c_void_p(surf.ptr + r0 * row_bytes), got.nbytes))
            assert np.array_equal(got.view(np.uint32), exp[(r0, r1)].view(np.uint32)), (kernel, r0)
        hist = hist_dev.to_array(np.uint32, (3, 256))
        assert hist.sum(axis=1).tolist() == [h * w] * 3, kernel
    surf.free(); hist_dev.free()
    # host read-back above 4 GiB: the f32 surface into page-locked memory, bands around the line and at both ends
    pin = ra.PinnedBytes(h * row_bytes)
    out = pin.array.view(np.float32).reshape(h, w, 4)
    ra._lib.check(ra._lib.lib().rd_render(pipe._h, w, h, ra.FMT_RGBA_F32, C.c_void_p(pin.ptr), pin.nbytes, None))
    for r0, r1 in bands:
        assert np.array_equal(out[r0:r1].view(np.uint32), exp[(r0, r1)].view(np.uint32)), ("host", r0)
    pin.free()
    pipe.close()


def test_matrix_layout_option(gpu_lib, refc, rng):
    """rd_options.matrix_layout of SURVEY.md section 8b: the default consumes the host's rows as COLUMNS like the reference
    (shaders.rs:209-214, quirk D4); RD_MATRIX_ROW_MAJOR applies the matrix as written.  Row-major with M must equal the
    reference layout with M^T bit for bit, on the pipeline and on the batch path, and must equal the oracle fed M^T."""
    from tests.gpu_util import DevBuf, sync
    ra = gpu_lib
    h, w = 130, 256
    cfa = random_cfa(rng, h, w)
    params = dict(exposure=-1.0, contrast=3.0, saturation=20.0, vibrance=0.3, whites=1.1, blacks=0.02)   # nothing clips wholesale
    m = CM_TEST
    mt = tuple(m[3 * c + r] for r in range(3) for c in range(3))
    exp_ref = oracle(refc, cfa, params, WB_DAYLIGHT, m)
    exp_row = oracle(refc, cfa, params, WB_DAYLIGHT, mt)
    assert not np.array_equal(exp_ref, exp_row)
    pipe = make_pipe(ra, cfa, params, WB_DAYLIGHT, m)
    assert np.array_equal(pipe.render().view(np.uint32), exp_ref.view(np.uint32))
    pipe.set_matrix_layout(ra.MATRIX_ROW_MAJOR)
    assert np.array_equal(pipe.render().view(np.uint32), exp_row.view(np.uint32))
    with pytest.raises(ra.RawdevError):
        pipe.set_matrix_layout(7)
    d_in, d_out = DevBuf.from_array(cfa), [DevBuf(h * w * 16), DevBuf(h * w * 16)]
    be = ra.BatchExporter(0, w, h, ra.FMT_RGBA_F32, False)
    fr = be.make_frames([d_in.ptr, d_in.ptr], [d_out[0].ptr, d_out[1].ptr], [ra.EditParams(**params)] * 2, WB_DAYLIGHT, m)
    fr[1].matrix_layout = ra.MATRIX_ROW_MAJOR
    be.develop(fr)
    sync()
    assert np.array_equal(d_out[0].to_array(np.float32, (h, w, 4)).view(np.uint32), exp_ref.view(np.uint32))
    assert np.array_equal(d_out[1].to_array(np.float32, (h, w, 4)).view(np.uint32), exp_row.view(np.uint32))
    fr[1].matrix_layout = 9
    with pytest.raises(ra.RawdevError):
        be.develop(fr)
    be.close()


def test_differential_fuzz_sample(gpu_lib):
    """400 draws of tests/fuzz_parity.py (random sizes, sliders on and off and far outside the UI, matrices, black levels,
    both arithmetic modes; 8 renders per frame against the oracle).  profiles/r02_fuzz_parity.txt holds a 200 000-frame run."""
    from tests import fuzz_parity as fz
    rng = np.random.default_rng([0x52415745, 99])
    for i in range(400):
        case = fz.draw_case(rng)
        bad, _ = fz.check(case)
        assert not bad, (i, bad, case[0].shape, case[1:])
    for i in range(25):                                          # rd_batch_develop: heterogeneous frames, random launch shapes
        bad, _ = fz.check_batch(rng)
        assert not bad, (i, bad)
