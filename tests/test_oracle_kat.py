"""CPU tests: the oracle against the analytic known-answer vectors K1-K10 of SURVEY.md section 8c
(hand-derived from the reference shader text, src/gpu/shaders.rs:104-267), and the two oracle
twins (C, numpy) against each other.  The reference ships no fixtures for this path, so these
KATs are the pin ("parity unpinned" against the reference's own outputs: oracle/develop_ref.h)."""
import numpy as np
import pytest

from oracle import develop_np as dn
from tests.helpers import (CM_IDENTITY, CM_TEST, WB_DAYLIGHT, expected_taps, random_cfa,
                           random_params, ulp_diff)

F = np.float32


def c_render(refc, cfa, params=None, wb=(1, 1, 1, 1), cm=CM_IDENTITY, tw=None, th=None, zoom=1.0,
             pan=(0.0, 0.0), pow_mode=0, black_level=0):
    u = refc.make_uniforms(params, wb, cm, zoom, pan[0], pan[1], black_level)
    return refc.render_f32(cfa, u, tw, th, pow_mode)


def np_render(cfa, params=None, wb=(1, 1, 1, 1), cm=CM_IDENTITY, tw=None, th=None, zoom=1.0,
              pan=(0.0, 0.0), pow_mode="pinned", black_level=0):
    u = dn.Uniforms(**(params or {}), wb=tuple(wb), cm=tuple(cm), zoom=zoom, pan_x=pan[0], pan_y=pan[1],
                    black_level=black_level)
    return dn.render_f32(cfa, u, tw, th, pow_mode)


def default_transfer(refc, raw):
    """Default-params per-channel transfer: v = raw/4096 -> /1.0001 -> gamma -> clamp.  With the
    default stack every other step of shaders.rs:192-257 is an exact identity."""
    v = F(raw) * F(1.0 / 4096.0)
    v = F(v - F(0.0)) / F(F(1.0) - F(0.0) + F(0.0001))
    out = refc.lib().ref_powf(float(v), float(F(1.0 / 2.2)), 0)
    return min(max(out, 0.0), 1.0)


# ---- K1-K3: flat fields -------------------------------------------------------------------------
@pytest.mark.parametrize("level,rgba8", [(2048, 186), (0, 0), (4095, 255)])
def test_k1_k3_flat_fields(refc, level, rgba8):
    cfa = np.full((6, 8), level, np.uint16)
    out = c_render(refc, cfa)
    assert np.all(out[..., 3] == 1.0)
    u8 = refc.pack_u8(out)
    assert np.all(u8[..., :3] == rgba8) and np.all(u8[..., 3] == 255)
    if level == 2048:
        assert abs(float(out[0, 0, 0]) - 0.5 ** 0 * (0.5 / 1.0001) ** (1 / 2.2)) < 2e-7
    assert ulp_diff(out, np_render(cfa)) == 0


# ---- K4: RGGB colour sites incl. the green-for-blue quirk and the row-0 clamp -------------------------
def test_k4_rggb_pattern(refc):
    h, w = 6, 6
    cfa = np.zeros((h, w), np.uint16)
    cfa[0::2, 0::2] = 1000  # R sites
    cfa[0::2, 1::2] = 2000  # G sites
    cfa[1::2, 0::2] = 2000  # G sites
    cfa[1::2, 1::2] = 3000  # B sites
    out = c_render(refc, cfa)
    f = {v: default_transfer(refc, v) for v in (1000, 2000, 3000)}
    # interior R site (2,2): blue is mis-sampled from the green site above (shaders.rs:146)
    assert tuple(out[2, 2, :3]) == (F(f[1000]), F(f[2000]), F(f[2000]))
    # interior G site on a red row (2,3): r = left, b = above (a true B site)
    assert tuple(out[2, 3, :3]) == (F(f[1000]), F(f[2000]), F(f[3000]))
    # G site on a blue row (3,2) and B site (3,3): (R below, G, B)
    assert tuple(out[3, 2, :3]) == (F(f[1000]), F(f[2000]), F(f[3000]))
    assert tuple(out[3, 3, :3]) == (F(f[1000]), F(f[2000]), F(f[3000]))
    # row 0: the "above" tap clamps onto the pixel itself (shaders.rs:163-166)
    assert tuple(out[0, 0, :3]) == (F(f[1000]), F(f[2000]), F(f[1000]))
    assert tuple(out[0, 1, :3]) == (F(f[1000]), F(f[2000]), F(f[2000]))


# ---- K5: per-site-unique planes pin every tap and every clamp, odd sizes included -------------------
@pytest.mark.parametrize("h,w", [(4, 4), (5, 7), (1, 1), (1, 6), (7, 1), (2, 2), (3, 8)])
def test_k5_unique_sites(refc, h, w):
    cfa = (np.arange(h * w, dtype=np.uint32) * 16 % 4096 + 3).astype(np.uint16).reshape(h, w)
    taps = expected_taps(cfa)
    out = c_render(refc, cfa)
    exp = np.vectorize(lambda v: F(default_transfer(refc, int(v))))(taps).astype(F)
    assert ulp_diff(out[..., :3], exp) == 0
    assert ulp_diff(out, np_render(cfa)) == 0


# ---- K6: the matrix rows are consumed as columns (shaders.rs:209-214) ---------------------------------
def test_k6_matrix_transpose(refc):
    cfa = np.zeros((4, 4), np.uint16)
    cfa[0::2, 0::2] = 3000
    cfa[0::2, 1::2] = 1000
    cfa[1::2, 0::2] = 1000
    cfa[1::2, 1::2] = 500
    cm = [0.0] * 9
    cm[1] = 1.0          # host row 0, col 1
    out = c_render(refc, cfa, cm=cm)
    # pixel (2,2) is an R site: r=3000, so out.y (green) receives r and out.x, out.z are 0
    assert out[2, 2, 1] == F(default_transfer(refc, 3000))
    assert out[2, 2, 0] == 0.0 and out[2, 2, 2] == 0.0


# ---- K7: negative into gamma -> NaN -> clamp -> 0 (shaders.rs:239, :261-264) ------------------------
def test_k7_negative_into_gamma(refc):
    cfa = np.full((4, 4), 100, np.uint16)
    out = c_render(refc, cfa, {"blacks": 0.1})
    assert np.all(out[..., :3] == 0.0) and not np.isnan(out).any()
    assert ulp_diff(out, np_render(cfa, {"blacks": float(F(0.1))})) == 0


# ---- K8: sliders at their range ends; exposure +1 doubles exactly ---------------------------------------
def test_k8_exposure_doubles_exactly(refc):
    L = refc.lib()
    assert L.ref_powf(2.0, 1.0, 0) == 2.0 and L.ref_powf(2.0, 0.0, 0) == 1.0
    assert L.ref_powf(2.0, -3.0, 0) == 0.125 and L.ref_powf(2.0, 5.0, 0) == 32.0
    assert L.ref_powf(1.0, float(F(1 / 2.2)), 0) == 1.0 and L.ref_powf(0.0, 0.4545, 0) == 0.0
    assert np.isnan(L.ref_powf(-1.0, 0.4545, 0))


@pytest.mark.parametrize("name", ["exposure", "contrast", "highlights", "shadows", "whites", "blacks",
                                  "vibrance", "saturation", "temperature", "tint"])
def test_k8_slider_range_ends(refc, name):
    from tests.helpers import UI_RANGES
    cfa = np.tile(np.array([[200, 1800, 3900, 900]], np.uint16), (4, 2))
    for v in UI_RANGES[name]:
        p = {name: float(F(v))}
        a = c_render(refc, cfa, p, WB_DAYLIGHT, CM_TEST)
        b = np_render(cfa, p, WB_DAYLIGHT, CM_TEST)
        assert ulp_diff(a, b) == 0
        assert np.all((a >= 0) & (a <= 1))


# ---- K9: the point-sampling map (shaders.rs:23-60, :184-187) ---------------------------------------------
def test_k9_preview_map(refc):
    w, h, tw, th = 47 * 4, 12, 40, 5     # scale 4.7 like 6016 -> 1280
    cfa = (np.arange(w * h) % 4096).astype(np.uint16).reshape(h, w)
    PX, PY, inside = dn.pixel_map(w, h, tw, th, 1.0, 0.0, 0.0)
    assert inside.all()
    assert np.array_equal(PX[0], [int((i + 0.5) * 4.7) for i in range(tw)])
    assert np.array_equal(PY[:, 0], [int((j + 0.5) * h / th) for j in range(th)])
    for zoom, pan in [(2.0, (0.1, -0.05)), (0.5, (0.0, 0.0)), (3.0, (-0.3, 0.3))]:
        a = c_render(refc, cfa, None, tw=tw, th=th, zoom=zoom, pan=pan)
        b = np_render(cfa, None, tw=tw, th=th, zoom=zoom, pan=pan)
        assert ulp_diff(a, b) == 0
    a = c_render(refc, cfa, None, tw=tw, th=th, zoom=0.5)
    assert tuple(a[0, 0]) == (0.0, 0.0, 0.0, 1.0)      # zoomed out: black border, alpha 1 (:174-178)
    assert a[th // 2, tw // 2, :3].any()


def test_k9_export_map_is_identity():
    for n in (6016, 4016, 11648, 8736, 1, 2, 3, 1280, 854):
        px, _, inside = dn.pixel_map(n, 1, n, 1, 1.0, 0.0, 0.0)
        assert inside.all() and np.array_equal(px[0], np.arange(n))


# ---- K10: histogram of K1 at the reference's 128 x 85 histogram target ----------------------------------
def test_k10_histogram(refc):
    assert refc.derived_dims(6016, 4016) == (1280, 854, 128, 85)        # pipeline.rs:125-133
    assert dn.derived_dims(6016, 4016) == (1280, 854, 128, 85)
    cfa = np.full((40, 60), 2048, np.uint16)
    out = c_render(refc, cfa, tw=128, th=85)
    hist = refc.histogram(refc.pack_u8(out))
    assert hist.shape == (3, 256) and np.all(hist[:, 186] == 128 * 85) and hist.sum() == 3 * 128 * 85
    assert np.array_equal(hist, dn.histogram(dn.pack_u8(out)))


# ---- twins agree bit for bit on random frames, params, maps ------------------------------------------------
@pytest.mark.parametrize("h,w", [(8, 12), (9, 13), (32, 48)])
def test_twins_agree_random(refc, rng, h, w):
    for trial in range(8):
        cfa = random_cfa(rng, h, w, 4096 if trial % 4 else 65536)
        p = random_params(rng) if trial else None
        cm = CM_TEST if trial % 2 else CM_IDENTITY
        kw = dict(tw=w + 5, th=h + 2, zoom=float(F(rng.uniform(0.6, 3))),
                  pan=(float(F(rng.uniform(-.2, .2))), float(F(rng.uniform(-.2, .2))))) if trial >= 5 else {}
        bl = 64 if trial == 3 else 0
        a = c_render(refc, cfa, p, WB_DAYLIGHT, cm, black_level=bl, **kw)
        b = np_render(cfa, p, WB_DAYLIGHT, cm, black_level=bl, **kw)
        assert ulp_diff(a, b) == 0
        assert np.array_equal(refc.pack_u8(a), dn.pack_u8(b))
        assert np.array_equal(refc.pack_f16(a).view(np.uint16), dn.pack_f16(b).view(np.uint16))


def test_pack_f16_all_cases(refc):
    x = np.concatenate([np.linspace(0, 1, 4097, dtype=F), F(2.0) ** np.arange(-30, 17, dtype=F),
                        np.array([0, 6.1e-5, 5.96e-8, 2.98e-8, 2.99e-8, 65504, 65519.9, 65520, 1e9], F),
                        np.random.default_rng(1).random(20000, dtype=F) * F(1e-4)])
    x = np.concatenate([x, -x])
    assert np.array_equal(refc.pack_f16(x).view(np.uint16), x.astype(np.float16).view(np.uint16))


# ---- the pinned pow pair: within a few ulp of libm where it matters, and <= 1 LSB at 8 bits -----------------
def test_pinned_pow_close_to_libm(refc):
    x = np.linspace(1 / 16, 1, 20001, dtype=F)
    pinned = dn.pow_pinned(x, dn.INV_GAMMA)
    exact = np.power(x.astype(np.float64), np.float64(dn.INV_GAMMA))
    rel = np.abs(pinned.astype(np.float64) - exact) / exact
    assert rel.max() < 2.5 * 2.0 ** -24          # ~2 ulp on [1/16, 1] (exhaustive scan: tools/pow_accuracy.c)
    rng = np.random.default_rng(7)
    cfa = random_cfa(rng, 24, 32)
    for _ in range(4):
        p = random_params(rng)
        a = c_render(refc, cfa, p, WB_DAYLIGHT, CM_TEST, pow_mode=0)
        b = c_render(refc, cfa, p, WB_DAYLIGHT, CM_TEST, pow_mode=1)
        d = np.abs(refc.pack_u8(a).astype(int) - refc.pack_u8(b).astype(int))
        assert d.max() <= 1


# ---- the contracted arithmetic (RD_MATH_CONTRACTED): twins agree bit for bit; vs strict <= 1 LSB at 8 bits ----
def test_contracted_mode_twins_and_deviation(refc, rng):
    worst_abs, worst_lsb = 0.0, 0
    for trial in range(12):
        cfa = random_cfa(rng, 24, 32, 4096 if trial % 3 else 65536)
        p = random_params(rng) if trial else {}
        cm = CM_TEST if trial % 2 else CM_IDENTITY
        uc = refc.make_uniforms(p, WB_DAYLIGHT, cm, math_mode=refc.MATH_CONTRACTED)
        a = refc.render_f32(cfa, uc)
        b = dn.render_f32(cfa, dn.Uniforms(**p, wb=WB_DAYLIGHT, cm=tuple(cm), math_mode="contracted"))
        assert ulp_diff(a, b) == 0
        s = refc.render_f32(cfa, refc.make_uniforms(p, WB_DAYLIGHT, cm))
        worst_abs = max(worst_abs, float(np.abs(a - s).max()))
        worst_lsb = max(worst_lsb, int(np.abs(refc.pack_u8(a).astype(int) - refc.pack_u8(s).astype(int)).max()))
    assert worst_lsb <= 1            # what the reference stores (Rgba8Unorm) moves by at most one code
    assert worst_abs < 5e-4          # f32 surface: cancellation (contrast, levels) amplifies the last-bit differences


def test_contracted_default_stack_is_still_the_identity_chain(refc):
    """With default sliders every contracted step is exact too: K1 holds in both modes."""
    cfa = np.full((4, 4), 2048, np.uint16)
    a = refc.render_f32(cfa, refc.make_uniforms(None, math_mode=refc.MATH_CONTRACTED))
    assert np.all(refc.pack_u8(a)[..., :3] == 186)


# ---- size-independent properties of the path (used again at 24 MP on the GPU) -------------------------------
def test_power_of_two_scaling_properties(refc, rng):
    """Everything before the tone steps is linear and a factor 2 is exact in binary floating point, so
    doubling the white balance, or doubling every CFA sample, is bit-identical to one more stop of exposure
    (shaders.rs:195-218).  Holds for any slider stack, matrix and size."""
    cfa = random_cfa(rng, 12, 16, 2048)                  # < 2048 so 2*cfa stays in u16 and below 4096
    for _ in range(4):
        p = random_params(rng)
        p["exposure"] = float(F(rng.uniform(-4, 3)))
        base = dict(p, exposure=float(F(p["exposure"] + 1.0)))
        a = c_render(refc, cfa, base, WB_DAYLIGHT, CM_TEST)
        b = c_render(refc, cfa, p, tuple(2 * x for x in WB_DAYLIGHT), CM_TEST)
        c = c_render(refc, (cfa * 2).astype(np.uint16), p, WB_DAYLIGHT, CM_TEST)
        assert ulp_diff(a, b) == 0 and ulp_diff(a, c) == 0


def test_native_build_of_the_oracle_is_bit_identical(refc, rng):
    """bench.py's cpu_baseline times the oracle compiled as SURVEY.md section 8d states (-O3 -march=native, contraction
    off, no fast-math; oracle/Makefile `native`, built on the machine that runs it).  Same source, other optimiser
    settings and vector ISA: every output bit must be the same as the portable -O2 checker's -- if -march=native changed
    one, that would be a finding about the restatement (an operation whose order the text does not pin)."""
    import ctypes as C
    native = refc.lib_native()
    plain = refc.lib()
    assert native is not plain
    for trial, (h, w) in enumerate([(37, 64), (64, 130), (5, 7)]):
        cfa = random_cfa(rng, h, w, hi=65536 if trial == 2 else 4096)
        for math in (refc.MATH_STRICT, refc.MATH_CONTRACTED):
            u = refc.make_uniforms(random_params(rng), WB_DAYLIGHT, CM_TEST, math_mode=math,
                                   zoom=1.0 if trial != 1 else 1.7, pan_x=0.0 if trial != 1 else 0.1)
            outs = []
            for L in (plain, native):
                out = np.empty((h, w, 4), np.float32)
                L.ref_render_f32(cfa.ctypes.data_as(C.POINTER(C.c_uint16)), w, h, C.byref(u), w, h, refc.POW_PINNED,
                                 out.ctypes.data_as(C.POINTER(C.c_float)))
                outs.append(out)
            assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32)), (trial, math)
    # and the golden vectors' expected bytes come out of the native build too
    from tests.golden_util import load_golden
    for case in load_golden():
        cfa = np.ascontiguousarray(case["cfa"])
        h, w = cfa.shape
        u = refc.make_uniforms(case["params"], case["wb"], case["cm"], case["zoom"], *case["pan"], case["black_level"])
        tw, th = case["tw"] or w, case["th"] or h
        out = np.empty((th, tw, 4), np.float32)
        native.ref_render_f32(cfa.ctypes.data_as(C.POINTER(C.c_uint16)), w, h, C.byref(u), tw, th, refc.POW_PINNED,
                              out.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(out.view(np.uint32), case["f32"].view(np.uint32)), case["name"]
