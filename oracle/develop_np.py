"""oracle/develop_np.py -- numpy float32 twin of the CPU oracle (TEST INFRASTRUCTURE ONLY).

A second, independently written restatement of the reference's develop path, used by tests/
to guard against a shared misreading in oracle/develop_ref.c.  It is vectorised
over whole images (parity masks + clipped gathers) where the C oracle is scalar per pixel, so
the two share no code structure -- only the specification:

    /root/reference/src/gpu/shaders.rs:23-60    output pixel -> CFA pixel map
    /root/reference/src/gpu/shaders.rs:104-169  nearest-neighbour demosaic
    /root/reference/src/gpu/shaders.rs:171-267  colour stack, gamma, clamp
    /root/reference/src/gpu/pipeline.rs:125-133 derived target sizes
    /root/reference/src/gpu/pipeline.rs:720-736 histogram

Never imported by the product package.  Parity status: see oracle/develop_ref.h.
"""
from __future__ import annotations

import dataclasses

import numpy as np

F = np.float32
_FLT_MIN = F(1.17549435e-38)

LOG2_C = [F(float.fromhex(h)) for h in (
    "0x1.715472p+0", "-0x1.7155bap-1", "0x1.ec7b64p-2", "-0x1.70bab4p-2",
    "0x1.2596a4p-2", "-0x1.001218p-2", "0x1.e526cap-3", "-0x1.2a7c18p-3")]
EXP2_Q = [F(float.fromhex(h)) for h in (
    "0x1p+0", "0x1.62e43p-1", "0x1.ebfbep-3", "0x1.c6aec2p-5",
    "0x1.3b2a72p-7", "0x1.5f4e2ep-10", "0x1.43e9d6p-13")]
REC709 = (F(0.2126), F(0.7152), F(0.0722))
INV_GAMMA = F(1.0 / 2.2)


@dataclasses.dataclass
class Uniforms:
    """gpu/pipeline.rs:17-46 without padding; params order = state/edit.rs:15-77."""
    exposure: float = 0.0
    contrast: float = 0.0
    highlights: float = 0.0
    shadows: float = 0.0
    whites: float = 1.0
    blacks: float = 0.0
    vibrance: float = 0.0
    saturation: float = 0.0
    temperature: float = 0.0
    tint: float = 0.0
    wb: tuple = (1.0, 1.0, 1.0, 1.0)
    cm: tuple = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)
    zoom: float = 1.0
    pan_x: float = 0.0
    pan_y: float = 0.0
    black_level: int = 0
    math_mode: str = "strict"      # "strict" (literal WGSL order) or "contracted" (fma + reciprocal multiply)


def fma32(a, b, c):
    """Exact float32 fused multiply-add: one rounding of a*b+c (round-to-odd in float64, then RNE)."""
    a = np.asarray(a, F).astype(np.float64)
    b = np.asarray(b, F).astype(np.float64)
    c = np.asarray(c, F).astype(np.float64)
    with np.errstate(invalid="ignore", over="ignore"):
        p = a * b                      # exact: 24x24 bits
        s = p + c                      # one float64 rounding
        bb = s - p
        err = (p - (s - bb)) + (c - bb)  # TwoSum: s + err == p + c exactly
        bits = s.view(np.int64).copy()
        fix = np.isfinite(s) & (err != 0) & ((bits & 1) == 0)
        step = np.where((err > 0) == (s > 0), 1, -1).astype(np.int64)
        bits = np.where(fix, bits + step, bits)
        return bits.view(np.float64).astype(F)


def log2_pinned(x):
    """x >= FLT_MIN (finite or +inf).  Same reduction and Horner order as DESIGN.md section 3."""
    x = np.asarray(x, F)
    ix = x.view(np.uint32) - np.uint32(0x3F3504F3)
    e = ix.view(np.int32) >> 23
    m = ((ix & np.uint32(0x007FFFFF)) + np.uint32(0x3F3504F3)).view(F)
    t = m - F(1.0)
    p = np.full_like(t, LOG2_C[7])
    for k in (6, 5, 4, 3, 2, 1, 0):
        p = fma32(p, t, LOG2_C[k])
    return fma32(t, p, e.astype(F))


def exp2_pinned(z):
    z = np.asarray(z, F)
    with np.errstate(invalid="ignore", over="ignore"):
        zc = np.where(np.isnan(z) | (z >= F(128.0)) | (z < F(-126.0)), F(0.0), z)
        n = np.rint(zc).astype(F)
        f = zc - n
        p = np.full_like(f, EXP2_Q[6])
        for k in (5, 4, 3, 2, 1, 0):
            p = fma32(p, f, EXP2_Q[k])
        r = (p.view(np.uint32) + (n.astype(np.int32).view(np.uint32) << np.uint32(23))).view(F)
        r = np.where(z >= F(128.0), F(np.inf), r)
        r = np.where(z < F(-126.0), F(0.0), r)
        r = np.where(np.isnan(z), F(np.nan), r)
    return r.astype(F)


def pow_pinned(x, y):
    """pow(x,y) = exp2(y*log2(x)) for y > 0: x<0 or NaN -> NaN, 0 <= x < FLT_MIN -> 0."""
    x = np.asarray(x, F)
    y = F(y)
    with np.errstate(invalid="ignore", over="ignore"):
        ok = x >= _FLT_MIN
        xs = np.where(ok, x, F(1.0))
        r = exp2_pinned(y * log2_pinned(xs))
        r = np.where(ok, r, F(0.0))
        r = np.where((x < F(0.0)) | np.isnan(x), F(np.nan), r)
    return r.astype(F)


def pow_libm(x, y):
    x = np.asarray(x, F)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        r = np.power(x.astype(np.float64), np.float64(F(y))).astype(F)
        r = np.where((x >= 0) & (x < _FLT_MIN), F(0.0), r)
        r = np.where((x < F(0.0)) | np.isnan(x), F(np.nan), r)
    return r.astype(F)


def derived_dims(w: int, h: int):
    """pipeline.rs:125-133."""
    aspect = F(w) / F(h)
    pw = min(w, 1280)
    ph = int(F(pw) / aspect)
    hh = int(F(128) / aspect)
    return pw, ph, 128, hh


def pixel_map(w, h, tw, th, zoom, pan_x, pan_y):
    """shaders.rs:23-60 + :174-187 -> (px, py, inside) int32/bool arrays of shape (th, tw)."""
    i = np.arange(tw, dtype=np.int64).astype(F)
    j = np.arange(th, dtype=np.int64).astype(F)
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        sx = (i + F(0.5)) / F(tw)
        sy = (j + F(0.5)) / F(th)
        tx = ((sx - F(0.5)) / F(zoom) - F(pan_x)) + F(0.5)
        ty = ((sy - F(0.5)) / F(zoom) - F(pan_y)) + F(0.5)
        inx = ~((tx < 0) | (tx > 1))           # as the shader writes the test (:174-175): a NaN coordinate passes it ...
        iny = ~((ty < 0) | (ty > 1))
        px = np.where(inx & ~np.isnan(tx), tx * F(w), F(0)).astype(np.int32)      # ... and converts to 0 (i32 of a NaN)
        py = np.where(iny & ~np.isnan(ty), ty * F(h), F(0)).astype(np.int32)
    # tx == 1.0 exactly -> px == w (one past the texture): kept as it is, the parity of shaders.rs:115-118 is taken on it;
    # every load clamps (demosaic)
    PX, PY = np.meshgrid(px, py)
    inside = np.logical_and.outer(iny, inx)
    return PX, PY, inside


def demosaic(cfa, PX, PY, black_level=0):
    """shaders.rs:104-169 as parity masks + clipped gathers."""
    h, w = cfa.shape
    raw = cfa.astype(np.int64)
    raw = np.maximum(raw - int(black_level), 0)
    v = raw.astype(F) * F(1.0 / 4096.0)

    def tap(dx, dy):
        return v[np.clip(PY + dy, 0, h - 1), np.clip(PX + dx, 0, w - 1)]

    row_even = ((PY + 1) % 2) == 0       # parity is taken on py+1 (shaders.rs:115)
    col_even = (PX % 2) == 0
    c = tap(0, 0)                        # (the centre load of :106 sees px == w when tex_coords.x == 1.0: lowered as a clamp)
    # four cases of shaders.rs:127-155
    r = np.where(row_even, np.where(col_even, tap(0, 1), tap(-1, 1)),
                 np.where(col_even, c, tap(-1, 0)))
    g = np.where(row_even, np.where(col_even, c, tap(-1, 0)),
                 np.where(col_even, tap(1, 0), c))
    b = np.where(row_even, np.where(col_even, tap(1, 0), c),
                 tap(0, -1))
    return r.astype(F), g.astype(F), b.astype(F)


def _dot709(r, g, b):
    return ((r * REC709[0]) + (g * REC709[1])) + (b * REC709[2])


def colour_stack(r, g, b, u: Uniforms, pow_mode="pinned"):
    """shaders.rs:192-266, literal operation order, float32 throughout."""
    powf = pow_pinned if pow_mode == "pinned" else pow_libm
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        r = r * F(u.wb[0]); g = g * F(u.wb[1]); b = b * F(u.wb[2])
        r = r * (F(1.0) + F(u.temperature) * F(0.3))
        b = b * (F(1.0) - F(u.temperature) * F(0.3))
        g = g * (F(1.0) + F(u.tint) * F(0.3))
        m = [F(x) for x in u.cm]
        x = ((m[0] * r) + (m[3] * g)) + (m[6] * b)
        y = ((m[1] * r) + (m[4] * g)) + (m[7] * b)
        z = ((m[2] * r) + (m[5] * g)) + (m[8] * b)
        r, g, b = x, y, z
        em = powf(np.array([2.0], F), F(u.exposure))[0]
        r = r * em; g = g * em; b = b * em
        L = _dot709(r, g, b)
        hl = F(1.0) + (L * F(u.highlights))
        r = r * hl; g = g * hl; b = b * hl
        sh = F(1.0) + ((F(1.0) - L) * F(u.shadows))
        r = r * sh; g = g * sh; b = b * sh
        cf = F(1.0) + (F(u.contrast) / F(100.0))
        r = (r - F(0.5)) * cf + F(0.5); g = (g - F(0.5)) * cf + F(0.5); b = (b - F(0.5)) * cf + F(0.5)
        den = (F(u.whites) - F(u.blacks)) + F(0.0001)
        r = (r - F(u.blacks)) / den; g = (g - F(u.blacks)) / den; b = (b - F(u.blacks)) / den
        Y = _dot709(r, g, b)
        s = F(1.0) + (F(u.saturation) / F(100.0))
        ys = Y * (F(1.0) - s)
        r = ys + r * s; g = ys + g * s; b = ys + b * s
        sat = np.fmax(r, np.fmax(g, b)) - np.fmin(r, np.fmin(g, b))
        va = F(u.vibrance) * (F(1.0) - sat)
        Y2 = _dot709(r, g, b)
        a2 = F(1.0) + va
        yv = Y2 * (F(1.0) - a2)
        r = yv + r * a2; g = yv + g * a2; b = yv + b * a2
        r = powf(r, INV_GAMMA); g = powf(g, INV_GAMMA); b = powf(b, INV_GAMMA)
        out = [np.fmin(np.fmax(c, F(0.0)), F(1.0)) for c in (r, g, b)]   # NaN -> 0
    return [c.astype(F) for c in out]


def _dot709_fma(r, g, b):
    return fma32(b, REC709[2], fma32(g, REC709[1], r * REC709[0]))


def colour_stack_contracted(r, g, b, u: Uniforms, pow_mode="pinned"):
    """Same shader, every mul+add contracted to one fma, levels division as x*RN(1/d)."""
    powf = pow_pinned if pow_mode == "pinned" else pow_libm
    one, half = F(1.0), F(0.5)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        r = r * F(u.wb[0]); g = g * F(u.wb[1]); b = b * F(u.wb[2])
        r = r * fma32(F(u.temperature), F(0.3), one)
        b = b * fma32(-F(u.temperature), F(0.3), one)
        g = g * fma32(F(u.tint), F(0.3), one)
        m = [F(x) for x in u.cm]
        x = fma32(m[6], b, fma32(m[3], g, m[0] * r))
        y = fma32(m[7], b, fma32(m[4], g, m[1] * r))
        z = fma32(m[8], b, fma32(m[5], g, m[2] * r))
        em = powf(np.array([2.0], F), F(u.exposure))[0]
        r = x * em; g = y * em; b = z * em
        L = _dot709_fma(r, g, b)
        hl = fma32(L, F(u.highlights), one)
        r = r * hl; g = g * hl; b = b * hl
        sh = fma32(one - L, F(u.shadows), one)
        r = r * sh; g = g * sh; b = b * sh
        cf = one + (F(u.contrast) / F(100.0))
        r = fma32(r - half, cf, half); g = fma32(g - half, cf, half); b = fma32(b - half, cf, half)
        den = (F(u.whites) - F(u.blacks)) + F(0.0001)
        rden = one / den
        r = (r - F(u.blacks)) * rden; g = (g - F(u.blacks)) * rden; b = (b - F(u.blacks)) * rden
        Y = _dot709_fma(r, g, b)
        s = one + (F(u.saturation) / F(100.0))
        ys = Y * (one - s)
        r = fma32(r, s, ys); g = fma32(g, s, ys); b = fma32(b, s, ys)
        sat = np.fmax(r, np.fmax(g, b)) - np.fmin(r, np.fmin(g, b))
        a2 = fma32(F(u.vibrance), one - sat, one)
        Y2 = _dot709_fma(r, g, b)
        yv = Y2 * (one - a2)
        r = fma32(r, a2, yv); g = fma32(g, a2, yv); b = fma32(b, a2, yv)
        r = powf(r, INV_GAMMA); g = powf(g, INV_GAMMA); b = powf(b, INV_GAMMA)
        out = [np.fmin(np.fmax(c, F(0.0)), F(1.0)) for c in (r, g, b)]
    return [c.astype(F) for c in out]


def render_f32(cfa, u: Uniforms, tw=None, th=None, pow_mode="pinned"):
    """(h,w) uint16 -> (th,tw,4) float32, alpha = 1."""
    cfa = np.asarray(cfa, np.uint16)
    h, w = cfa.shape
    tw = w if tw is None else tw
    th = h if th is None else th
    PX, PY, inside = pixel_map(w, h, tw, th, u.zoom, u.pan_x, u.pan_y)
    r, g, b = demosaic(cfa, PX, PY, u.black_level)
    stack = colour_stack_contracted if u.math_mode == "contracted" else colour_stack
    r, g, b = stack(r, g, b, u, pow_mode)
    out = np.empty((th, tw, 4), F)
    out[..., 0] = np.where(inside, r, F(0))
    out[..., 1] = np.where(inside, g, F(0))
    out[..., 2] = np.where(inside, b, F(0))
    out[..., 3] = F(1.0)
    return out


def pack_u8(rgba):
    """Rgba8Unorm store (pipeline.rs:322), pinned as trunc(x*255 + 0.5)."""
    return (np.asarray(rgba, F) * F(255.0) + F(0.5)).astype(np.uint8)


def pack_f16(rgba):
    return np.asarray(rgba, F).astype(np.float16)


def histogram(rgba8):
    """pipeline.rs:720-736 -> (3,256) uint32."""
    px = np.asarray(rgba8, np.uint8).reshape(-1, 4)
    return np.stack([np.bincount(px[:, c], minlength=256) for c in range(3)]).astype(np.uint32)
