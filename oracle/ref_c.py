"""oracle/ref_c.py -- ctypes binding of the C oracle (oracle/libdevelop_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py -- never by the product package.  Build with `make -C oracle`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdevelop_ref.so")


class RefEditParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "exposure", "contrast", "highlights", "shadows", "whites", "blacks",
        "vibrance", "saturation", "temperature", "tint")]


class RefUniforms(C.Structure):
    _fields_ = [("p", RefEditParams), ("wb", C.c_float * 4), ("cm", C.c_float * 9),
                ("zoom", C.c_float), ("pan_x", C.c_float), ("pan_y", C.c_float),
                ("black_level", C.c_uint32), ("math_mode", C.c_uint32)]


PARAM_NAMES = [n for n, _ in RefEditParams._fields_]
POW_PINNED, POW_LIBM = 0, 1
MATH_STRICT, MATH_CONTRACTED = 0, 1


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("develop_ref.c", "develop_ref.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True)
    return _SO


_SO_NATIVE = os.path.join(_HERE, "libdevelop_ref_native.so")
_NATIVE_TAG = _SO_NATIVE + ".host"


def _host_tag() -> str:
    """Which CPU a -march=native build belongs to: model name + the flag set of the first core."""
    model, flags = "", ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and not model:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("flags") and not flags:
                    flags = line.split(":", 1)[1].strip()
                if model and flags:
                    break
    except OSError:
        pass
    import hashlib
    return f"{model}|{hashlib.sha256(flags.encode()).hexdigest()[:16]}"


def build_native() -> str:
    """The -O3 -march=native flavour (bench.py's cpu_baseline), compiled on THIS machine: rebuilt whenever the
    sources changed or the file was built for another CPU."""
    src = [os.path.join(_HERE, f) for f in ("develop_ref.c", "develop_ref.h", "Makefile")]
    tag = _host_tag()
    have = ""
    try:
        have = open(_NATIVE_TAG).read()
    except OSError:
        pass
    stale = (not os.path.exists(_SO_NATIVE)) or have != tag or any(os.path.getmtime(s) > os.path.getmtime(_SO_NATIVE) for s in src)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "-s", "native"], check=True)
        with open(_NATIVE_TAG, "w") as f:
            f.write(tag)
    return _SO_NATIVE


_lib = None
_lib_native = None


def _bind(path):
        L = C.CDLL(path)
        u16p, f32p, u8p, u32p = (C.POINTER(t) for t in (C.c_uint16, C.c_float, C.c_uint8, C.c_uint32))
        UP = C.POINTER(RefUniforms)
        L.ref_log2f.restype = C.c_float; L.ref_log2f.argtypes = [C.c_float]
        L.ref_exp2f.restype = C.c_float; L.ref_exp2f.argtypes = [C.c_float]
        L.ref_powf.restype = C.c_float; L.ref_powf.argtypes = [C.c_float, C.c_float, C.c_int]
        L.ref_default_params.argtypes = [C.POINTER(RefEditParams)]
        L.ref_derived_dims.argtypes = [C.c_uint32, C.c_uint32] + [u32p] * 4
        L.ref_render_f32.argtypes = [u16p, C.c_uint32, C.c_uint32, UP, C.c_uint32, C.c_uint32, C.c_int, f32p]
        L.ref_render_f32_mt.argtypes = [u16p, C.c_uint32, C.c_uint32, UP, C.c_uint32, C.c_uint32, C.c_int, f32p, C.c_int]
        L.ref_render_f32_band.argtypes = [u16p, C.c_uint32, C.c_uint32, UP, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_int, f32p]
        L.ref_render_f32_band.restype = None
        L.ref_bench_mt.argtypes = [u16p, C.c_uint32, C.c_uint32, UP, C.c_int, C.c_double, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.ref_bench_mt.restype = None
        L.ref_pack_u8.argtypes = [f32p, C.c_size_t, u8p]
        L.ref_pack_f16.argtypes = [f32p, C.c_size_t, u16p]
        L.ref_histogram.argtypes = [u8p, C.c_size_t, u32p]
        for f in (L.ref_default_params, L.ref_derived_dims, L.ref_render_f32, L.ref_render_f32_mt,
                  L.ref_pack_u8, L.ref_pack_f16, L.ref_histogram):
            f.restype = None
        return L


def lib():
    global _lib
    if _lib is None:
        _lib = _bind(build())
    return _lib


def lib_native():
    """The same oracle compiled -O3 -march=native on this machine (cpu_baseline only; the checker is lib())."""
    global _lib_native
    if _lib_native is None:
        _lib_native = _bind(build_native())
    return _lib_native


def make_uniforms(params=None, wb=(1, 1, 1, 1), cm=(1, 0, 0, 0, 1, 0, 0, 0, 1),
                  zoom=1.0, pan_x=0.0, pan_y=0.0, black_level=0, math_mode=MATH_STRICT) -> RefUniforms:
    """params: dict of slider name -> value (missing = EditParams::default(), state/edit.rs:81-95)."""
    u = RefUniforms()
    lib().ref_default_params(C.byref(u.p))
    for k, v in (params or {}).items():
        if k not in PARAM_NAMES:
            raise KeyError(k)
        setattr(u.p, k, float(v))
    u.wb[:] = [float(x) for x in wb]
    u.cm[:] = [float(x) for x in cm]
    u.zoom, u.pan_x, u.pan_y, u.black_level = float(zoom), float(pan_x), float(pan_y), int(black_level)
    u.math_mode = int(math_mode)
    return u


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def render_f32(cfa: np.ndarray, u: RefUniforms, tw=None, th=None, pow_mode=POW_PINNED, nthreads=1) -> np.ndarray:
    cfa = np.ascontiguousarray(cfa, np.uint16)
    h, w = cfa.shape
    tw = w if tw is None else int(tw)
    th = h if th is None else int(th)
    out = np.empty((th, tw, 4), np.float32)
    if tw and th:
        lib().ref_render_f32_mt(_ptr(cfa, C.c_uint16), w, h, C.byref(u), tw, th, pow_mode,
                                _ptr(out, C.c_float), int(nthreads))
    return out


def render_band(cfa: np.ndarray, u: RefUniforms, row0: int, row1: int, tw=None, th=None,
                pow_mode=POW_PINNED) -> np.ndarray:
    """Rows [row0,row1) of the target surface only (full-size frames: sample a few bands)."""
    cfa = np.ascontiguousarray(cfa, np.uint16)
    h, w = cfa.shape
    tw = w if tw is None else int(tw)
    th = h if th is None else int(th)
    out = np.empty((row1 - row0, tw, 4), np.float32)
    lib().ref_render_f32_band(_ptr(cfa, C.c_uint16), w, h, C.byref(u), tw, th, row0, row1, pow_mode,
                              _ptr(out, C.c_float))
    return out


def pack_u8(rgba: np.ndarray) -> np.ndarray:
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.empty(rgba.shape, np.uint8)
    lib().ref_pack_u8(_ptr(rgba, C.c_float), rgba.size, _ptr(out, C.c_uint8))
    return out


def pack_f16(rgba: np.ndarray) -> np.ndarray:
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.empty(rgba.shape, np.uint16)
    lib().ref_pack_f16(_ptr(rgba, C.c_float), rgba.size, _ptr(out, C.c_uint16))
    return out.view(np.float16)


def histogram(rgba8: np.ndarray) -> np.ndarray:
    rgba8 = np.ascontiguousarray(rgba8, np.uint8)
    out = np.zeros(768, np.uint32)
    lib().ref_histogram(_ptr(rgba8, C.c_uint8), rgba8.size // 4, _ptr(out, C.c_uint32))
    return out.reshape(3, 256)


def derived_dims(w: int, h: int):
    v = [C.c_uint32() for _ in range(4)]
    lib().ref_derived_dims(w, h, *[C.byref(x) for x in v])
    return tuple(x.value for x in v)
