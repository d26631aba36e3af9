"""RenderPipeline -- host mirror of `gpu::RenderPipeline` (reference src/gpu/pipeline.rs:81-100,
:112-737) over the librawdev C ABI.  Same method names, argument meaning and error behaviour:
`new` raises RawdevError where the reference returns Err(String); renders return tightly packed
RGBA8 bytes (1-D uint8 arrays, the analogue of Vec<u8>).  All pixel work happens in HIP kernels
on the MI355X; nothing here computes pixels on the CPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import FMT_RGBA_F16, FMT_RGBA_F32, FMT_RGBA_U8, FMT_RGB_U8, BYTES_PER_PIXEL, RawdevError, check
from .edit import EditParams

IDENTITY_MATRIX = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)


def calculate_cam_to_srgb_matrix(xyz_to_cam: Sequence[float]) -> Tuple[float, ...]:
    """color::calculate_cam_to_srgb_matrix (reference src/color.rs:35-47): the reference returns the
    identity for ANY input (the real maths is commented out), and so does this mirror."""
    if len(xyz_to_cam) != 9:
        raise ValueError("xyz_to_cam must have 9 elements")
    return IDENTITY_MATRIX


def is_identity_matrix(matrix: Sequence[float]) -> bool:
    """color::is_identity_matrix (reference src/color.rs:172-178): every element within 0.001 of the identity."""
    if len(matrix) != 9:
        raise ValueError("matrix must have 9 elements")
    return all(abs(float(np.float32(m)) - float(np.float32(i))) < 0.001 for m, i in zip(matrix, IDENTITY_MATRIX))


def derived_dims(width: int, height: int) -> Tuple[int, int, int, int]:
    """(preview_w, preview_h, hist_w, hist_h) with pipeline.rs:125-133's truncating f32 arithmetic."""
    v = [C.c_uint32() for _ in range(4)]
    check(_lib.lib().rd_derived_dims(width, height, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


ELIDED_STEP_NAMES = {1: "temperature/tint", 2: "matrix", 4: "exposure", 8: "highlights", 16: "shadows", 32: "saturation",
                     64: "vibrance", 128: "divide fix-up", 256: "blacks"}


def elided_steps(params: EditParams, wb_multipliers: Sequence[float], color_matrix: Sequence[float],
                 math_mode: int = 0) -> int:
    """Bit mask of the colour-stack steps the export kernel skips for these uniforms (exact identities; diagnostic,
    needs no device).  See ELIDED_STEP_NAMES and DESIGN.md section 4."""
    wb = (C.c_float * 4)(*[float(x) for x in wb_multipliers])
    cm = (C.c_float * 9)(*[float(x) for x in color_matrix])
    cp = params.to_c()
    return int(_lib.lib().rd_elided_steps(C.byref(cp), wb, cm, int(math_mode)))


class PinnedBytes:
    """Page-locked host memory (rd_host_alloc): a render destination the DMA engines write directly -- what a host
    hands to render_full_res_to_bytes(out=...) to skip the staging copy.  `.array` is a uint8 view; free() (or the
    destructor) returns the memory, after which the view must not be used."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = int(device), int(nbytes)
        p = C.c_void_p()
        check(_lib.lib().rd_host_alloc(self.device, self.nbytes, C.byref(p)))
        self.ptr = p.value
        self.array = np.frombuffer((C.c_uint8 * self.nbytes).from_address(self.ptr), dtype=np.uint8)

    def free(self) -> None:
        if getattr(self, "ptr", 0):
            self.array = None
            check(_lib.lib().rd_host_free(self.device, C.c_void_p(self.ptr)))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class BorrowedSurface:
    """A surface lent by rd_render_full_res_borrow; give it back with release() (or use it as a context manager)."""

    def __init__(self, pipe, ptr: int, nbytes: int):
        self._pipe, self.ptr, self.nbytes = pipe, ptr, nbytes
        self.array = np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=np.uint8)

    def release(self) -> None:
        if self.ptr:
            self.array = None
            check(_lib.lib().rd_surface_release(self._pipe._h, C.c_void_p(self.ptr)))
            self.ptr = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()
        return False


def measure_hbm(device: int = 0, nbytes: int = 1 << 30, reps: int = 5):
    """(copy, fill, read, hipMemsetAsync) GB/s of this device right now (rd_measure_hbm: librawdev's own streaming kernels)."""
    v = [C.c_double() for _ in range(4)]
    check(_lib.lib().rd_measure_hbm(int(device), int(nbytes), int(reps), *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def measure_valu(device: int = 0) -> float:
    """Nanoseconds one full-rate VALU wave-instruction costs a SIMD of this device right now (rd_measure_valu)."""
    v = C.c_double()
    check(_lib.lib().rd_measure_valu(int(device), C.byref(v)))
    return v.value


class RenderPipeline:
    """Owns one CFA plane in HBM plus the current uniforms (EditParams, wb, matrix, zoom/pan)."""

    def __init__(self):
        raise TypeError("use RenderPipeline.new(...)")

    @classmethod
    def new(cls, image_id: int, raw_data, width: int, height: int, params: EditParams,
            wb_multipliers: Sequence[float], color_matrix: Sequence[float], device: int = 0) -> "RenderPipeline":
        """pipeline.rs:114-363.  raw_data: width*height u16, row-major, one sample per photosite."""
        raw = np.ascontiguousarray(raw_data, dtype=np.uint16).reshape(-1)
        if raw.size != int(width) * int(height):
            raise RawdevError(-1, f"raw_data has {raw.size} samples, expected {width}x{height}")
        if len(wb_multipliers) != 4 or len(color_matrix) != 9:
            raise RawdevError(-1, "wb_multipliers must have 4 and color_matrix 9 elements")
        self = object.__new__(cls)
        self._h = C.c_void_p()
        self._device = device
        wb = (C.c_float * 4)(*[float(x) for x in wb_multipliers])
        cm = (C.c_float * 9)(*[float(x) for x in color_matrix])
        cp = params.to_c()
        check(_lib.lib().rd_pipeline_create(device, int(image_id), raw.ctypes.data_as(C.c_void_p), int(width),
                                            int(height), C.byref(cp), wb, cm, C.byref(self._h)))
        return self._finish()

    @classmethod
    def from_device(cls, image_id: int, cfa_dev: int, width: int, height: int, params: EditParams,
                    wb_multipliers: Sequence[float], color_matrix: Sequence[float], device: int = 0) -> "RenderPipeline":
        """The same pipeline over a CFA plane that already lives in `device`'s HBM (rd_pipeline_create_from_device:
        the plane is copied device-to-device; `cfa_dev` is a device pointer as int)."""
        if len(wb_multipliers) != 4 or len(color_matrix) != 9:
            raise RawdevError(-1, "wb_multipliers must have 4 and color_matrix 9 elements")
        self = object.__new__(cls)
        self._h = C.c_void_p()
        self._device = device
        wb = (C.c_float * 4)(*[float(x) for x in wb_multipliers])
        cm = (C.c_float * 9)(*[float(x) for x in color_matrix])
        cp = params.to_c()
        check(_lib.lib().rd_pipeline_create_from_device(device, int(image_id), C.c_void_p(int(cfa_dev)), int(width),
                                                        int(height), C.byref(cp), wb, cm, C.byref(self._h)))
        return self._finish()

    def _finish(self) -> "RenderPipeline":
        info = _lib.RdInfo()
        check(_lib.lib().rd_pipeline_info(self._h, C.byref(info)))
        # the reference's pub fields (pipeline.rs:89-96)
        self.width, self.height = info.width, info.height
        self.preview_width, self.preview_height = info.preview_width, info.preview_height
        self.histogram_width, self.histogram_height = info.histogram_width, info.histogram_height
        self.image_id = info.image_id
        return self

    # -- uniforms ---------------------------------------------------------------------------------
    def update_uniforms(self, params: EditParams) -> None:                       # pipeline.rs:367-369
        cp = params.to_c()
        check(_lib.lib().rd_update_uniforms(self._h, C.byref(cp)))

    def update_uniforms_with_zoom(self, params: EditParams, zoom: float, pan_x: float, pan_y: float) -> None:
        cp = params.to_c()                                                        # pipeline.rs:373-398
        check(_lib.lib().rd_update_uniforms_with_zoom(self._h, C.byref(cp), float(zoom), float(pan_x), float(pan_y)))

    def set_black_level(self, black_level: int) -> None:
        check(_lib.lib().rd_pipeline_set_black_level(self._h, int(black_level)))

    def set_matrix_layout(self, layout: int) -> None:
        """MATRIX_REFERENCE (default: rows consumed as columns, shaders.rs:209-214) or MATRIX_ROW_MAJOR (out = M c)."""
        check(_lib.lib().rd_pipeline_set_matrix_layout(self._h, int(layout)))

    def set_math_mode(self, math_mode: int) -> None:
        """MATH_STRICT (default, literal WGSL order) or MATH_CONTRACTED (fma + reciprocal multiply)."""
        check(_lib.lib().rd_pipeline_set_math_mode(self._h, int(math_mode)))

    # -- renders ----------------------------------------------------------------------------------
    def _bytes(self, fn, w: int, h: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        if out is None:
            out = np.empty(w * h * 4, np.uint8)           # the analogue of the reference's fresh Vec<u8>
        elif out.dtype != np.uint8 or not out.flags.c_contiguous or out.size != w * h * 4:
            raise RawdevError(-1, f"out must be a contiguous uint8 array of {w * h * 4} bytes")
        check(fn(self._h, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def render_to_bytes(self) -> np.ndarray:                                      # pipeline.rs:442-522
        return self._bytes(_lib.lib().rd_render_to_bytes, self.preview_width, self.preview_height)

    def render_full_res_to_bytes(self, out: Optional[np.ndarray] = None) -> np.ndarray:   # pipeline.rs:526-606
        """`out` (optional): the destination to fill instead of a fresh array -- a PinnedBytes(...).array takes the
        direct-DMA path, any other array the staged one (rd_render_full_res_to_bytes in include/rawdev.h)."""
        return self._bytes(_lib.lib().rd_render_full_res_to_bytes, self.width, self.height, out)

    def render_full_res_borrowed(self):
        """render_full_res_to_bytes into page-locked memory the pipeline owns: returns a BorrowedSurface whose `.array` is the
        uint8 view (valid until `.release()` / the end of the `with` block).  No allocation, no page faults, no host copy."""
        data, n = C.c_void_p(), C.c_size_t()
        check(_lib.lib().rd_render_full_res_borrow(self._h, C.byref(data), C.byref(n)))
        return BorrowedSurface(self, data.value, n.value)

    def render_to_histogram_bytes(self) -> np.ndarray:                            # pipeline.rs:615-716
        return self._bytes(_lib.lib().rd_render_to_histogram_bytes, self.histogram_width, self.histogram_height)

    def calculate_histogram(self, rgba_bytes) -> np.ndarray:                      # pipeline.rs:720-736
        """[[u32;256];3] as a (3,256) uint32 array; trailing bytes that do not fill a pixel are
        ignored like chunks_exact(4)."""
        buf = np.ascontiguousarray(rgba_bytes, dtype=np.uint8).reshape(-1)
        n = buf.size - buf.size % 4
        hist = np.zeros(768, np.uint32)
        check(_lib.lib().rd_calculate_histogram(self._h, buf.ctypes.data_as(C.c_void_p), n,
                                                hist.ctypes.data_as(C.c_void_p)))
        return hist.reshape(3, 256)

    def dimensions(self) -> Tuple[int, int]:                                      # pipeline.rs:609-611
        return (self.width, self.height)

    # -- extensions beyond the reference surface -----------------------------------------------------
    def render(self, out_w: Optional[int] = None, out_h: Optional[int] = None, fmt: int = FMT_RGBA_F32,
               with_histogram: bool = False):
        """Any target size / surface format with the current uniforms.  Returns the surface as an
        (h, w, 4) array (float32 / float16 / uint8) and, if requested, the fused (3,256) histogram."""
        w = self.width if out_w is None else int(out_w)
        h = self.height if out_h is None else int(out_h)
        dt = {FMT_RGBA_F32: np.float32, FMT_RGBA_F16: np.float16, FMT_RGBA_U8: np.uint8, FMT_RGB_U8: np.uint8}.get(fmt)
        if dt is None:
            raise RawdevError(-1, f"unknown format {fmt}")
        out = np.empty((h, w, 3 if fmt == FMT_RGB_U8 else 4), dt)
        hist = np.zeros(768, np.uint32) if with_histogram else None
        check(_lib.lib().rd_render(self._h, w, h, fmt, out.ctypes.data_as(C.c_void_p), out.nbytes,
                                   hist.ctypes.data_as(C.c_void_p) if with_histogram else None))
        return (out, hist.reshape(3, 256)) if with_histogram else out

    def render_device(self, out_w: int, out_h: int, fmt: int, dst_dev: int, hist_dev: int = 0, stream: int = 0) -> None:
        """Enqueue on `stream` (a hipStream_t as int) into device memory; not synchronised."""
        check(_lib.lib().rd_render_device(self._h, int(out_w), int(out_h), fmt, C.c_void_p(dst_dev),
                                          C.c_void_p(hist_dev) if hist_dev else None,
                                          C.c_void_p(stream) if stream else None))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().rd_pipeline_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):                                                           # pipeline.rs:103-110
        return f"RenderPipeline {{ width: {self.width}, height: {self.height}, .. }}"
