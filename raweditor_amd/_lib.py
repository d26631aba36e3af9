"""ctypes binding of librawdev.so (include/rawdev.h).  Fails loudly: there is no fallback path."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

from .build import LIB_PATH

RD_OK = 0
RD_ERR_INVALID_ARG, RD_ERR_NO_DEVICE, RD_ERR_HIP, RD_ERR_OOM, RD_ERR_UNSUPPORTED, RD_ERR_INTERNAL = -1, -2, -3, -4, -5, -6   # rd_status
FAULT_BAD_ALLOC, FAULT_THREAD_START, FAULT_RUNTIME, FAULT_FOREIGN = 1, 2, 3, 4                                            # rd_fault_kind
ABI_VERSION = 5          # include/rawdev.h RD_ABI_VERSION
FMT_RGBA_F32, FMT_RGBA_F16, FMT_RGBA_U8, FMT_RGB_U8 = 0, 1, 2, 3
MATH_STRICT, MATH_CONTRACTED = 0, 1
MATRIX_REFERENCE, MATRIX_ROW_MAJOR = 0, 1
BYTES_PER_PIXEL = {FMT_RGBA_F32: 16, FMT_RGBA_F16: 8, FMT_RGBA_U8: 4, FMT_RGB_U8: 3}


class RawdevError(RuntimeError):
    """A non-zero rd_status; the reference's `Err(String)` (pipeline.rs:122)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"librawdev error {code}: {message}")
        self.code = code
        self.message = message


class RdEditParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "exposure", "contrast", "highlights", "shadows", "whites", "blacks",
        "vibrance", "saturation", "temperature", "tint")]


class RdInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32),
                ("preview_width", C.c_uint32), ("preview_height", C.c_uint32),
                ("histogram_width", C.c_uint32), ("histogram_height", C.c_uint32),
                ("image_id", C.c_int64)]


class RdFrame(C.Structure):
    _fields_ = [("cfa_dev", C.c_void_p), ("out_dev", C.c_void_p), ("params", RdEditParams),
                ("wb_multipliers", C.c_float * 4), ("color_matrix", C.c_float * 9),
                ("black_level", C.c_uint32), ("matrix_layout", C.c_uint32)]


# every symbol include/rawdev.h declares: (restype, argtypes)
_VP, _SZ, _U32, _I = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
PROTOTYPES = {
    "rd_abi_version": (_I, []),
    "rd_last_error": (C.c_char_p, []),
    "rd_device_count": (_I, [C.POINTER(_I)]),
    "rd_device_identity": (_I, [_I, C.c_char_p, _SZ, C.c_char_p, _SZ]),
    "rd_edit_params_default": (None, [C.POINTER(RdEditParams)]),
    "rd_derived_dims": (_I, [_U32, _U32] + [C.POINTER(_U32)] * 4),
    "rd_format_bytes_per_pixel": (_SZ, [_U32]),
    "rd_elided_steps": (_U32, [C.POINTER(RdEditParams), C.POINTER(C.c_float), C.POINTER(C.c_float), _U32]),
    "rd_pipeline_create": (_I, [_I, C.c_int64, _VP, _U32, _U32, C.POINTER(RdEditParams),
                                C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_VP)]),
    "rd_pipeline_create_from_device": (_I, [_I, C.c_int64, _VP, _U32, _U32, C.POINTER(RdEditParams),
                                            C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_VP)]),
    "rd_pipeline_destroy": (None, [_VP]),
    "rd_pipeline_info": (_I, [_VP, C.POINTER(RdInfo)]),
    "rd_pipeline_set_black_level": (_I, [_VP, _U32]),
    "rd_pipeline_set_math_mode": (_I, [_VP, _U32]),
    "rd_pipeline_set_matrix_layout": (_I, [_VP, _U32]),
    "rd_update_uniforms": (_I, [_VP, C.POINTER(RdEditParams)]),
    "rd_update_uniforms_with_zoom": (_I, [_VP, C.POINTER(RdEditParams), C.c_float, C.c_float, C.c_float]),
    "rd_render_to_bytes": (_I, [_VP, _VP, _SZ]),
    "rd_render_full_res_to_bytes": (_I, [_VP, _VP, _SZ]),
    "rd_render_to_histogram_bytes": (_I, [_VP, _VP, _SZ]),
    "rd_render_full_res_borrow": (_I, [_VP, C.POINTER(_VP), C.POINTER(_SZ)]),
    "rd_surface_release": (_I, [_VP, _VP]),
    "rd_calculate_histogram": (_I, [_VP, _VP, _SZ, _VP]),
    "rd_render": (_I, [_VP, _U32, _U32, _U32, _VP, _SZ, _VP]),
    "rd_render_device": (_I, [_VP, _U32, _U32, _U32, _VP, _VP, _VP]),
    "rd_batch_create": (_I, [_I, _U32, _U32, _U32, _U32, C.POINTER(_VP)]),
    "rd_batch_destroy": (None, [_VP]),
    "rd_batch_set_math_mode": (_I, [_VP, _U32]),
    "rd_batch_develop": (_I, [_VP, C.POINTER(RdFrame), _SZ, _U32, _VP]),
    "rd_batch_plan_launches": (_I, [_U32, _U32, _U32, _U32, C.POINTER(RdFrame), _SZ, _U32, C.POINTER(_U32), _SZ]),
    "rd_batch_last_launch_count": (_U32, [_VP]),
    "rd_batch_set_launch_timing": (_I, [_VP, _U32]),
    "rd_batch_launch_timeline": (_I, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(_U32), _SZ, C.POINTER(_U32)]),
    "rd_batch_probe_pattern": (_I, [_VP, C.POINTER(RdFrame), _SZ, _VP]),
    "rd_batch_measure_clock": (_I, [_VP, C.POINTER(RdFrame), _SZ, _VP] + [C.POINTER(C.c_double)] * 4),
    "rd_batch_histogram": (_I, [_VP, _VP, _VP]),
    "rd_node_batch_create": (_I, [C.POINTER(_I), _U32, _U32, _U32, _U32, _U32, C.POINTER(_VP)]),
    "rd_node_batch_destroy": (None, [_VP]),
    "rd_node_batch_set_math_mode": (_I, [_VP, _U32]),
    "rd_node_batch_device_of": (_U32, [_U32, _SZ]),
    "rd_node_batch_develop": (_I, [_VP, C.POINTER(RdFrame), _SZ, _U32]),
    "rd_node_batch_histogram": (_I, [_VP, _VP]),
    "rd_node_batch_histogram_enqueue": (_I, [_VP]),
    "rd_node_batch_histogram_fetch": (_I, [_VP, _VP]),
    "rd_node_batch_synchronize": (_I, [_VP]),
    "rd_node_batch_stream": (_VP, [_VP, _U32]),
    "rd_node_batch_last_launch_count": (_U32, [_VP, _U32]),
    "rd_node_batch_reduce_kind": (_I, [_VP]),
    "rd_exporter_create": (_I, [_I, _U32, _U32, _U32, _U32, _U32, C.POINTER(_VP)]),
    "rd_exporter_destroy": (None, [_VP]),
    "rd_exporter_submit": (_I, [_VP, C.POINTER(RdFrame), C.POINTER(_U32)]),
    "rd_exporter_submit_host": (_I, [_VP, C.POINTER(RdFrame), _VP, C.POINTER(_U32)]),
    "rd_exporter_wait": (_I, [_VP, _U32, C.POINTER(_VP), C.POINTER(_SZ)]),
    "rd_exporter_release": (_I, [_VP, _U32]),
    "rd_selftest_q8": (_I, [_I, C.POINTER(C.c_uint64), C.POINTER(_U32), C.POINTER(C.c_uint64), C.POINTER(C.c_float)]),
    "rd_selftest_q8_codes": (_I, [_I, _U32, _U32, _VP]),
    "rd_selftest_q8_lut": (_I, [_I, C.POINTER(C.c_uint64), C.POINTER(_U32)]),
    "rd_selftest_q8_lut_codes": (_I, [_I, _U32, _U32, _VP]),
    "rd_q8_lut_table": (_I, [_VP, _SZ]),
    "rd_selftest_f16": (_I, [_I, C.POINTER(C.c_uint64), C.POINTER(_U32), C.POINTER(C.c_uint64)]),
    "rd_selftest_f16_halves": (_I, [_I, _U32, _U32, _VP]),
    "rd_selftest_f16_lut": (_I, [_I, C.POINTER(C.c_uint64), C.POINTER(_U32), C.POINTER(C.c_uint64)]),
    "rd_selftest_f16_lut_values": (_I, [_I, _U32, _U32, _VP]),
    "rd_f16_lut_tables": (_I, [_VP, _SZ, _VP, _SZ]),
    "rd_ljpeg_decode": (_I, [_VP, _SZ, _VP, _SZ, C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_U32), C.POINTER(_U32)]),
    "rd_host_alloc": (_I, [_I, _SZ, C.POINTER(_VP)]),
    "rd_host_free": (_I, [_I, _VP]),
    "rd_measure_hbm": (_I, [_I, _SZ, _U32] + [C.POINTER(C.c_double)] * 4),
    "rd_measure_valu": (_I, [_I, C.POINTER(C.c_double)]),
    "rd_device_malloc": (_I, [_I, _SZ, C.POINTER(_VP)]),
    "rd_device_free": (_I, [_I, _VP]),
    "rd_device_memory": (_I, [_I, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "rd_memcpy_h2d": (_I, [_I, _VP, _VP, _SZ]),
    "rd_memcpy_d2h": (_I, [_I, _VP, _VP, _SZ]),
    "rd_device_synchronize": (_I, [_I]),
    "rd_stream_create": (_I, [_I, C.POINTER(_VP)]),
    "rd_stream_synchronize": (_I, [_I, _VP]),
    "rd_stream_destroy": (_I, [_I, _VP]),
    "rd_debug_poison_scheduler": (_I, [_VP, _VP]),
    "rd_debug_scheduler_entries": (_U32, [_VP]),
    "rd_debug_lane_count": (_U32, [_VP]),
    "rd_debug_is_pinned_host": (_I, [_VP, _SZ]),
    "rd_debug_node_histogram_of": (_I, [_VP, _U32, _VP]),
    "rd_debug_inject_fault": (_I, [C.c_char_p, _U32, _U32]),
}

_lib = None


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME
    as /opt/rocm's); whichever is mapped first serves both.  If torch is installed, map ITS copy
    first so a later `import torch` (bench.py, torch.distributed) shares our runtime instead of
    bringing up a second one.  RAWDEV_HIP_RUNTIME=system skips this."""
    if os.environ.get("RAWDEV_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load librawdev.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        path = os.environ.get("RAWDEV_LIB") or LIB_PATH          # RAWDEV_LIB: an A/B build of the same ABI (tools/)
        if not os.path.exists(path):
            raise RawdevError(-5, f"{path} is missing: run `python -m raweditor_amd.build` "
                                  "(hipcc --offload-arch=gfx950); librawdev has no CPU fallback")
        _preload_hip_runtime()
        L = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)          # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        if L.rd_abi_version() != ABI_VERSION:
            raise RawdevError(-5, f"ABI version {L.rd_abi_version()} != {ABI_VERSION}")
        _lib = L
    return _lib


def check(code: int) -> None:
    if code != RD_OK:
        raise RawdevError(code, lib().rd_last_error().decode("utf-8", "replace"))


def inject_fault(site, kind: int = FAULT_BAD_ALLOC, after: int = 0) -> None:
    """Test hook (rd_debug_inject_fault): the (after + 1)-th passage through fault point `site` throws a C++ exception of
    `kind` inside the library, once; the entry point must turn it into a status.  site None disarms."""
    check(lib().rd_debug_inject_fault(site.encode() if site else None, int(kind), int(after)))


def device_count() -> int:
    n = C.c_int(0)
    try:
        check(lib().rd_device_count(C.byref(n)))
    except RawdevError:
        return 0
    return n.value
