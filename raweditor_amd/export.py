"""Export feed -- the step after the path.

The reference exports one frame per click: `render_full_res_to_bytes` renders, copies the texture to a
MAP_READ buffer, blocks in `device.poll(Wait)`, de-pads 96 MB row by row (src/gpu/pipeline.rs:552-605,
"1-2 seconds for 24MP"), then strips alpha on the CPU for JPEG (src/main.rs:1777-1786) before handing the
bytes to the `image` crate.  `Exporter` is the GPU half of that for a stream of frames: a ring of pinned
host buffers filled by asynchronous D2H copies on a second stream (the copy of frame i overlaps the kernel
of frame i+1), RGBA8 for the PNG path or RGB8 (alpha strip fused into the kernel) for the JPEG path.
Encoding itself (the `image` crate) stays out of scope.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Iterator, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import FMT_RGB_U8, FMT_RGBA_U8, BYTES_PER_PIXEL, MATH_STRICT, RdFrame, check
from .edit import EditParams


class Exporter:
    def __init__(self, device: int, width: int, height: int, fmt: int = FMT_RGBA_U8, n_slots: int = 2,
                 math_mode: int = MATH_STRICT):
        self._h = C.c_void_p()
        self.width, self.height, self.fmt, self.n_slots = int(width), int(height), int(fmt), int(n_slots)
        check(_lib.lib().rd_exporter_create(device, self.width, self.height, self.fmt, int(math_mode), self.n_slots,
                                            C.byref(self._h)))
        self._dtype = {0: np.float32, 1: np.float16}.get(self.fmt, np.uint8)
        self._channels = 3 if self.fmt == FMT_RGB_U8 else 4

    @staticmethod
    def frame(cfa_dev: int, params: EditParams, wb: Sequence[float], cm: Sequence[float], black_level: int = 0) -> RdFrame:
        f = RdFrame()
        f.cfa_dev, f.out_dev = cfa_dev, None
        f.params = params.to_c()
        f.wb_multipliers[:] = [float(x) for x in wb]
        f.color_matrix[:] = [float(x) for x in cm]
        f.black_level = int(black_level)
        return f

    def submit(self, frame: RdFrame) -> int:
        slot = C.c_uint32()
        check(_lib.lib().rd_exporter_submit(self._h, C.byref(frame), C.byref(slot)))
        return slot.value

    def submit_host(self, cfa, frame: RdFrame) -> int:
        """The frame's CFA plane from HOST memory (rd_exporter_submit_host): `cfa` is a contiguous uint16 array of
        width*height samples -- a PinnedBytes view is read by the DMA engine in place (keep it untouched until wait()),
        anything else is staged and free again when this returns.  frame.cfa_dev is ignored."""
        a = np.ascontiguousarray(cfa, dtype=np.uint16).reshape(-1)
        if a.size != self.width * self.height:
            raise _lib.RawdevError(-1, f"cfa has {a.size} samples, expected {self.width}x{self.height}")
        slot = C.c_uint32()
        check(_lib.lib().rd_exporter_submit_host(self._h, C.byref(frame), a.ctypes.data_as(C.c_void_p), C.byref(slot)))
        return slot.value

    def wait(self, slot: int) -> np.ndarray:
        """(h, w, c) view of the slot's pinned host buffer; valid until release(slot)."""
        data, n = C.c_void_p(), C.c_size_t()
        check(_lib.lib().rd_exporter_wait(self._h, int(slot), C.byref(data), C.byref(n)))
        buf = (C.c_uint8 * n.value).from_address(data.value)
        return np.frombuffer(buf, dtype=self._dtype).reshape(self.height, self.width, self._channels)

    def release(self, slot: int) -> None:
        check(_lib.lib().rd_exporter_release(self._h, int(slot)))

    def export(self, frames: Iterable[RdFrame]) -> Iterator[Tuple[int, np.ndarray]]:
        """Pipeline a stream of frames through the ring: yields (index, surface view); the view is released
        when the consumer asks for the next one (copy it, or encode it, before advancing)."""
        pending = []                       # (index, slot)
        for i, fr in enumerate(frames):
            if len(pending) == self.n_slots:
                j, s = pending.pop(0)
                yield j, self.wait(s)
                self.release(s)
            pending.append((i, self.submit(fr)))
        for j, s in pending:
            yield j, self.wait(s)
            self.release(s)

    def export_host(self, frames: Iterable[Tuple[np.ndarray, RdFrame]]) -> Iterator[Tuple[int, np.ndarray]]:
        """export() for (cfa plane in host memory, frame) pairs: upload, develop and read-back of neighbouring frames overlap."""
        pending = []
        for i, (cfa, fr) in enumerate(frames):
            if len(pending) == self.n_slots:
                j, s = pending.pop(0)
                yield j, self.wait(s)
                self.release(s)
            pending.append((i, self.submit_host(cfa, fr)))
        for j, s in pending:
            yield j, self.wait(s)
            self.release(s)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().rd_exporter_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
