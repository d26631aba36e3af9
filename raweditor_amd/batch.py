"""Batch export: whole frames sharded over the GPUs of a node, one process per GPU.

The reference has no batch path (it exports one frame per click, src/main.rs:1744-1799); this is
the north-star's addition.  Frames share nothing -- the demosaic stencil clamps at the frame edge
(shaders.rs:163-166) -- so frame i simply belongs to rank i mod N and no pixel data ever crosses
xGMI.  The only exchange is the global histogram: one all-reduce of 768 x u64 per batch
(u64 because 2048 x 24 MP overflows u32), done with torch.distributed (backend "nccl" = RCCL on
ROCm, "gloo" in the CPU tests).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

from . import _lib
from ._lib import RdFrame, check
from .edit import EditParams


def shard_frames(n_frames: int, rank: int, world_size: int) -> List[int]:
    """Static round-robin: frame i -> rank i mod world_size."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    return list(range(rank, n_frames, world_size))


def allreduce_histogram(hist64):
    """Sum a (768,) or (3,256) int64 tensor over all ranks in place (no-op without a process group).
    int64 carries the u64 counts bit-exactly for any realistic batch (< 2^63 pixels)."""
    import torch
    import torch.distributed as dist
    if hist64.dtype != torch.int64:
        raise TypeError("histogram tensor must be int64")
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(hist64, op=dist.ReduceOp.SUM)
    return hist64


class NodeBatch:
    """rd_node_batch: the batch path over several GPUs of one node from ONE process (what a Rust / C host uses).
    Frame i belongs to devices[i mod N]; the global histogram is one RCCL all-reduce of 768 x u64."""

    def __init__(self, devices: Sequence[int], width: int, height: int, fmt: int, with_histogram: bool = True,
                 math_mode: int = _lib.MATH_STRICT):
        self._h = C.c_void_p()
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        check(_lib.lib().rd_node_batch_create(arr, len(self.devices), int(width), int(height), int(fmt),
                                              1 if with_histogram else 0, C.byref(self._h)))
        if math_mode != _lib.MATH_STRICT:
            check(_lib.lib().rd_node_batch_set_math_mode(self._h, int(math_mode)))

    def device_of(self, frame_index: int) -> int:
        """The device (not the index into `devices`) that owns frame `frame_index`."""
        return self.devices[_lib.lib().rd_node_batch_device_of(len(self.devices), int(frame_index))]

    def develop(self, frames, row_bands: int = 1) -> None:
        check(_lib.lib().rd_node_batch_develop(self._h, frames, len(frames), int(row_bands)))

    def histogram(self):
        """Global (3, 256) uint64 histogram of everything developed since the last call; synchronises."""
        import numpy as np
        out = np.zeros(768, np.uint64)
        check(_lib.lib().rd_node_batch_histogram(self._h, out.ctypes.data_as(C.c_void_p)))
        return out.reshape(3, 256)

    def histogram_enqueue(self) -> None:
        """Fold + all-reduce + read-back enqueued on the devices' streams; nothing is waited for."""
        check(_lib.lib().rd_node_batch_histogram_enqueue(self._h))

    def histogram_fetch(self):
        """(3, 256) uint64 counts of everything developed since the last fetch (every histogram_enqueue() in between
        adds its interval); waits for the last read-back only."""
        import numpy as np
        out = np.zeros(768, np.uint64)
        check(_lib.lib().rd_node_batch_histogram_fetch(self._h, out.ctypes.data_as(C.c_void_p)))
        return out.reshape(3, 256)

    def synchronize(self) -> None:
        check(_lib.lib().rd_node_batch_synchronize(self._h))

    def stream(self, index: int) -> int:
        """The hipStream_t (as int) devices[index]'s share is enqueued on (measurement aid: record events there)."""
        return int(_lib.lib().rd_node_batch_stream(self._h, int(index)) or 0)

    def last_launch_count(self, index: int = 0) -> int:
        return int(_lib.lib().rd_node_batch_last_launch_count(self._h, int(index)))

    REDUCE_KINDS = {0: "none (one device)", 1: "rccl all-reduce", 2: "host fold (RD_NODE_REDUCE=host)"}

    def reduce_kind(self) -> str:
        return self.REDUCE_KINDS.get(int(_lib.lib().rd_node_batch_reduce_kind(self._h)), "?")

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().rd_node_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchExporter:
    """rd_batch: fused demosaic+develop(+histogram) launches for same-sized frames on one device."""

    def __init__(self, device: int, width: int, height: int, fmt: int, with_histogram: bool = True,
                 math_mode: int = _lib.MATH_STRICT):
        self._h = C.c_void_p()
        self.device, self.width, self.height, self.fmt = device, int(width), int(height), int(fmt)
        self.with_histogram = bool(with_histogram)
        check(_lib.lib().rd_batch_create(device, self.width, self.height, self.fmt,
                                         1 if with_histogram else 0, C.byref(self._h)))
        if math_mode != _lib.MATH_STRICT:
            check(_lib.lib().rd_batch_set_math_mode(self._h, int(math_mode)))

    @staticmethod
    def make_frames(cfa_ptrs: Sequence[int], out_ptrs: Sequence[int], params: Sequence[EditParams],
                    wb: Sequence[float], cm: Sequence[float], black_level: int = 0, matrix_layout: int = 0):
        """Build the rd_frame array once (device pointers as ints); reuse it for every step."""
        n = len(cfa_ptrs)
        if not (len(out_ptrs) == len(params) == n):
            raise ValueError("cfa_ptrs, out_ptrs and params must have the same length")
        arr = (RdFrame * n)()
        for i in range(n):
            arr[i].cfa_dev = cfa_ptrs[i]
            arr[i].out_dev = out_ptrs[i]
            arr[i].params = params[i].to_c()
            arr[i].wb_multipliers[:] = [float(x) for x in wb]
            arr[i].color_matrix[:] = [float(x) for x in cm]
            arr[i].black_level = int(black_level)
            arr[i].matrix_layout = int(matrix_layout)
        return arr

    def develop(self, frames, row_bands: int = 1, stream: int = 0) -> None:
        """Enqueue the frames on `stream` (multi-frame launches by default, see rd_batch_develop); not synchronised."""
        check(_lib.lib().rd_batch_develop(self._h, frames, len(frames), int(row_bands),
                                          C.c_void_p(stream) if stream else None))

    def last_launch_count(self) -> int:
        """Fused kernel launches the last develop() enqueued (multi-frame launches cover up to 8 frames each)."""
        return int(_lib.lib().rd_batch_last_launch_count(self._h))

    def set_launch_timing(self, keep_calls: int) -> None:
        """Measurement aid: keep a HIP event pair around every fused launch of the last `keep_calls` develop() /
        probe_pattern() calls (0 = off).  The pairs sit between the launches: look at a timed call, quote an untimed one."""
        check(_lib.lib().rd_batch_set_launch_timing(self._h, int(keep_calls)))

    def launch_timeline(self):
        """After the stream has been synchronised: [(call, start_us, end_us), ...] of the kept launches, oldest first
        (times since the first kept launch's start; call 0 = the oldest kept call)."""
        n = C.c_uint32(0)
        check(_lib.lib().rd_batch_launch_timeline(self._h, None, None, None, 0, C.byref(n)))
        if not n.value:
            return []
        st, en, ca = (C.c_float * n.value)(), (C.c_float * n.value)(), (C.c_uint32 * n.value)()
        check(_lib.lib().rd_batch_launch_timeline(self._h, st, en, ca, n.value, C.byref(n)))
        return [(int(ca[i]), float(st[i]), float(en[i])) for i in range(n.value)]

    def probe_pattern(self, frames, stream: int = 0) -> None:
        """Measurement aid: develop()'s launches for these frames with the arithmetic removed (rd_batch_probe_pattern) --
        the surfaces receive the raw samples as floats.  RGBA-f32 contexts only."""
        check(_lib.lib().rd_batch_probe_pattern(self._h, frames, len(frames), C.c_void_p(stream) if stream else None))

    def measure_clock(self, frames, stream: int = 0):
        """Measurement aid (rd_batch_measure_clock): one ordinary develop of `frames` through the kernel instance that stamps
        its clocks, synchronised -> (GHz median, min, max over the last launch's workgroups, median workgroup busy us)."""
        v = [C.c_double(0.0) for _ in range(4)]
        check(_lib.lib().rd_batch_measure_clock(self._h, frames, len(frames), C.c_void_p(stream) if stream else None,
                                                *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def histogram(self, hist_dev: int, stream: int = 0) -> None:
        """Fold the accumulated counts into a device u64[768] and reset the accumulator."""
        check(_lib.lib().rd_batch_histogram(self._h, C.c_void_p(hist_dev), C.c_void_p(stream) if stream else None))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().rd_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
