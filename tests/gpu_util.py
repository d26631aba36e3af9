"""Helpers for the -m gpu tests: device buffers through the C ABI's plumbing calls."""
import ctypes as C

import numpy as np

from raweditor_amd import _lib
from raweditor_amd._lib import check


class DevBuf:
    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(_lib.lib().rd_device_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    @classmethod
    def from_array(cls, a, device=0):
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes, device)
        check(_lib.lib().rd_memcpy_h2d(device, C.c_void_p(b.ptr), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return b

    def to_array(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(_lib.lib().rd_memcpy_d2h(self.device, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(_lib.lib().rd_device_free(self.device, C.c_void_p(self.ptr)))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync(device=0):
    check(_lib.lib().rd_device_synchronize(device))
