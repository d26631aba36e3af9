#!/usr/bin/env python3
"""round 5: what rd_host_memory_kind says about host ranges page-locked with hipHostRegister -- a whole
registration, its inside, and a range whose two ENDS are registered but whose middle is not (two registrations with a gap).
    python tools/probe_registered_host.py        (on the GPU box)"""
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import raweditor_amd as ra
from raweditor_amd import _lib
L = _lib.lib()
path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)     # the runtime this process already has mapped
hip = C.CDLL(path)
n = 64 << 20
a = np.zeros(n + 8192, np.uint8)
base = (a.ctypes.data + 4095) & ~4095
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
print("before register:", L.rd_debug_is_pinned_host(C.c_void_p(base), n))
rc = hip.hipHostRegister(C.c_void_p(base), n, 0)
print("hipHostRegister rc", rc)
print("registered range:", L.rd_debug_is_pinned_host(C.c_void_p(base), n))
print("inside:", L.rd_debug_is_pinned_host(C.c_void_p(base + 4096), n - 8192))
# two adjacent registrations with a gap
b = np.zeros(3 * (1 << 20) + 8192, np.uint8)
bb = (b.ctypes.data + 4095) & ~4095
print("reg A", hip.hipHostRegister(C.c_void_p(bb), 1 << 20, 0), "reg C", hip.hipHostRegister(C.c_void_p(bb + (2 << 20)), 1 << 20, 0))
print("range spanning the gap (ends registered, middle not):", L.rd_debug_is_pinned_host(C.c_void_p(bb), 3 << 20))
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostUnregister(C.c_void_p(base)); hip.hipHostUnregister(C.c_void_p(bb)); hip.hipHostUnregister(C.c_void_p(bb + (2 << 20)))
