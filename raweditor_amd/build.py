"""Build librawdev.so (the HIP kernels + C ABI) in-tree for gfx950.

`python -m raweditor_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.
-ffp-contract=off is part of the numerical contract: the colour stack's operation
order is the WGSL text's, and only the explicit FMAs of the pinned pow pair are fused.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "librawdev.so")
SOURCES = ["rawdev.hip"]
HEADERS = ["rd_math.h", "rd_uniforms.h", "rd_kernels.h", os.path.join("..", "..", "include", "rawdev.h")]
# -fno-slp-vectorize: v_pk_*_f32 has no throughput advantage on gfx950 (measured), and a packed
# operand pair that happens to contain a pending load's register makes hipcc drain the store queue
# mid-loop (false vmcnt dependency); scalar f32 code keeps the s_waitcnt placement exact.
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-std=c++17", "-Wall", "-Wextra"]


def find_hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
