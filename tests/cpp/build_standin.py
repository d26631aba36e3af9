"""Builds tests/cpp/rccl_standin.cpp -> tests/cpp/librccl_standin.so (host code over the HIP runtime; hipcc, a few seconds).

Test infrastructure: the -m gpu suite drives the RCCL branch of rd_node_batch_* with several ranks on ONE GPU through this
stand-in (RD_NODE_REDUCE=standin + RAWDEV_RCCL_LIB=<this file>).  No pytest import here, so __graft_entry__.build() can call it
without the test package's dependencies; a failure to build it is a warning there (the GPU test builds it on demand)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
STANDIN_SRC = os.path.join(HERE, "rccl_standin.cpp")
STANDIN_SO = os.path.join(HERE, "librccl_standin.so")


def build_rccl_standin() -> str:
    if not os.path.exists(STANDIN_SO) or os.path.getmtime(STANDIN_SO) < os.path.getmtime(STANDIN_SRC):
        hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "-O1", "-fPIC", "-shared", "-o", STANDIN_SO, STANDIN_SRC], check=True)
    return STANDIN_SO


if __name__ == "__main__":
    print(build_rccl_standin())
