"""The C++ host mirror (include/rawdev.hpp): built with g++ against librawdev.so + the oracle, run as a
subprocess.  CPU run: EditParams tests (the reference's own, edit.rs:129-163) + ABI error paths;
-m gpu run: the RenderPipeline surface against the oracle."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_host_mirror")


def _build(refc):
    from raweditor_amd import _lib
    _lib.lib()
    src = os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp")
    deps = [src, os.path.join(ROOT, "include", "rawdev.hpp"), os.path.join(ROOT, "include", "rawdev.h")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) > os.path.getmtime(d) for d in deps):
        return EXE
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pthread", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "oracle"), src, "-o", EXE,
           os.path.join(ROOT, "raweditor_amd", "librawdev.so"), os.path.join(ROOT, "oracle", "libdevelop_ref.so"),
           "-Wl,-rpath," + os.path.join(ROOT, "raweditor_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return EXE


def test_cpp_mirror_cpu(refc):
    out = subprocess.run([_build(refc)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok (0 failure(s)) [cpu]" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_gpu(refc):
    out = subprocess.run([_build(refc), "--gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok (0 failure(s)) [gpu]" in out.stdout
