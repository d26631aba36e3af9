"""CPU: nothing thrown inside librawdev.so crosses the C ABI (SURVEY.md section 8b "Errors": never abort, never throw across
the ABI; the reference's constructor returns Result<Self, String>, pipeline.rs:122, :156, :169).

  * the source: every extern "C" definition in raweditor_amd/csrc is a function-try-block that ends in RD_CATCH_* with its
    own name, and there are as many of them as include/rawdev.h declares;
  * the behaviour, without a GPU: a C++ exception injected at an entry point (rd_debug_inject_fault, or RD_FAULT_INJECT in
    the environment) comes back as a status + message -- std::bad_alloc as RD_ERR_OOM, std::system_error (a thread that cannot
    start), std::runtime_error and a type outside std::exception as RD_ERR_INTERNAL -- for int, void and value-returning
    entry points; the process is still there and the next call behaves normally.  (The real allocation and thread-start
    sites need a device: tests/test_gpu_faults.py.)
"""
import ctypes as C
import glob
import os
import re
import subprocess
import sys

import pytest

import raweditor_amd as ra
from raweditor_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raweditor_amd", "csrc")


def _extern_c_definitions():
    """(file, name, signature-to-body text) of every extern "C" definition under csrc/."""
    found = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.inl")) + glob.glob(os.path.join(CSRC, "*.h"))):
        text = open(path).read()
        for m in re.finditer(r'^extern "C"[^;{]*?\b(rd_[a-z0-9_]+)\s*\(', text, flags=re.M):
            end = text.find("\nRD_CATCH_", m.start())
            one_line_end = text.find("\n", m.start())
            found.append((os.path.basename(path), m.group(1), text[m.start():one_line_end], text[m.start():end + 200 if end >= 0 else m.start() + 200]))
    return found


def test_every_entry_point_is_a_function_try_block():
    defs = _extern_c_definitions()
    names = [d[1] for d in defs]
    assert len(names) == len(set(names))
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rawdev.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(rd_[a-z0-9_]+)\s*\(", header)))
    assert sorted(names) == declared, "csrc defines other extern \"C\" symbols than include/rawdev.h declares"
    for fname, name, first_line, chunk in defs:
        # `) try` ends the signature (possibly a line or two further down), and the handler names the same function
        sig_end = re.search(r"\)\s*try\b", chunk)
        assert sig_end, f"{fname}: {name} is not a function-try-block"
        before = chunk[:sig_end.start()]
        assert "{" not in before, f"{fname}: {name}: a body opens before `try`"
        assert re.search(r"RD_CATCH_(INT|VOID|VAL)\(" + name + r"[,)]", chunk[sig_end.end():]) or \
            re.search(r"\}\s*RD_CATCH_(INT|VOID|VAL)\(" + name + r"[,)]", first_line), f"{fname}: {name} has no RD_CATCH_* of its own"
    # the handler macros themselves catch everything
    src = open(os.path.join(CSRC, "rawdev.hip")).read()
    for macro in ("RD_CATCH_INT", "RD_CATCH_VOID", "RD_CATCH_VAL"):
        assert re.search(r"#define " + macro + r"\([^)]*\) catch \(\.\.\.\)", src), macro
    # and no std::thread is constructed outside a try (the round-4 hazard: a joinable thread destroyed by an unwinding vector)
    for path in glob.glob(os.path.join(CSRC, "*")):
        if not os.path.isfile(path):
            continue
        t = open(path).read()
        assert "emplace_back([" not in t or "std::thread" not in t.split("emplace_back([")[0][-200:], path


@pytest.fixture()
def disarm():
    yield
    _lib.inject_fault(None, 0)


CASES = [(_lib.FAULT_BAD_ALLOC, _lib.RD_ERR_OOM, "std::bad_alloc"), (_lib.FAULT_THREAD_START, _lib.RD_ERR_INTERNAL, "thread-start"),
         (_lib.FAULT_RUNTIME, _lib.RD_ERR_INTERNAL, "injected fault at"), (_lib.FAULT_FOREIGN, _lib.RD_ERR_INTERNAL, "unknown C++ exception")]


@pytest.mark.parametrize("kind,code,text", CASES)
def test_injected_exceptions_become_statuses(disarm, kind, code, text):
    L = _lib.lib()
    p = _lib.RdEditParams()
    L.rd_edit_params_default(C.byref(p))
    wb = (C.c_float * 4)(2.0, 1.0, 1.5, 1.0)
    cm = (C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    cfa = (C.c_uint16 * 64)()
    out = C.c_void_p()
    calls = {
        "rd_pipeline_create": lambda: L.rd_pipeline_create(0, 1, cfa, 8, 8, C.byref(p), wb, cm, C.byref(out)),
        "rd_batch_develop": lambda: L.rd_batch_develop(None, None, 0, 1, None),
        "rd_node_batch_develop": lambda: L.rd_node_batch_develop(None, None, 0, 1),
        "rd_render": lambda: L.rd_render(None, 8, 8, 0, None, 0, None),
        "rd_exporter_create": lambda: L.rd_exporter_create(0, 128, 8, 2, 0, 2, C.byref(out)),
    }
    destroy = {"rd_pipeline_create": L.rd_pipeline_destroy, "rd_exporter_create": L.rd_exporter_destroy}

    def settle(name):                                             # (on a machine WITH a gfx950 device the creates succeed: tidy up)
        if out.value and name in destroy:
            destroy[name](out)
        out.value = None

    for name, call in calls.items():
        normal = call()                                           # what the call says without a fault (no device / NULL argument)
        assert normal in (_lib.RD_OK, _lib.RD_ERR_INVALID_ARG, _lib.RD_ERR_NO_DEVICE, _lib.RD_ERR_HIP, _lib.RD_ERR_UNSUPPORTED), (name, normal)
        assert (normal == _lib.RD_OK) == bool(out.value), (name, normal)
        settle(name)
        _lib.inject_fault(name, kind)
        rc = call()
        msg = L.rd_last_error().decode()
        assert rc == code, (name, rc, msg)
        assert msg.startswith(name + ":") and text in msg, msg
        assert not out.value, f"{name}: a failed create must not hand out a handle"
        assert call() == normal, f"{name}: the fault is one-shot"
        settle(name)


def test_void_and_value_entry_points_swallow_and_report(disarm):
    L = _lib.lib()
    _lib.inject_fault("rd_format_bytes_per_pixel", _lib.FAULT_RUNTIME)
    assert L.rd_format_bytes_per_pixel(ra.FMT_RGBA_F32) == 0
    assert "rd_format_bytes_per_pixel" in L.rd_last_error().decode()
    assert L.rd_format_bytes_per_pixel(ra.FMT_RGBA_F32) == 16
    p = _lib.RdEditParams(*([7.0] * 10))
    _lib.inject_fault("rd_edit_params_default", _lib.FAULT_FOREIGN)
    L.rd_edit_params_default(C.byref(p))                          # void: the exception ends here, the struct is untouched
    assert p.whites == 7.0 and "rd_edit_params_default" in L.rd_last_error().decode()
    L.rd_edit_params_default(C.byref(p))
    assert p.whites == 1.0 and p.exposure == 0.0
    _lib.inject_fault("rd_node_batch_stream", _lib.FAULT_BAD_ALLOC)
    assert L.rd_node_batch_stream(None, 0) is None


def test_any_site_and_countdown(disarm):
    L = _lib.lib()
    n = C.c_int()
    _lib.inject_fault("*", _lib.FAULT_BAD_ALLOC, after=2)         # the third fault point passed, whichever it is
    assert L.rd_device_count(C.byref(n)) in (0, _lib.RD_ERR_NO_DEVICE)
    assert L.rd_format_bytes_per_pixel(2) == 4
    assert L.rd_device_count(C.byref(n)) == _lib.RD_ERR_OOM
    assert L.rd_device_count(C.byref(n)) in (0, _lib.RD_ERR_NO_DEVICE)
    with pytest.raises(_lib.RawdevError):
        _lib.inject_fault("x" * 80, 1)
    with pytest.raises(_lib.RawdevError):
        _lib.inject_fault("rd_render", 9)
    _lib.inject_fault("rd_render", _lib.FAULT_RUNTIME)
    _lib.inject_fault(None, 0)                                    # disarmed again
    assert L.rd_render(None, 8, 8, 0, None, 0, None) == _lib.RD_ERR_INVALID_ARG


def test_fault_from_the_environment():
    code = ("import ctypes as C; from raweditor_amd import _lib; L = _lib.lib(); n = C.c_int(); "
            "rc = L.rd_device_count(C.byref(n)); print(rc, L.rd_last_error().decode()); "
            "print(L.rd_device_count(C.byref(n)) in (0, -2))")
    env = dict(os.environ, RD_FAULT_INJECT="rd_device_count:3")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    first, second = out.stdout.strip().splitlines()[-2:]
    assert first.startswith("-6 rd_device_count:") and "injected fault at rd_device_count" in first, out.stdout
    assert second == "True"


def test_a_compiled_c_host_sees_statuses_not_exceptions(tmp_path):
    """The same from C (no C++ runtime on the caller's side to unwind through): link against librawdev.so, inject, call."""
    src = r"""
#include <stdio.h>
#include <string.h>
#include "rawdev.h"
int main(void) {
    rd_edit_params p; rd_edit_params_default(&p);
    float wb[4] = {2, 1, 1.5f, 1}, cm[9] = {1,0,0, 0,1,0, 0,0,1};
    uint16_t cfa[64] = {0};
    rd_pipeline *pipe = (rd_pipeline *)0;
    int bad = 0;
    for (unsigned kind = 1; kind <= 4; ++kind) {
        if (rd_debug_inject_fault("rd_pipeline_create", kind, 0) != RD_OK) return 2;
        int rc = rd_pipeline_create(0, 1, cfa, 8, 8, &p, wb, cm, &pipe);
        int want = kind == RD_FAULT_BAD_ALLOC ? RD_ERR_OOM : RD_ERR_INTERNAL;
        printf("kind %u: rc %d (%s)\n", kind, rc, rd_last_error());
        if (rc != want || pipe || !strstr(rd_last_error(), "rd_pipeline_create:")) bad = 1;
    }
    if (rd_debug_inject_fault("rd_batch_destroy", RD_FAULT_RUNTIME, 0) != RD_OK) return 2;
    rd_batch_destroy((rd_batch *)0);                        /* destroy entry points carry no fault point: nothing fires */
    rd_debug_inject_fault((const char *)0, 0, 0);
    return bad;
}
"""
    c = tmp_path / "host.c"
    c.write_text(src)
    exe = tmp_path / "host"
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(c), "-L", lib_dir, "-l:librawdev.so",
                    "-Wl,-rpath," + lib_dir], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("rc -") == 4


def test_measure_hbm_refuses_sizes_its_kernels_cannot_cover():
    """rd_measure_hbm: sizes below one whole step of the copy grid used to give zero-byte buffers, and sizes
    that are not whole steps let waves run past their range; the size check comes before the device check."""
    L = _lib.lib()
    v = [C.c_double() for _ in range(4)]
    for nbytes in (0, 64 << 20, 256 << 20, (512 << 20) - 1):
        assert L.rd_measure_hbm(0, nbytes, 5, *[C.byref(x) for x in v]) == _lib.RD_ERR_INVALID_ARG
        assert "512 MiB" in L.rd_last_error().decode()
    assert L.rd_measure_hbm(0, 1 << 30, 0, *[C.byref(x) for x in v]) == _lib.RD_ERR_INVALID_ARG
