import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 0x52415745  # "RAWE" 


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A fresh checkout has no librawdev.so (built artefacts are git-ignored): build it once (hipcc cross-compiles
    # gfx950 without a GPU, ~10 s).  If that is impossible the tests that need the library fail loudly.
    try:
        from raweditor_amd.build import build_library
        build_library()
    except Exception as e:  # noqa: BLE001
        print(f"[conftest] could not build librawdev.so: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def refc():
    """The C oracle (built on demand with gcc)."""
    from oracle import ref_c
    ref_c.lib()
    return ref_c


@pytest.fixture()
def rng():
    return np.random.default_rng(SEED)


@pytest.fixture(scope="session")
def gpu_lib():
    """librawdev.so on a machine with a gfx950 device; fails (not skips) if the extension is missing."""
    import raweditor_amd as ra
    from raweditor_amd import _lib
    _lib.lib()  # raises if the .so is not built -- the product has no fallback
    assert ra.device_count() >= 1, "no HIP device visible on a -m gpu run"
    return ra
