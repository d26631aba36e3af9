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
