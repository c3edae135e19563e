import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """ctypes binding of the CPU oracle (test infrastructure only)."""
    from oracle import oracle_ctypes
    oracle_ctypes.lib()
    return oracle_ctypes


@pytest.fixture(scope="session")
def hip():
    """The product package; the HIP library must load and see a device (no fallback)."""
    import ctypes as C
    import lexls_amd
    from lexls_amd import capi
    n = C.c_int()
    capi.check(capi.lib().lexls_device_count(C.byref(n)))
    return lexls_amd
