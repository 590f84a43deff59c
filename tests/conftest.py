import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")


@pytest.fixture(scope="session")
def emu():
    """Host emulation build of the HIP kernels (tests/emu) bound through the same ctypes prototypes."""
    from editanything_amd.csrc import build
    from editanything_amd import _lib
    path = build.build_emu(verbose=False)
    return _lib.bind(path)
