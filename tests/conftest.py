"""pytest configuration: import paths + the `gpu` marker.

`-m "not gpu"` = oracle vs golden vectors, host logic, C-ABI export check (runs anywhere).
`-m gpu`       = parity of the HIP path (through the C-ABI) against the oracle; needs an MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "custom-diffusion360_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def tune():
    """tune(gemm_cfg=4, ...) overrides tiling / kernel choices of the C ABI (cd360_set_tuning) for one test; every field is restored
    afterwards.  The C side reads no environment variables."""
    from cd360 import _lib
    saved = _lib.get_tuning()
    yield _lib.set_tuning
    _lib.set_tuning(**saved)
