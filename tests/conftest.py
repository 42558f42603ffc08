import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_lib():
    """The C-ABI library with a device behind it; GPU tests fail (not skip) when the HIP extension is missing."""
    from pyrite_amd import _lib

    lib = _lib.lib()
    assert lib.pyr_device_count() >= 1, "no HIP device visible: -m gpu tests must run on the GPU box"
    return lib
