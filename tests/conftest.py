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


def pytest_sessionfinish(session, exitstatus):
    """The largest per-pixel relL2 every parity assertion saw, per test (tests/test_gpu_parity.py OBSERVED): the tolerance is a
    bound, this is the measurement. Written where gpurun collects files; nothing reads it back."""
    import json

    mod = sys.modules.get("test_gpu_parity")
    observed = getattr(mod, "OBSERVED", None)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if observed and os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_observed.json"), "w") as f:
            json.dump({"max_rel_l2": max(observed.values()), "tests": len(observed),
                       "fuzz_scenes_by_kernel_form": getattr(sys.modules.get("test_gpu_fuzz"), "KERNEL_FORMS", None) or None,
                       "per_test": dict(sorted(observed.items()))}, f, indent=1)
