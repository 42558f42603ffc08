"""Loader for the C-ABI library (pyrite_amd/csrc/libpyrite_gpu.so).

There is no CPU fallback: if the library is missing or a symbol is absent this raises, loudly. The library is
built in-tree by `__graft_entry__.build()` / `python -m pyrite_amd.build`."""
import ctypes
import os

from . import abi

# PYRITE_GPU_LIB selects another build of the same library (A/B experiments on kernel variants); never a CPU path.
LIB_PATH = os.environ.get("PYRITE_GPU_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libpyrite_gpu.so")
_lib = None


class PyriteGpuError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("pyrite_gpu error %d: %s" % (status, message))
        self.status = status


def _share_torch_hip_runtime():
    """One HIP runtime per process. PyTorch-ROCm ships its own libamdhip64.so.7 (rpath $ORIGIN) and libpyrite_gpu.so is linked
    against /opt/rocm's copy of the same soname: whichever is loaded first serves both, and torch on top of the system copy
    finds no device ("No HIP GPUs are available", seen when a process rendered first and imported torch afterwards). The
    python layer hands torch's device pointers and streams to the library (renderer.render_device, distributed.py), so when
    torch is installed its runtime is the one to load, whatever the import order. Without torch the system copy is used."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: the HIP extension has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`). "
                "pyrite_amd has no CPU fallback." % LIB_PATH
            )
        _share_torch_hip_runtime()
        _lib = abi.bind(ctypes.CDLL(LIB_PATH))
        version = _lib.pyr_abi_version()
        if version != abi.PYR_ABI_VERSION:
            raise ImportError("libpyrite_gpu.so has ABI version %d, python side expects %d" % (version, abi.PYR_ABI_VERSION))
    return _lib


def check(status):
    if status != abi.PYR_OK:
        raise PyriteGpuError(status, lib().pyr_last_error().decode("utf-8", "replace"))
