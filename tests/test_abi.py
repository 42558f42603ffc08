"""The C-ABI library loads and exports every symbol include/pyrite_gpu.h declares; the ctypes mirror matches the C
layout. No compute calls here (no GPU in this tier)."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from pyrite_amd import abi
from pyrite_amd import build as gpu_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pyrite_gpu.h")

STRUCTS = ["PyrGrain", "PyrFilmDesc", "PyrRenderParams", "PyrCamera", "PyrOperand", "PyrInstr", "PyrProgram", "PyrSpectrum", "PyrComponent",
           "PyrMaterial", "PyrLamp", "PyrTexture", "PyrSceneDesc", "PyrCounters", "PyrHit", "PyrBvhInfo", "PyrPathInfo", "PyrDevelopParams"]


@pytest.fixture(scope="module")
def lib():
    path = gpu_build.build()
    return C.CDLL(path)


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pyr_[a-z_]+)\s*\(", text)))


def test_header_and_ctypes_name_the_same_entry_points():
    assert declared_functions() == sorted(abi.ENTRY_POINTS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), "libpyrite_gpu.so does not export %s" % name
    abi.bind(lib)
    assert lib.pyr_abi_version() == abi.PYR_ABI_VERSION


def test_errors_are_reported_not_aborted(lib):
    abi.bind(lib)
    handle = C.c_void_p()
    rc = lib.pyr_scene_create(None, 0, C.byref(handle))
    assert rc == abi.PYR_ERR_INVALID_ARGUMENT
    assert b"null" in lib.pyr_last_error()
    assert lib.pyr_scene_bvh_info(None, None) == abi.PYR_ERR_INVALID_ARGUMENT
    assert lib.pyr_scene_path_info(None, None, None) == abi.PYR_ERR_INVALID_ARGUMENT
    lib.pyr_scene_destroy(None)  # must be a no-op


def test_ctypes_layout_matches_the_header():
    """sizeof / offsetof of every struct, as gcc lays the header out, against the ctypes mirror."""
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "%s"' % HEADER, "int main(void){"]
    for s in STRUCTS:
        cls = getattr(abi, s)
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (s, s))
        for field, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (s, field, s, field))
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "layout.c"), os.path.join(d, "layout")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-o", exe, src])
        out = subprocess.check_output([exe]).decode().split("\n")
    expect = dict(l.split() for l in out if l)
    for s in STRUCTS:
        cls = getattr(abi, s)
        assert int(expect[s]) == C.sizeof(cls), s
        for field, _ in cls._fields_:
            assert int(expect["%s.%s" % (s, field)]) == getattr(cls, field).offset, "%s.%s" % (s, field)
    assert C.sizeof(abi.PyrInstr) == 64 and C.sizeof(abi.PyrGrain) == 8


def test_product_never_imports_the_oracle():
    """Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may touch oracle/."""
    pkg = os.path.join(ROOT, "pyrite_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "oracle_render" not in text and "import oracle" not in text, f
