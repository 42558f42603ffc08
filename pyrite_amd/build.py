"""In-tree build of csrc/libpyrite_gpu.so (HIP kernels + C-ABI host code) for gfx950.

    python -m pyrite_amd.build [--force]

hipcc cross-compiles without a GPU. The .so stays in-tree (git-ignored, shipped to the GPU box by gpurun)."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
OUT = os.path.join(CSRC, "libpyrite_gpu.so")
SOURCES = ["kernels.hip", "api.cpp", "multi.cpp", "bvh.cpp"]
HEADERS = ["bvh.h", "device_scene.h", "api_internal.h", "exact_math.h", os.path.join("..", "..", "include", "pyrite_gpu.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared",
    "-munsafe-fp-atomics",  # atomicAdd(float*) -> one global_atomic_add_f32, no CAS loop
    "-fno-fast-math",
    "-ffp-contract=off",  # the reference is Rust (never fuses a*b+c); the kernels must round like the CPU oracle does
    "-Wall", "-Wno-unused-function",
    # VectorCombine's early run widens `{t.o.x, t.o.x}` (a splat of a scalar read through a reference, before the walker's
    # methods are inlined) into overlapping <2 x float> loads of the ray; SROA then cannot split the walker's ray into
    # registers and the ray origin / direction live in scratch for the whole render (24 B per lane stored at every new ray,
    # read back at every traversal entry and hit). Without the pass: no such alloca, C3 352 -> 363 Msamples/s at 32 spp.
    "-mllvm", "-disable-vector-combine",
    # The SLP vectorizer pairs scalar f32 multiplies / adds into v_pk_mul_f32 / v_pk_add_f32 wherever two happen to be
    # independent -- and pays for each pair with v_mov's that bring the operands into consecutive registers, plus the
    # register pressure of the tuples: in kernels bound by vector issue that is a loss everywhere. The packed arithmetic that
    # pays (box tests, triangle pairs) is written with ext_vector types and stays. Without the pass: no VGPR spills left in
    # the stage-scheduled kernel (388 -> 268 B of scratch = the traversal stack's deep end), intersect_kernel 84 -> 78 VGPRs
    # (5 -> 6 waves per SIMD); C3 396 -> 426, C5 350 -> 375, C2 745 -> 772 Msamples/s, World::intersect 8.07 -> 8.50 Grays/s.
    "-fno-slp-vectorize",
]


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return os.path.getmtime(os.path.abspath(__file__)) > t or any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


IMAGES_OUT = os.path.join(CSRC, "libpyrite_images.so")


def build_images(force=False, verbose=False):
    """Host-only texture ingest helper (baseline JPEG reader), plain C."""
    src = os.path.join(CSRC, "jpeg.c")
    if not force and os.path.exists(IMAGES_OUT) and os.path.getmtime(IMAGES_OUT) >= os.path.getmtime(src):
        return IMAGES_OUT
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-Wall", "-o", IMAGES_OUT, src, "-lm"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return IMAGES_OUT


HOST_DIR = os.path.join(CSRC, "host")
HOST_OUT = os.path.join(HOST_DIR, "libpyrite_host.so")
HOST_TOOL = os.path.join(HOST_DIR, "pyrite_host_tool")
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-I" + os.path.join(CSRC, "..", "..", "include")]


def build_host(force=False, verbose=False):
    """The C++ host layer above the C ABI (include/pyrite_host.hpp): libpyrite_host.so + pyrite_host_tool, plain g++.
    Both link libpyrite_gpu.so (rpath $ORIGIN/..), which must exist."""
    deps = [os.path.join(HOST_DIR, f) for f in ("pyrite_host.cpp", "lua_project.cpp", "images.cpp", "builtin_tables.inc")] + [os.path.join(CSRC, "jpeg.c")] + [os.path.join(CSRC, "..", "..", "include", h)
                                                                                           for h in ("pyrite_host.hpp", "pyrite_gpu.h")]
    cxx = os.environ.get("CXX", "g++")

    def newer(out, sources):
        return not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in sources)

    if force or newer(HOST_OUT, deps + [OUT]):
        jpeg = [os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-c", os.path.join(CSRC, "jpeg.c"), "-o", os.path.join(HOST_DIR, "jpeg.o")]
        if verbose:
            print(" ".join(jpeg))
        subprocess.check_call(jpeg, cwd=HOST_DIR)
        cmd = [cxx] + HOST_FLAGS + ["-shared", "pyrite_host.cpp", "lua_project.cpp", "images.cpp", "jpeg.o", "-L" + CSRC, "-lpyrite_gpu", "-Wl,-rpath,$ORIGIN/..", "-o", HOST_OUT]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=HOST_DIR)
    tool_src = os.path.join(HOST_DIR, "pyrite_host_tool.cpp")
    if force or newer(HOST_TOOL, deps + [tool_src, HOST_OUT]):
        cmd = [cxx] + HOST_FLAGS + ["pyrite_host_tool.cpp", "-L" + HOST_DIR, "-lpyrite_host", "-L" + CSRC, "-lpyrite_gpu", "-Wl,-rpath,$ORIGIN",
                                    "-Wl,-rpath,$ORIGIN/..", "-o", HOST_TOOL]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=HOST_DIR)
    return HOST_OUT


def compile_library(out, extra_flags=(), verbose=False):
    """The four sources -> objects in parallel -> one shared library. kernels.hip is compiled three times (-DPYR_TU=0: everything but
    the interpreter builds of the stage scheduler; -DPYR_TU=1: only those, the heaviest kernels; -DPYR_TU=2: their PRODUCT forms) so that
    the parts build side by side: 120 s -> ~50 s. -DPYR_PHASE_PROFILE builds keep one translation unit (their device-side counters are one variable)."""
    import tempfile

    flags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    split = not any("PYR_PHASE_PROFILE" in f or "PYR_DEV_ONLY" in f for f in extra_flags)
    units = [("kernels.hip", ["-DPYR_TU=0"]), ("kernels.hip", ["-DPYR_TU=1"]), ("kernels.hip", ["-DPYR_TU=2"])] if split else [("kernels.hip", [])]
    units += [(src, []) for src in SOURCES if src != "kernels.hip"]
    with tempfile.TemporaryDirectory(prefix="pyrite_build_") as tmp:
        jobs = []
        for k, (src, unit_flags) in enumerate(units):
            obj = os.path.join(tmp, "%d_%s.o" % (k, os.path.splitext(src)[0]))
            cmd = [HIPCC] + flags + unit_flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((subprocess.Popen(cmd, cwd=CSRC), cmd, obj))
        objects = []
        for proc, cmd, obj in jobs:
            if proc.wait() != 0:
                raise subprocess.CalledProcessError(proc.returncode, cmd)
            objects.append(obj)
        link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objects + ["-ldl"]
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link, cwd=CSRC)
    return out


def build(force=False, extra_flags=(), verbose=False):
    build_images(force, verbose)
    if force or stale():
        compile_library(OUT, extra_flags, verbose)
    build_host(force, verbose)
    return OUT


def build_variant(name, extra_flags, verbose=False):
    """A/B builds of the kernels (developer tool): csrc/variants/lib_<name>.so with extra -D flags, loaded through
    PYRITE_GPU_LIB. Never a CPU path: the same sources, other compile-time switches."""
    out_dir = os.path.join(CSRC, "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "lib_%s.so" % name)
    return compile_library(out, extra_flags, verbose)


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m pyrite_amd.build --variant NAME -DFLAG ...
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
        sys.exit(0)
    build(force="--force" in sys.argv, verbose=True)
    print(OUT)
