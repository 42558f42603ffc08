"""Lowering of the project tree (project.py) to the flat scene the C ABI consumes.

Mirrors, stage by stage, what the reference does between parsing and `Renderer::render`:

  * constant evaluation of expressions            pyrite/src/project/expressions.rs:75-258 (EvalContext)
  * ProgramCompiler::compile (expression -> ISA)  pyrite/src/program/compiler.rs:48-586, operand coercion :682-968
  * SurfaceMaterial::from_project (Mix/Add -> weighted component lists)  pyrite/src/materials/mod.rs:90-227
  * World::from_project (shapes, lamps, mesh ingest)  pyrite/src/world.rs:39-271, make_triangle :308-374
  * Camera::from_project / Renderer::from_project  pyrite/src/cameras.rs:30-55, pyrite/src/renderer/mod.rs:31-75
  * OBJ ingest as the `obj` crate presents it (objects -> groups -> polys of (v, vt, vn) index tuples)

The output is a `FlatScene` of numpy arrays; `FlatScene.desc()` wraps them in the ctypes `PyrSceneDesc`.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import struct

import numpy as np

from . import abi
from .project import Expr, Material, Node

f32 = np.float32
DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
_tables = None


def tables():
    """Built-in spectra the reference generates at build time (build.rs:18-59, :131-187), re-encoded as npz."""
    global _tables
    if _tables is None:
        with np.load(os.path.join(DATA_DIR, "tables.npz")) as z:
            _tables = {k: z[k] for k in z.files}
    return _tables


def _bits(x):
    return struct.unpack("<I", struct.pack("<f", float(f32(x))))[0]


class ProjectError(Exception):
    pass


# ------------------------------------------------------------------------------------------------
# constant evaluation (project/expressions.rs:75-258)
# ------------------------------------------------------------------------------------------------
def is_number(e):
    return isinstance(e, (int, float, np.floating, np.integer)) and not isinstance(e, bool)


def eval_number(e):
    """Evaluate<f32> (ExpressionValue for f32, expressions.rs:270-296)."""
    if is_number(e):
        return f32(e)
    t = e.type
    if t == "binary":
        l, r = eval_number(e.lhs), eval_number(e.rhs)
        with np.errstate(all="ignore"):
            return {"add": l + r, "sub": l - r, "mul": l * r, "div": l / r}[e.operator]
    if t == "mix":
        amount = min(max(eval_number(e.amount), f32(0)), f32(1))
        return eval_number(e.lhs) * (f32(1) - amount) + eval_number(e.rhs) * amount
    if t == "clamp":
        return max(min(eval_number(e.value), eval_number(e.max)), eval_number(e.min))
    if t == "vector":
        raise ProjectError("expected a number, but found a vector")
    if t == "rgb":
        raise ProjectError("expected a number, but found an RGB color")
    raise ProjectError("cannot evaluate %s as a constant" % t)


def eval_vector(e):
    """Evaluate<Vector> (ExpressionValue for Vector, expressions.rs:326-353) -> float32[4]."""
    if is_number(e):
        return np.full(4, f32(e), dtype=f32)
    t = e.type
    if t == "vector":
        return np.array([eval_number(e.x), eval_number(e.y), eval_number(e.z), eval_number(e.w)], dtype=f32)
    if t == "binary":
        l, r = eval_vector(e.lhs), eval_vector(e.rhs)
        with np.errstate(all="ignore"):
            return {"add": l + r, "sub": l - r, "mul": l * r, "div": l / r}[e.operator].astype(f32)
    if t == "mix":
        amount = min(max(eval_number(e.amount), f32(0)), f32(1))
        l, r = eval_vector(e.lhs), eval_vector(e.rhs)
        return (l + (r - l) * amount).astype(f32)
    if t == "rgb":
        raise ProjectError("expected a vector, but found an RGB color")
    raise ProjectError("cannot evaluate %s as a constant" % t)


def _normalize(v):
    """cgmath normalize: v * (1 / |v|) with |v| = sqrt((x*x + y*y) + z*z), all in f32."""
    v = np.asarray(v, dtype=f32)
    mag = np.sqrt(f32(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]))
    return (v * (f32(1) / mag)).astype(f32)


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f32)


def eval_transform(t):
    """Transform::LookAt (project/mod.rs:250-266): Matrix4::look_at(from, to, up).invert(), column-major float32[16].

    cgmath's look_at builds the view matrix from s = normalize(f x up), u = s x f, f = normalize(to - from); the
    inverse of that rigid transform is [s u -f | from], written out directly instead of a general 4x4 inverse."""
    if t is None:
        return None
    if t.type != "look_at":
        raise ProjectError("unknown transform %s" % t.type)
    frm = eval_vector(t.from_ if t.from_ is not None else 0.0)[:3]
    to = eval_vector(t.to if t.to is not None else 0.0)[:3]
    up = eval_vector(t.up)[:3] if t.up is not None else np.array([0, 1, 0], dtype=f32)
    f = _normalize(to - frm)
    s = _normalize(_cross(f, up))
    u = _cross(s, f)
    m = np.zeros(16, dtype=f32)
    m[0:3] = s
    m[4:7] = u
    m[8:11] = -f
    m[12:15] = frm
    m[15] = 1
    return m


def _transform_point(m, p):
    x = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12]
    y = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13]
    z = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14]
    w = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15]
    inv = f32(1) / f32(w)
    return np.array([x * inv, y * inv, z * inv], dtype=f32)


def _transform_vector(m, v):
    return np.array(
        [m[0] * v[0] + m[4] * v[1] + m[8] * v[2], m[1] * v[0] + m[5] * v[1] + m[9] * v[2], m[2] * v[0] + m[6] * v[1] + m[10] * v[2]],
        dtype=f32,
    )


# [3P] cgmath 0.17: Quaternion::from(Matrix3::from_cols(c0, c1, c2)) as (s, x, y, z), Quaternion * Vector3 -- the same
# formulas as oracle.cpp's quat_from_cols / quat_rotate, in f32.
def _quat_from_cols(c0, c1, c2):
    m00, m01, m02 = f32(c0[0]), f32(c0[1]), f32(c0[2])
    m10, m11, m12 = f32(c1[0]), f32(c1[1]), f32(c1[2])
    m20, m21, m22 = f32(c2[0]), f32(c2[1]), f32(c2[2])
    half, one = f32(0.5), f32(1.0)
    with np.errstate(all="ignore"):
        trace = m00 + m11 + m22
        if trace >= 0:
            s = np.sqrt(one + trace, dtype=f32)
            w = half * s
            s = half / s
            q = [w, (m12 - m21) * s, (m20 - m02) * s, (m01 - m10) * s]
        elif m00 > m11 and m00 > m22:
            s = np.sqrt((m00 - m11 - m22) + one, dtype=f32)
            x = half * s
            s = half / s
            q = [(m12 - m21) * s, x, (m10 + m01) * s, (m02 + m20) * s]
        elif m11 > m22:
            s = np.sqrt((m11 - m00 - m22) + one, dtype=f32)
            y = half * s
            s = half / s
            q = [(m20 - m02) * s, (m10 + m01) * s, y, (m21 + m12) * s]
        else:
            s = np.sqrt((m22 - m00 - m11) + one, dtype=f32)
            z = half * s
            s = half / s
            q = [(m01 - m10) * s, (m02 + m20) * s, (m21 + m12) * s, z]
    return np.array(q, dtype=f32)


def _quat_rotate(q, vec):
    v = q[1:4]
    with np.errstate(all="ignore"):
        tmp = _cross(v, vec) + vec * q[0]
        return (_cross(v, tmp) * f32(2.0) + vec).astype(f32)


def _ortho(v):  # math.rs:98-114
    eps = f32(1.0e-4)
    if abs(v[0]) < eps:
        unit = np.array([1, 0, 0], dtype=f32)
    elif abs(v[1]) < eps:
        unit = np.array([0, 1, 0], dtype=f32)
    elif abs(v[2]) < eps:
        unit = np.array([0, 0, 1], dtype=f32)
    else:
        unit = np.array([-v[1], v[0], 0], dtype=f32)
    return _cross(v, unit)


def _normal_transform(xform, vector, frame):  # Normal::transform, shapes/mod.rs:572-583
    with np.errstate(all="ignore"):
        n = _normalize(_transform_vector(xform, vector))
        x = _normalize(_transform_vector(xform, _quat_rotate(frame, np.array([1, 0, 0], dtype=f32))))
        y = _normalize(_transform_vector(xform, _quat_rotate(frame, np.array([0, 1, 0], dtype=f32))))
    return n, _quat_from_cols(x, y, n)


# ------------------------------------------------------------------------------------------------
# expression helpers used by material flattening (expressions.rs:20-63)
# ------------------------------------------------------------------------------------------------
def insert_sub(lhs, rhs):
    if is_number(lhs) and is_number(rhs):
        return float(lhs) - float(rhs)  # Expression::Number is f64
    return Expr("binary", operator="sub", lhs=lhs, rhs=rhs)


def insert_mul(lhs, rhs):
    if is_number(lhs) and is_number(rhs):
        return float(lhs) * float(rhs)
    return Expr("binary", operator="mul", lhs=lhs, rhs=rhs)


def insert_clamp(value, mn, mx):
    if is_number(value) and is_number(mn) and is_number(mx):
        return max(min(float(value), float(mx)), float(mn))
    return Expr("clamp", value=value, min=mn, max=mx)


# ------------------------------------------------------------------------------------------------
# the flat scene
# ------------------------------------------------------------------------------------------------
class _Pending(Exception):
    def __init__(self, child):
        self.child = child


class FlatScene:
    """Arrays of include/pyrite_gpu.h's PyrSceneDesc, plus the builders that fill them."""

    def __init__(self):
        self.tri_positions, self.tri_normals, self.tri_uvs, self.tri_material = [], [], [], []
        self.spheres, self.sphere_tex_scale, self.sphere_material = [], [], []
        self.planes, self.plane_material = [], []
        self.lamps, self.materials, self.components, self.programs, self.instrs = [], [], [], [], []
        self.spectra, self.spectrum_data = [], []
        self.tri_frames, self.plane_frames = [], []
        self.textures, self._texture_ids = [], {}  # (format, texels) in id order; (path | id(array), linear, mono) -> id
        self.uses_normal_maps = False
        self._spectrum_ids = {}
        self.uses_rgb_basis = False
        self.sky_program = 0
        self._keep = None

    # ---- spectra (SpectrumId::from_lua, project/spectra.rs:116-145: one id per Lua table) ----
    def spectrum_id(self, e):
        key = id(e)
        if key in self._spectrum_ids:
            return self._spectrum_ids[key]
        name = e.get("name")
        if name is not None:
            t = tables()
            if name not in ("a", "d65"):
                raise ProjectError("unknown builtin spectrum: %s" % name)
            fmt, mn, mx, data = abi.SPECTRUM_ARRAY, float(t["light_min"]), float(t["light_max"]), t[name].astype(f32)
            count = len(data)
        elif e.get("format") == "array":
            fmt, mn, mx = abi.SPECTRUM_ARRAY, float(e.min), float(e.max)
            data = np.asarray(e.points, dtype=f32)
            count = len(data)
        elif e.get("format") == "curve":
            fmt, mn, mx = abi.SPECTRUM_CURVE, 0.0, 0.0
            data = np.asarray(e.points, dtype=f32).reshape(-1, 2)
            count = len(data)
            data = data.reshape(-1)
        else:
            raise ProjectError("unknown spectrum format %r" % e.get("format"))
        sid = len(self.spectra)
        self.spectra.append((fmt, mn, mx, len(self.spectrum_data), count))
        self.spectrum_data.extend(float(x) for x in data)
        self._spectrum_ids[key] = sid
        self._keep_alive = getattr(self, "_keep_alive", [])
        self._keep_alive.append(e)  # ids stay unique while the expression object lives
        return sid

    # ---- textures (TextureLoader::load_color / load_mono, project/textures.rs:56-118: one id per file and kind) ----
    def texture_id(self, e, base_dir=None):
        source, linear, mono = e.path, bool(e.get("linear")), e.type == "mono_texture"
        if isinstance(source, (str, os.PathLike)):
            path = os.fspath(source)
            if not os.path.isabs(path):
                path = os.path.join(base_dir or getattr(self, "base_dir", "."), path)
            key = (os.path.normpath(path), linear, mono)
        else:
            key = (id(source), linear, mono)
        if key in self._texture_ids:
            return self._texture_ids[key]
        from . import images

        try:
            texels = images.load_texture(key[0], linear, mono) if isinstance(key[0], str) else images.linearise(source, linear, mono)
        except (OSError, ValueError) as error:  # textures.rs:78-83 / :102-107
            raise ProjectError("could not load %s as %s texture: %s" % (key[0], "mono" if mono else "color", error))
        tid = len(self.textures)
        self.textures.append((abi.TEXTURE_MONO if mono else abi.TEXTURE_COLOR, texels))
        self._texture_ids[key] = tid
        self._keep_alive = getattr(self, "_keep_alive", [])
        self._keep_alive.append(source)
        return tid

    # ---- ProgramCompiler::compile (program/compiler.rs:48-586) ----
    def compile(self, expression, allow_wavelength=True, output="number"):
        """Returns the index of the compiled program. `output` is "number" (f32 programs: colours, probabilities)
        or "vector" (normal maps); `allow_wavelength=False` mirrors NormalInput (tracer.rs:58-68)."""
        if is_number(expression):  # compiler.rs:62-69
            self.programs.append(dict(kind=abi.PROGRAM_CONSTANT, constant=float(f32(expression)), first=0, n=0,
                                      output_kind=abi.OUTPUT_NUMBER if output == "number" else abi.OUTPUT_VECTOR,
                                      output_reg=0, numbers=0, vectors=0, rgbs=0))
            return len(self.programs) - 1

        status = {id(expression): ("pending", expression)}
        pending = [expression]
        instructions = []
        counts = {"n": 0, "v": 0, "c": 0}

        def next_reg(kind):
            r = counts[kind]
            counts[kind] += 1
            return r

        def emit(**kw):
            base = dict(op=0, value_type=0, operator=0, deps=0, output=0, a=0, b=0, x=None, y=None, z=None, w=None)
            base.update(kw)
            instructions.append(base)

        def number_input():  # get_number_input, compiler.rs:970-975
            if not allow_wavelength:
                raise ProjectError("the wavelength is not available during normal mapping")
            return (abi.OPERAND_INPUT, abi.INPUT_WAVELENGTH), abi.DEP_WAVELENGTH

        VEC_INPUT = {"normal": (abi.INPUT_NORMAL, abi.DEP_NORMAL), "incident": (abi.INPUT_INCIDENT, abi.DEP_INCIDENT),
                     "texture": (abi.INPUT_TEXTURE, abi.DEP_TEXTURE)}

        def try_get_register(e):  # compiler.rs:609-634
            if is_number(e):
                return ("number", f32(e))
            st = status.setdefault(id(e), ("pending", e))
            if st[0] == "done":
                return ("register", st[1], st[2])
            raise _Pending(e)

        def try_get_number_value(e):  # compiler.rs:636-680
            got = try_get_register(e)
            if got[0] == "number":
                return (abi.OPERAND_CONSTANT, _bits(got[1])), 0
            (kind, reg), deps = got[1], got[2]
            if kind == "n":
                return (abi.OPERAND_REGISTER, reg), deps
            if kind == "v":
                raise ProjectError("cannot use a vector as a number")
            wl, wl_deps = number_input()
            out = next_reg("n")
            self.uses_rgb_basis = True
            emit(op=abi.OP_RGB_SPECTRUM, x=wl, a=reg, output=out, deps=deps | wl_deps)
            return (abi.OPERAND_REGISTER, out), deps | wl_deps

        def const_operand(v):
            return (abi.OPERAND_CONSTANT, _bits(v))

        def number_constant_to(kind, number):  # compiler.rs:991-1008, :1047-1064
            out = next_reg(kind)
            c = const_operand(number)
            if kind == "v":
                emit(op=abi.OP_VECTOR, x=c, y=c, z=c, w=c, output=out, deps=0)
            else:
                emit(op=abi.OP_RGB, x=c, y=c, z=c, output=out, deps=0)
            return out

        def number_register_to(kind, reg, deps):  # compiler.rs:1010-1028, :1066-1083
            out = next_reg(kind)
            r = (abi.OPERAND_REGISTER, reg)
            if kind == "v":
                emit(op=abi.OP_VECTOR, x=r, y=r, z=r, w=r, output=out, deps=deps)
            else:
                emit(op=abi.OP_RGB, x=r, y=r, z=r, output=out, deps=deps)
            return out

        def rgb_register_to_vector(reg, deps):  # compiler.rs:1030-1045
            out = next_reg("v")
            emit(op=abi.OP_RGB_TO_VECTOR, a=reg, output=out, deps=deps)
            return out

        VT = {"n": abi.VT_NUMBER, "v": abi.VT_VECTOR, "c": abi.VT_RGB}

        def convert_operands(lhs, rhs):  # compiler.rs:682-968
            def as_reg(x):
                return (x[1][0], x[1][1], x[2])  # kind, reg, deps

            if lhs[0] == "number" and rhs[0] == "number":
                lo, ro = next_reg("n"), next_reg("n")
                emit(op=abi.OP_NUMBER, x=const_operand(lhs[1]), output=lo, deps=0)
                emit(op=abi.OP_NUMBER, x=const_operand(rhs[1]), output=ro, deps=0)
                return "n", (lo, 0), (ro, 0)
            if lhs[0] == "number":
                rk, rr, rd = as_reg(rhs)
                if rk == "n":
                    lo = next_reg("n")
                    emit(op=abi.OP_NUMBER, x=const_operand(lhs[1]), output=lo, deps=0)
                    return "n", (lo, 0), (rr, rd)
                return rk, (number_constant_to(rk, lhs[1]), 0), (rr, rd)
            if rhs[0] == "number":
                lk, lr, ld = as_reg(lhs)
                if lk == "n":
                    ro = next_reg("n")
                    emit(op=abi.OP_NUMBER, x=const_operand(rhs[1]), output=ro, deps=0)
                    return "n", (lr, ld), (ro, 0)
                return lk, (lr, ld), (number_constant_to(lk, rhs[1]), 0)
            lk, lr, ld = as_reg(lhs)
            rk, rr, rd = as_reg(rhs)
            if lk == rk:
                return lk, (lr, ld), (rr, rd)
            if lk == "n":  # number with vector / rgb: widen the number
                return rk, (number_register_to(rk, lr, ld), ld), (rr, rd)
            if rk == "n":
                return lk, (lr, ld), (number_register_to(lk, rr, rd), rd)
            if lk == "v":  # vector with rgb: rgb -> vector
                return "v", (lr, ld), (rgb_register_to_vector(rr, rd), rd)
            return "v", (rgb_register_to_vector(lr, ld), ld), (rr, rd)

        def done(e, kind, reg, deps):
            status[id(e)] = ("done", (kind, reg), deps)

        while pending:
            e = pending.pop()
            if status[id(e)][0] == "done":
                continue
            try:
                t = e.type
                if t == "vector":
                    x, xd = try_get_number_value(e.x)
                    y, yd = try_get_number_value(e.y)
                    z, zd = try_get_number_value(e.z)
                    w, wd = try_get_number_value(e.w)
                    out, deps = next_reg("v"), xd | yd | zd | wd
                    emit(op=abi.OP_VECTOR, x=x, y=y, z=z, w=w, output=out, deps=deps)
                    done(e, "v", out, deps)
                elif t == "rgb":
                    r, rd = try_get_number_value(e.red)
                    g, gd = try_get_number_value(e.green)
                    b, bd = try_get_number_value(e.blue)
                    out, deps = next_reg("c"), rd | gd | bd
                    emit(op=abi.OP_RGB, x=r, y=g, z=b, output=out, deps=deps)
                    done(e, "c", out, deps)
                elif t == "fresnel":
                    (ni, nd), (ii, idp) = VEC_INPUT["normal"], VEC_INPUT["incident"]
                    ior, iord = try_get_number_value(e.ior)
                    env, envd = try_get_number_value(e.env_ior)
                    out, deps = next_reg("n"), nd | idp | iord | envd
                    emit(op=abi.OP_FRESNEL, x=ior, y=env, a=ni, b=ii, output=out, deps=deps)
                    done(e, "n", out, deps)
                elif t == "blackbody":
                    wl, wld = number_input()
                    temp, td = try_get_number_value(e.temperature)
                    out, deps = next_reg("n"), wld | td
                    emit(op=abi.OP_BLACKBODY, x=wl, y=temp, output=out, deps=deps)
                    done(e, "n", out, deps)
                elif t == "spectrum":
                    wl, deps = number_input()
                    out = next_reg("n")
                    emit(op=abi.OP_SPECTRUM, x=wl, a=self.spectrum_id(e), output=out, deps=deps)
                    done(e, "n", out, deps)
                elif t in ("color_texture", "mono_texture"):  # compiler.rs:282-323
                    ti, td = VEC_INPUT["texture"]
                    kind = "c" if t == "color_texture" else "n"
                    out = next_reg(kind)
                    emit(op=abi.OP_COLOR_TEXTURE if kind == "c" else abi.OP_MONO_TEXTURE, a=self.texture_id(e), b=ti, output=out, deps=td)
                    done(e, kind, out, td)
                elif t == "mix":
                    amount, ad = try_get_number_value(e.amount)
                    lhs = try_get_register(e.lhs)
                    rhs = try_get_register(e.rhs)
                    kind, (l, ld), (r, rd) = convert_operands(lhs, rhs)
                    deps = ad | ld | rd
                    out = next_reg(kind)
                    done(e, kind, out, deps)
                    emit(op=abi.OP_MIX, value_type=VT[kind], a=l, b=r, x=amount, output=out, deps=deps)
                elif t == "binary":
                    lhs = try_get_register(e.lhs)
                    rhs = try_get_register(e.rhs)
                    kind, (l, ld), (r, rd) = convert_operands(lhs, rhs)
                    deps = ld | rd
                    out = next_reg(kind)
                    done(e, kind, out, deps)
                    op = {"add": abi.BIN_ADD, "sub": abi.BIN_SUB, "mul": abi.BIN_MUL, "div": abi.BIN_DIV}[e.operator]
                    emit(op=abi.OP_BINARY, value_type=VT[kind], operator=op, a=l, b=r, output=out, deps=deps)
                elif t == "clamp":
                    v, vd = try_get_number_value(e.value)
                    mn, mnd = try_get_number_value(e.min)
                    mx, mxd = try_get_number_value(e.max)
                    out, deps = next_reg("n"), vd | mnd | mxd
                    emit(op=abi.OP_CLAMP, x=v, y=mn, z=mx, output=out, deps=deps)
                    done(e, "n", out, deps)
                else:
                    raise ProjectError("not an expression: %r" % (e,))
            except _Pending as p:  # unwrap_or_push!, compiler.rs:25-36
                pending.append(e)
                pending.append(p.child)

        st = status[id(expression)]
        if st[0] != "done":
            raise ProjectError("the expression was not compiled to completion")
        (kind, reg), deps = st[1], st[2]
        if output == "number":  # compiler.rs:528-563
            if kind == "v":
                raise ProjectError("cannot use a vector as a number")
            if kind == "c":
                wl, wld = number_input()
                out = next_reg("n")
                self.uses_rgb_basis = True
                emit(op=abi.OP_RGB_SPECTRUM, x=wl, a=reg, output=out, deps=deps | wld)
                reg = out
            output_kind = abi.OUTPUT_NUMBER
        else:
            if kind == "n":
                reg = number_register_to("v", reg, deps)
            elif kind == "c":
                reg = rgb_register_to_vector(reg, deps)
            output_kind = abi.OUTPUT_VECTOR
        if counts["n"] > abi.MAX_NUMBER_REGISTERS or counts["v"] > abi.MAX_VECTOR_REGISTERS or counts["c"] > abi.MAX_RGB_REGISTERS:
            raise ProjectError("program needs more registers than the GPU VM provides")
        first = len(self.instrs)
        self.instrs.extend(instructions)
        self.programs.append(dict(kind=abi.PROGRAM_INSTRUCTIONS, constant=0.0, first=first, n=len(instructions), output_kind=output_kind,
                                  output_reg=reg, numbers=counts["n"], vectors=counts["v"], rgbs=counts["c"]))
        return len(self.programs) - 1

    # ---- SurfaceMaterial::from_project (materials/mod.rs:90-227) + Material::from_project (:33-46) ----
    def add_material(self, mat):
        surface = mat["surface"] if isinstance(mat, dict) else mat.surface
        normal_map = mat.get("normal_map") if isinstance(mat, dict) else mat.get("normal_map")
        normal_map_program = -1
        if normal_map is not None:  # materials/mod.rs:41-44: a Vector program over NormalInput (no wavelength)
            normal_map_program = self.compile(normal_map, allow_wavelength=False, output="vector")
            self.uses_normal_maps = True
        stack = [(surface, None)]
        components, emissive = [], []
        while stack:
            node, probability = stack.pop()
            if not isinstance(node, Material):
                raise ProjectError("missing material")
            t = node.type
            if t in ("emissive", "diffuse", "mirror", "refractive"):  # leaves compile their probability expression
                prob_program = -1 if probability is None else self.compile(probability)
            if t in ("emissive", "diffuse", "mirror"):
                comp = dict(bsdf={"emissive": abi.BSDF_EMISSIVE, "diffuse": abi.BSDF_DIFFUSE, "mirror": abi.BSDF_MIRROR}[t],
                            probability=prob_program, color=self.compile(node.color), ior=0.0, env_ior=0.0, dispersion=0.0, env_dispersion=0.0)
                components.append(comp)
                if t == "emissive":
                    emissive.append(dict(comp))
            elif t == "refractive":
                comp = dict(bsdf=abi.BSDF_REFRACTIVE, probability=prob_program, color=self.compile(node.color),
                            ior=float(eval_number(node.ior)),
                            env_ior=float(eval_number(node.env_ior)) if node.env_ior is not None else 1.0,
                            dispersion=float(eval_number(node.dispersion)) if node.dispersion is not None else 0.0,
                            env_dispersion=float(eval_number(node.env_dispersion)) if node.env_dispersion is not None else 0.0)
                components.append(comp)
            elif t == "mix":
                amount = insert_clamp(node.amount, 0.0, 1.0)
                lhs_probability = insert_mul(probability, amount) if probability is not None else amount
                stack.append((node.lhs, lhs_probability))
                stack.append((node.rhs, insert_sub(1.0, lhs_probability)))
            elif t == "binary":
                stack.append((node.lhs, probability))
                stack.append((node.rhs, probability))
            else:
                raise ProjectError("unknown material type %s" % t)
        for c in components:
            c["compensation"] = float(len(components))
        for c in emissive:
            c["compensation"] = float(len(emissive))
        first_component = len(self.components)
        self.components.extend(components)
        first_emissive = len(self.components)
        self.components.extend(emissive)
        self.materials.append((first_component, len(components), first_emissive, len(emissive), normal_map_program))
        return len(self.materials) - 1, len(emissive) > 0

    # ---- World::from_project (world.rs:39-271) ----
    def add_world(self, world, base_dir="."):
        self.base_dir = base_dir  # texture paths are relative to the project directory (project/textures.rs:58-67)
        sky = world.get("sky") if isinstance(world, dict) else None
        self.sky_program = self.compile(sky if sky is not None else 0.0)
        objects = world["objects"] if isinstance(world, dict) else world.objects
        for i, obj in enumerate(objects):
            t = obj.type
            if t == "sphere":
                m, emissive = self.add_material(obj.material)
                position = eval_vector(obj.position)[:3]
                radius = eval_number(obj.radius)
                scale = eval_vector(obj.texture_scale)[:2] if obj.texture_scale is not None else np.array([1, 1], dtype=f32)
                self.spheres.append([position[0], position[1], position[2], radius])
                self.sphere_tex_scale.append(scale)
                self.sphere_material.append(m)
                if emissive:
                    self.lamps.append(dict(kind=abi.LAMP_SHAPE, shape_kind=abi.SHAPE_SPHERE, shape_index=len(self.spheres) - 1))
            elif t == "plane":
                m, emissive = self.add_material(obj.material)
                normal = _normalize(eval_vector(obj.normal)[:3])
                origin = eval_vector(obj.origin)[:3]
                scale = eval_vector(obj.texture_scale)[:2] if obj.texture_scale is not None else np.array([1, 1], dtype=f32)
                self.planes.append([*origin, *normal, *scale])
                z = _normalize(_ortho(normal))  # math::utils::basis, math.rs:119-123
                y = _normalize(_cross(z, normal))
                self.plane_frames.append(_quat_from_cols(y, z, normal))  # world.rs:95-99
                self.plane_material.append(m)
            elif t == "mesh":
                self._add_mesh(i, obj, base_dir)
            elif t == "directional_light":
                self.lamps.append(dict(kind=abi.LAMP_DIRECTIONAL, v=eval_vector(obj.direction)[:3], width=float(eval_number(obj.width)),
                                       color=self.compile(obj.color)))
            elif t == "point_light":
                self.lamps.append(dict(kind=abi.LAMP_POINT, v=eval_vector(obj.position)[:3], color=self.compile(obj.color)))
            elif t == "ray_marched":
                raise ProjectError("ray-marched shapes are out of scope for the GPU path (SURVEY.md section 8)")
            else:
                raise ProjectError("objects[%d]: unknown object type %s" % (i, t))
        return self

    def _add_mesh(self, i, obj, base_dir):  # world.rs:184-236
        if isinstance(obj.file, dict):  # an already loaded mesh (generated geometry)
            mesh = obj.file
        else:
            mesh = load_obj(obj.file if os.path.isabs(obj.file) else os.path.join(base_dir, obj.file))
        materials = dict(obj.materials)
        for o in mesh["objects"]:
            if o["name"] not in materials:
                raise ProjectError("objects[%d]: missing material for '%s'" % (i, o["name"]))
            m, emissive = self.add_material(materials.pop(o["name"]))
            xform = eval_transform(obj.transform)
            if xform is None:
                xform = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1], dtype=f32)
            scale = eval_number(obj.scale) if obj.scale is not None else f32(1.0)
            for poly in o["polys"]:
                if len(poly) != 3:
                    continue  # only `[x, y, z]` polys are taken, world.rs:218-232
                self._add_triangle(mesh, poly, m, scale, xform)
                if emissive:
                    self.lamps.append(dict(kind=abi.LAMP_SHAPE, shape_kind=abi.SHAPE_TRIANGLE, shape_index=len(self.tri_material) - 1))

    def _add_triangle(self, mesh, poly, material, scale, xform):  # make_triangle world.rs:308-374 + scale/transform
        P, T, N = mesh["position"], mesh["texture"], mesh["normal"]
        v = [P[ix[0]].astype(f32) for ix in poly]
        if all(ix[2] is not None for ix in poly):
            n = [N[ix[2]].astype(f32) for ix in poly]
        else:
            flat = _normalize(_cross(v[1] - v[0], v[2] - v[0]))
            n = [flat, flat, flat]
        uv = [T[ix[1]].astype(f32) if ix[1] is not None else np.zeros(2, dtype=f32) for ix in poly]
        # tangent space from the uv deltas, world.rs:337-346 (before scale / transform)
        with np.errstate(all="ignore"):
            dp1, dp2 = v[1] - v[0], v[2] - v[0]
            dt1, dt2 = uv[1] - uv[0], uv[2] - uv[0]
            r = f32(1.0) / (dt1[0] * dt2[1] - dt1[1] * dt2[0])
            tangent = (dp1 * dt2[1] - dp2 * dt1[1]) * r
            bitangent = (dp2 * dt1[0] - dp1 * dt2[0]) * r
            frames = [_quat_from_cols(tangent, bitangent, x) for x in n]
        v = [(p * scale).astype(f32) for p in v]  # Shape::scale, shapes/mod.rs:290-320
        transformed = [_normal_transform(xform, x, q) for x, q in zip(n, frames)]  # Normal::transform, shapes/mod.rs:572-583
        n = [t[0] for t in transformed]
        frames = [t[1] for t in transformed]
        v = [_transform_point(xform, p) for p in v]  # Shape::transform, shapes/mod.rs:322-344
        self.add_triangle(v, n, uv, material, frames)

    def add_triangle(self, positions, normals, uvs, material, frames=None):
        self.tri_positions.append(np.asarray(positions, dtype=f32).reshape(9))
        self.tri_normals.append(np.asarray(normals, dtype=f32).reshape(9))
        self.tri_uvs.append(np.asarray(uvs, dtype=f32).reshape(6))
        self.tri_frames.append(np.asarray(frames, dtype=f32).reshape(12) if frames is not None else np.tile(np.array([1, 0, 0, 0], dtype=f32), 3))
        self.tri_material.append(material)

    def add_triangles(self, positions, normals, material, emissive=False):
        """Bulk path for generated meshes: positions/normals float32 [n,3,3]."""
        positions = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 9)
        normals = np.ascontiguousarray(normals, dtype=f32).reshape(-1, 9)
        base = len(self.tri_material)
        self.tri_positions.append(positions)
        self.tri_normals.append(normals)
        self.tri_uvs.append(np.zeros((len(positions), 6), dtype=f32))
        self.tri_frames.append(np.tile(np.array([1, 0, 0, 0], dtype=f32), (len(positions), 3)))
        self.tri_material.extend([material] * len(positions))
        if emissive:
            for k in range(len(positions)):
                self.lamps.append(dict(kind=abi.LAMP_SHAPE, shape_kind=abi.SHAPE_TRIANGLE, shape_index=base + k))

    # ---- ctypes view ----
    def desc(self):
        keep = {}

        def farr(name, rows, width):
            if len(rows) == 0:
                a = np.zeros((0, width), dtype=f32)
            else:
                a = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=f32).reshape(-1, width) for r in rows], axis=0))
            keep[name] = a
            return a.ctypes.data_as(C.POINTER(C.c_float)), len(a)

        def uarr(name, values):
            a = np.ascontiguousarray(np.asarray(values, dtype=np.uint32))
            keep[name] = a
            return a.ctypes.data_as(C.POINTER(C.c_uint32))

        d = abi.PyrSceneDesc()
        d.tri_positions, n_tri = farr("tp", self.tri_positions, 9)
        d.tri_normals, _ = farr("tn", self.tri_normals, 9)
        d.tri_uvs, _ = farr("tu", self.tri_uvs, 6)
        d.tri_material = uarr("tm", self.tri_material)
        d.num_triangles = n_tri
        assert n_tri == len(self.tri_material)
        d.spheres, n_sph = farr("sp", self.spheres, 4)
        d.sphere_tex_scale, _ = farr("st", self.sphere_tex_scale, 2)
        d.sphere_material = uarr("sm", self.sphere_material)
        d.num_spheres = n_sph
        d.planes, n_pl = farr("pl", self.planes, 8)
        d.plane_material = uarr("pm", self.plane_material)
        d.num_planes = n_pl

        lamps = (abi.PyrLamp * max(1, len(self.lamps)))()
        for k, l in enumerate(self.lamps):
            lamps[k].kind = l["kind"]
            lamps[k].shape_kind = l.get("shape_kind", 0)
            lamps[k].shape_index = l.get("shape_index", 0)
            lamps[k].color_program = l.get("color", 0)
            v = l.get("v", (0, 0, 0))
            for j in range(3):
                lamps[k].v[j] = float(v[j])
            lamps[k].width = l.get("width", 0.0)
        d.lamps, d.num_lamps = lamps, len(self.lamps)

        mats = (abi.PyrMaterial * max(1, len(self.materials)))()
        for k, m in enumerate(self.materials):
            mats[k] = abi.PyrMaterial(*m)
        d.materials, d.num_materials = mats, len(self.materials)

        comps = (abi.PyrComponent * max(1, len(self.components)))()
        for k, c in enumerate(self.components):
            comps[k] = abi.PyrComponent(c["bsdf"], c["color"], c["probability"], c["compensation"], c["ior"], c["env_ior"], c["dispersion"],
                                        c["env_dispersion"])
        d.components, d.num_components = comps, len(self.components)

        progs = (abi.PyrProgram * max(1, len(self.programs)))()
        for k, p in enumerate(self.programs):
            progs[k] = abi.PyrProgram(p["kind"], p["constant"], p["first"], p["n"], p["output_kind"], p["output_reg"], p["numbers"], p["vectors"],
                                      p["rgbs"])
        d.programs, d.num_programs = progs, len(self.programs)

        instrs = (abi.PyrInstr * max(1, len(self.instrs)))()
        for k, ins in enumerate(self.instrs):
            r = instrs[k]
            r.op, r.value_type, r.operator_, r.deps = ins["op"], ins["value_type"], ins["operator"], ins["deps"]
            r.output, r.a, r.b = ins["output"], ins["a"], ins["b"]
            for name in "xyzw":
                operand = ins[name] or (abi.OPERAND_CONSTANT, 0)
                setattr(r, name, abi.PyrOperand(operand[0], operand[1]))
        d.instrs, d.num_instrs = instrs, len(self.instrs)

        spectra = (abi.PyrSpectrum * max(1, len(self.spectra)))()
        for k, s in enumerate(self.spectra):
            spectra[k] = abi.PyrSpectrum(*s)
        d.spectra, d.num_spectra = spectra, len(self.spectra)
        sd = np.ascontiguousarray(np.asarray(self.spectrum_data, dtype=f32))
        keep["sd"] = sd
        d.spectrum_data, d.num_spectrum_floats = sd.ctypes.data_as(C.POINTER(C.c_float)), len(sd)

        if self.uses_rgb_basis:
            t = tables()
            basis = np.ascontiguousarray(t["rgb_basis"], dtype=f32)
            keep["rgb"] = basis
            d.rgb_basis = basis.ctypes.data_as(C.POINTER(C.c_float))
            d.rgb_basis_count, d.rgb_basis_min, d.rgb_basis_max = len(basis), float(t["rgb_min"]), float(t["rgb_max"])
        d.sky_program = self.sky_program
        if self.uses_normal_maps and n_tri:
            d.tri_frames, _ = farr("tf", self.tri_frames, 12)
        if n_pl and len(self.plane_frames) == n_pl:
            d.plane_frames, _ = farr("pf", self.plane_frames, 4)
        if self.textures:
            records = (abi.PyrTexture * len(self.textures))()
            offset, blobs = 0, []
            for k, (fmt, texels) in enumerate(self.textures):
                records[k] = abi.PyrTexture(fmt, texels.shape[1], texels.shape[0], 0, offset)
                blobs.append(np.ascontiguousarray(texels, dtype=f32).reshape(-1))
                offset += blobs[-1].size
            data = np.ascontiguousarray(np.concatenate(blobs))
            keep["tex"], keep["texrec"] = data, records
            d.textures, d.num_textures = records, len(self.textures)
            d.texture_data, d.num_texture_floats = data.ctypes.data_as(C.POINTER(C.c_float)), data.size
        keep.update(lamps=lamps, mats=mats, comps=comps, progs=progs, instrs=instrs, spectra=spectra)
        self._keep = keep  # the descriptor borrows these buffers
        return d


# ------------------------------------------------------------------------------------------------
# OBJ ingest (`obj` crate 0.10.2 data model: objects -> groups -> polys of IndexTuple(v, vt?, vn?))
# ------------------------------------------------------------------------------------------------
def load_obj(path):
    position, texture, normal = [], [], []
    objects = []
    current = None

    def ensure_object():
        nonlocal current
        if current is None:
            current = {"name": "default", "polys": []}
            objects.append(current)
        return current

    def index(token, count):
        if token == "":
            return None
        k = int(token)
        return k - 1 if k > 0 else count + k

    with open(path) as f:
        for line in f:
            parts = line.split()
            if not parts or parts[0].startswith("#"):
                continue
            tag = parts[0]
            if tag == "v":
                position.append([float(x) for x in parts[1:4]])
            elif tag == "vt":
                uv = [float(x) for x in parts[1:3]]
                texture.append(uv + [0.0] * (2 - len(uv)))
            elif tag == "vn":
                normal.append([float(x) for x in parts[1:4]])
            elif tag == "o":
                current = {"name": parts[1] if len(parts) > 1 else "", "polys": []}
                objects.append(current)
            elif tag == "f":
                poly = []
                for tok in parts[1:]:
                    fields = (tok.split("/") + ["", ""])[:3]
                    poly.append((index(fields[0], len(position)), index(fields[1], len(texture)), index(fields[2], len(normal))))
                ensure_object()["polys"].append(poly)
    return {
        "position": np.asarray(position, dtype=f32).reshape(-1, 3),
        "texture": np.asarray(texture, dtype=f32).reshape(-1, 2),
        "normal": np.asarray(normal, dtype=f32).reshape(-1, 3),
        "objects": objects,
    }


# ------------------------------------------------------------------------------------------------
# camera / renderer / film parameters
# ------------------------------------------------------------------------------------------------
def camera_from_project(cam):
    """Camera::from_project, cameras.rs:30-55."""
    if cam.type != "perspective":
        raise ProjectError("unknown camera %s" % cam.type)
    fov = eval_number(cam.fov)
    half = f32(fov * f32(0.5)) * f32(math.pi / 180.0)  # cgmath Deg -> Rad
    view_plane = f32(np.cos(half, dtype=f32) / np.sin(half, dtype=f32))
    out = abi.PyrCamera()
    m = eval_transform(cam.transform)
    for k in range(16):
        out.cam_to_world[k] = float(m[k])
    out.view_plane = float(view_plane)
    out.focus_distance = float(eval_number(cam.focus_distance)) if cam.focus_distance is not None else 1.0
    out.aperture = float(eval_number(cam.aperture)) if cam.aperture is not None else 0.0
    return out


DEFAULT_SPECTRUM_SPAN = (380.0, 780.0)  # renderer/mod.rs:16


def renderer_from_project(r):
    """Renderer::from_project / from_shared, renderer/mod.rs:31-75 (defaults :63-75). Only `simple` is in scope."""
    if r.type != "simple":
        raise ProjectError("renderer.%s is out of scope: only the camera-to-light `simple` renderer is built" % r.type)
    return dict(
        bounces=r.bounces if r.bounces is not None else 8,
        pixel_samples=int(r.pixel_samples),
        light_samples=r.light_samples if r.light_samples is not None else 4,
        spectrum_samples=r.spectrum_samples if r.spectrum_samples is not None else 10,
        spectrum_bins=r.spectrum_resolution if r.spectrum_resolution is not None else 64,
        spectrum_span=DEFAULT_SPECTRUM_SPAN,
        tile_size=r.tile_size if r.tile_size is not None else 32,
    )
