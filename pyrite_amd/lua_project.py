"""Loader for Pyrite project files (`*.lua`).

The reference runs project files through `mlua` with the prelude pyrite/src/project/lib.lua, which turns the script into
plain tagged tables (project/mod.rs:55-100). There is no Lua interpreter in this image, so this module evaluates the
declarative subset of Lua that project files are written in -- `local` and global assignments, `return`, table
constructors, function and method calls with parenthesised / table / string arguments, field access, arithmetic, string
concatenation, comparison and logic operators, `require` -- against an environment that offers the prelude's names
(`vector`, `rgb`, `spectrum`, `texture`, `blackbody`, `fresnel`, `mix`, `light_source`, `material.*`, `shape.*`, `light.*`,
`transform.look_at`, `camera.perspective`, `renderer.*`, `ray_marched.*`, `bounds.*`, `:with{}` / `:clone()`) through
pyrite_amd.project. Control flow and function definitions -- which no project file in pyrite/test uses -- are rejected with a
clear error instead of being half-supported.

    project, base_dir = load_project("pyrite/test/spheres/spheres.lua")
    world, camera, renderer, film = scenes.build(project, seed=1, base_dir=base_dir)
"""
from __future__ import annotations

import inspect
import os
import re

from . import project as P


class LuaError(Exception):
    pass


# ------------------------------------------------------------------------------------------------ tokens
_TOKEN = re.compile(r"""
    (?P<ws>\s+)
  | (?P<longcomment>--\[(?P<lc_eq>=*)\[.*?\](?P=lc_eq)\])
  | (?P<comment>--[^\n]*)
  | (?P<number>0[xX][0-9a-fA-F]+|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)
  | (?P<name>[A-Za-z_][A-Za-z_0-9]*)
  | (?P<string>"(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*')
  | (?P<longstring>\[(?P<ls_eq>=*)\[.*?\](?P=ls_eq)\])
  | (?P<op>\.\.\.|\.\.|==|~=|<=|>=|[-+*/%^#<>=(){}\[\];:,.])
""", re.X | re.S)

_KEYWORDS = {"and", "break", "do", "else", "elseif", "end", "false", "for", "function", "goto", "if", "in", "local", "nil", "not", "or",
             "repeat", "return", "then", "true", "until", "while"}
_ESCAPES = {"n": "\n", "t": "\t", "r": "\r", "\\": "\\", '"': '"', "'": "'", "a": "\a", "b": "\b", "f": "\f", "v": "\v", "0": "\0", "\n": "\n"}


def _tokenize(text, name):
    tokens, pos, line = [], 0, 1
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            raise LuaError("%s:%d: unexpected character %r" % (name, line, text[pos]))
        kind = m.lastgroup
        value = m.group(kind)
        if kind == "number":
            tokens.append(("number", float(int(value, 16)) if value[:2].lower() == "0x" else float(value), line))
        elif kind == "name":
            tokens.append(("keyword" if value in _KEYWORDS else "name", value, line))
        elif kind == "string":
            body = re.sub(r"\\(.)", lambda e: _ESCAPES.get(e.group(1), e.group(1)), value[1:-1], flags=re.S)
            tokens.append(("string", body, line))
        elif kind == "longstring":
            inner = value[value.index("[", 1) + 1:-(len(m.group("ls_eq")) + 2)]
            tokens.append(("string", inner[1:] if inner.startswith("\n") else inner, line))
        elif kind == "op":
            tokens.append(("op", value, line))
        line += value.count("\n")
        pos = m.end()
    tokens.append(("eof", None, line))
    return tokens


# ------------------------------------------------------------------------------------------------ values
class LuaTable:
    """A table as written in a project file: positional items and named fields (lib.lua leaves these as plain tables)."""

    def __init__(self):
        self.items, self.fields = [], {}

    def to_python(self):
        if self.fields and not self.items:
            return {k: _to_python(v) for k, v in self.fields.items()}
        if self.items and not self.fields:
            return [_to_python(v) for v in self.items]
        if not self.items and not self.fields:
            return {}
        out = {k: _to_python(v) for k, v in self.fields.items()}
        for i, v in enumerate(self.items):
            out[i + 1] = _to_python(v)
        return out


def _to_python(v):
    if isinstance(v, LuaTable):
        return v.to_python()
    if isinstance(v, float) and v.is_integer() and abs(v) < 2 ** 53:
        return v  # stays a float: Expression::Number is f64 in the reference; integer parameters are cast where they are read
    return v


class _Namespace:
    """A table of the prelude (`material`, `shape`, ...): attribute access only."""

    def __init__(self, source):
        self._source = source

    def get(self, key):
        try:
            return getattr(self._source, key)
        except AttributeError:
            raise LuaError("attempt to index a nil value (field '%s')" % key)


def _call(fn, args):
    """Lua call -> Python call: one table argument with named fields becomes keyword arguments (`shape.sphere {radius = 1}`),
    anything else stays positional (`vector(0, 1, 0)`, `texture("a.png", "linear")`). Parameters the script leaves out and
    that have no default (a template `shape.sphere` without material) are passed as None, as absent Lua fields are nil."""
    if len(args) == 1 and isinstance(args[0], LuaTable) and not args[0].items:
        kwargs = {("from_" if k == "from" else k): _to_python(v) for k, v in args[0].fields.items()}
        try:
            params = inspect.signature(fn).parameters
        except (TypeError, ValueError):
            params = {}
        accepts_extra = any(p.kind == p.VAR_KEYWORD for p in params.values())
        for name, p in params.items():
            if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY) and p.default is p.empty and name not in kwargs:
                kwargs[name] = None
        if not accepts_extra:  # fields the typed tree does not know are ignored, as typed_nodes ignores them (dragon.lua's `_ior`)
            kwargs = {k: v for k, v in kwargs.items() if k in params}
        return fn(**kwargs)
    return fn(*[_to_python(a) for a in args])


def _index(obj, key):
    if isinstance(obj, LuaTable):
        if isinstance(key, float) and key.is_integer() and 1 <= key <= len(obj.items):
            return obj.items[int(key) - 1]
        return obj.fields.get(key)
    if isinstance(obj, _Namespace):
        return obj.get(key)
    if isinstance(obj, P.Node):
        if key in ("with", "clone"):
            return _BoundMethod(obj, key)
        return obj.props.get("from_" if key == "from" else key)
    if isinstance(obj, dict):
        return obj.get(key)
    if isinstance(obj, list) and isinstance(key, float) and key.is_integer():
        return obj[int(key) - 1] if 1 <= key <= len(obj) else None
    raise LuaError("attempt to index a %s value" % _type_name(obj))


class _BoundMethod:
    def __init__(self, obj, name):
        self.obj, self.name = obj, name


def _method(obj, name, args):
    """obj:name(args). `with` / `clone` are the prelude's (lib.lua:44-74); plain tables get the same treatment."""
    if name == "with":
        changes = args[0] if args else LuaTable()
        if not isinstance(changes, LuaTable):
            raise LuaError(":with expects a table")
        if isinstance(obj, P.Node):
            return obj.with_(**{("from_" if k == "from" else k): _to_python(v) for k, v in changes.fields.items()})
        if isinstance(obj, LuaTable):
            out = LuaTable()
            out.items, out.fields = list(obj.items), dict(obj.fields)
            out.fields.update(changes.fields)
            return out
    if name == "clone":
        if isinstance(obj, P.Node):
            return obj.clone()
        if isinstance(obj, LuaTable):
            out = LuaTable()
            out.items, out.fields = list(obj.items), dict(obj.fields)
            return out
    if name == "mix" and isinstance(obj, P.Node):
        return P.mix(obj, *[_to_python(a) for a in args])
    raise LuaError("attempt to call method '%s' on a %s value" % (name, _type_name(obj)))


def _type_name(v):
    if v is None:
        return "nil"
    if isinstance(v, bool):
        return "boolean"
    if isinstance(v, float):
        return "number"
    if isinstance(v, str):
        return "string"
    return "table"


# ------------------------------------------------------------------------------------------------ parser / evaluator
_BINARY = [  # (operators, right associative), loosest first -- Lua 5.3 reference manual 3.4.8
    (("or",), False), (("and",), False), (("<", ">", "<=", ">=", "~=", "=="), False), (("..",), True), (("+", "-"), False),
    (("*", "/", "%"), False),
]


class _Evaluator:
    def __init__(self, text, name, env, base_dir, loading):
        self.tokens, self.pos, self.name = _tokenize(text, name), 0, name
        self.env, self.locals, self.base_dir, self.loading = env, {}, base_dir, loading

    # -- token helpers
    def peek(self):
        return self.tokens[self.pos]

    def next(self):
        tok = self.tokens[self.pos]
        self.pos += 1
        return tok

    def accept(self, kind, value=None):
        tok = self.peek()
        if tok[0] == kind and (value is None or tok[1] == value):
            self.pos += 1
            return tok
        return None

    def expect(self, kind, value=None):
        tok = self.accept(kind, value)
        if tok is None:
            got = self.peek()
            self.fail("expected %s, found %r" % (value or kind, got[1] if got[1] is not None else "end of file"))
        return tok

    def fail(self, message):
        raise LuaError("%s:%d: %s" % (self.name, self.peek()[2], message))

    # -- statements
    def run(self):
        while True:
            tok = self.peek()
            if tok[0] == "eof":
                return None
            if self.accept("op", ";"):
                continue
            if tok[0] == "keyword":
                if tok[1] == "return":
                    self.next()
                    value = None if self.peek()[0] == "eof" else self.expression_list()[0]  # a chunk's first return value
                    self.accept("op", ";")
                    if self.peek()[0] != "eof":
                        self.fail("'return' must be the last statement")
                    return value
                if tok[1] == "local":
                    self.next()
                    if self.peek() == ("keyword", "function", self.peek()[2]):
                        self.fail("function definitions are not supported in project files")
                    names = [self.expect("name")[1]]
                    while self.accept("op", ","):
                        names.append(self.expect("name")[1])
                    values = []
                    if self.accept("op", "="):
                        values = self.expression_list()
                    for i, n in enumerate(names):
                        self.locals[n] = values[i] if i < len(values) else None
                    continue
                self.fail("'%s' is not supported in project files (only assignments, calls and return are)" % tok[1])
            # assignment or call statement
            target = self.suffixed(allow_target=True)
            if isinstance(target, tuple) and target[0] == "target":
                self.expect("op", "=")
                value = self.expression()
                _, container, key = target
                if container is None:
                    (self.locals if key in self.locals else self.env)[key] = value
                elif isinstance(container, LuaTable):
                    container.fields[key] = value
                elif isinstance(container, P.Node):
                    container.props[key] = _to_python(value)
                else:
                    self.fail("cannot assign to a field of a %s value" % _type_name(container))

    def expression_list(self):
        values = [self.expression()]
        while self.accept("op", ","):
            values.append(self.expression())
        return values

    # -- expressions
    def expression(self, level=0):
        if level == len(_BINARY):
            return self.unary()
        ops, right = _BINARY[level]
        lhs = self.expression(level + 1)
        while True:
            tok = self.peek()
            if tok[0] in ("op", "keyword") and tok[1] in ops:
                self.next()
                rhs = self.expression(level if right else level + 1)
                lhs = self.binary(tok[1], lhs, rhs)
                if right:
                    return lhs
            else:
                return lhs

    def unary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] == "-":
            self.next()
            v = self.unary()
            return -v if isinstance(v, float) else self.binary("*", -1.0, v)
        if tok[0] == "keyword" and tok[1] == "not":
            self.next()
            v = self.unary()
            return v is None or v is False
        if tok[0] == "op" and tok[1] == "#":
            self.next()
            v = self.unary()
            return float(len(v.items if isinstance(v, LuaTable) else v))
        return self.power()

    def power(self):
        base = self.suffixed()
        if self.accept("op", "^"):
            return self.binary("^", base, self.unary())
        return base

    def binary(self, op, a, b):
        if op == "and":
            return b if not (a is None or a is False) else a
        if op == "or":
            return a if not (a is None or a is False) else b
        if op == "==":
            return a is b or (type(a) is type(b) and not isinstance(a, (P.Node, LuaTable)) and a == b)
        if op == "~=":
            return not self.binary("==", a, b)
        if op == "..":
            def text(v):
                return ("%d" % v if v.is_integer() else repr(v)) if isinstance(v, float) else str(v)
            return text(a) + text(b)
        try:
            if op == "+":
                return a + b
            if op == "-":
                return a - b
            if op == "*":
                return a * b
            if op == "/":
                return a / b
            if op == "%":
                return a % b
            if op == "^":
                return a ** b
            if op in ("<", ">", "<=", ">="):
                return {"<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b}[op]
        except TypeError:
            self.fail("attempt to perform arithmetic on a %s and a %s value" % (_type_name(a), _type_name(b)))
        except ZeroDivisionError:
            return float("inf") if (a > 0) == (b >= 0 and str(b) != "-0.0") else float("-inf")
        self.fail("unknown operator %s" % op)

    def primary(self):
        tok = self.next()
        if tok[0] == "number" or tok[0] == "string":
            return tok[1], None
        if tok[0] == "keyword":
            if tok[1] in ("nil", "true", "false"):
                return {"nil": None, "true": True, "false": False}[tok[1]], None
            if tok[1] == "function":
                self.pos -= 1
                self.fail("function definitions are not supported in project files")
        if tok[0] == "name":
            if tok[1] in self.locals:
                return self.locals[tok[1]], (None, tok[1])
            if tok[1] in self.env:
                return self.env[tok[1]], (None, tok[1])
            return None, (None, tok[1])
        if tok[0] == "op" and tok[1] == "(":
            v = self.expression()
            self.expect("op", ")")
            return v, None
        if tok[0] == "op" and tok[1] == "{":
            self.pos -= 1
            return self.table(), None
        self.pos -= 1
        self.fail("unexpected %r" % (tok[1] if tok[1] is not None else "end of file"))

    def suffixed(self, allow_target=False):
        value, target = self.primary()
        while True:
            tok = self.peek()
            if tok[0] == "op" and tok[1] == ".":
                self.next()
                key = self.next()
                if key[0] not in ("name", "keyword"):
                    self.fail("expected a field name")
                container = value
                if container is None:
                    self.fail("attempt to index a nil value")
                value, target = _index(container, key[1]), (container, key[1])
            elif tok[0] == "op" and tok[1] == "[":
                self.next()
                key = self.expression()
                self.expect("op", "]")
                container = value
                if container is None:
                    self.fail("attempt to index a nil value")
                value, target = _index(container, key), (container, key)
            elif tok[0] == "op" and tok[1] == ":":
                self.next()
                name = self.next()[1]
                args = self.call_arguments()
                value, target = self.guard(lambda: _method(value, name, args)), None
            elif (tok[0] == "op" and tok[1] in ("(", "{")) or tok[0] == "string":
                args = self.call_arguments()
                fn = value
                if fn is None:
                    self.fail("attempt to call a nil value%s" % (" (global '%s')" % target[1] if target and target[0] is None else ""))
                value, target = self.guard(lambda: self.call(fn, args)), None
            else:
                break
        if allow_target and target is not None and self.peek()[0] == "op" and self.peek()[1] == "=":
            return ("target", target[0], target[1])
        return value

    def guard(self, thunk):
        try:
            return thunk()
        except LuaError as e:
            if str(e).startswith(self.name + ":"):
                raise
            self.fail(str(e))
        except (TypeError, ValueError, P_ERRORS) as e:
            self.fail(str(e))

    def call(self, fn, args):
        if isinstance(fn, _BoundMethod):
            return _method(fn.obj, fn.name, args[1:] if args and args[0] is fn.obj else args)
        if fn is _REQUIRE:
            return self.require(args)
        if not callable(fn):
            raise LuaError("attempt to call a %s value" % _type_name(fn))
        return _call(fn, args)

    def call_arguments(self):
        tok = self.peek()
        if tok[0] == "string":
            self.next()
            return [tok[1]]
        if tok[0] == "op" and tok[1] == "{":
            return [self.table()]
        self.expect("op", "(")
        args = []
        if not self.accept("op", ")"):
            args = self.expression_list()
            self.expect("op", ")")
        return args

    def table(self):
        self.expect("op", "{")
        t = LuaTable()
        while not self.accept("op", "}"):
            tok = self.peek()
            if tok[0] == "name" and self.tokens[self.pos + 1][:2] == ("op", "="):
                self.next()
                self.next()
                t.fields[tok[1]] = self.expression()
            elif tok[0] == "op" and tok[1] == "[":
                self.next()
                key = self.expression()
                self.expect("op", "]")
                self.expect("op", "=")
                t.fields[key] = self.expression()
            else:
                t.items.append(self.expression())
            if not (self.accept("op", ",") or self.accept("op", ";")):
                self.expect("op", "}")
                break
        return t

    # -- require: a module is another file of the same kind next to the project (mlua's package.path is the project directory)
    def require(self, args):
        if len(args) != 1 or not isinstance(args[0], str):
            raise LuaError("require expects a module name")
        path = os.path.join(self.base_dir, args[0].replace(".", os.sep) + ".lua")
        key = os.path.normpath(path)
        cache = self.env["__modules__"]
        if key not in cache:
            if key in self.loading:
                raise LuaError("circular require of '%s'" % args[0])
            if not os.path.exists(path):
                raise LuaError("module '%s' not found (looked for %s)" % (args[0], path))
            with open(path) as f:
                text = f.read()
            cache[key] = _Evaluator(text, os.path.basename(path), self.env, self.base_dir, self.loading | {key}).run()
        return cache[key]


P_ERRORS = (AttributeError, KeyError)
_REQUIRE = object()


def _environment():
    env = {name: getattr(P, name) for name in ("vector", "rgb", "spectrum", "blackbody", "fresnel", "mix", "texture")}
    for name in ("light_source", "material", "shape", "light", "transform", "camera", "renderer", "ray_marched", "bounds", "quaternion_julia"):
        env[name] = _Namespace(getattr(P, name))
    env["require"] = _REQUIRE
    env["__modules__"] = {}
    return env


def evaluate(text, name="project.lua", base_dir="."):
    """Evaluates one project file's text; returns what it `return`s as plain Python data (dicts, lists, numbers, strings and
    pyrite_amd.project nodes)."""
    try:
        return _to_python(_Evaluator(text, name, _environment(), base_dir, frozenset()).run())
    except RecursionError:  # Lua itself refuses chunks nested deeper than 200 levels (LUAI_MAXCCALLS); the C++ reader counts them
        raise LuaError("%s: chunk has too many syntax levels" % name) from None


def load_project(path):
    """-> (project dict as pyrite_amd.scenes.build takes it, directory that mesh / texture paths are relative to)."""
    path = os.fspath(path)
    base_dir = os.path.dirname(os.path.abspath(path))
    with open(path) as f:
        project = evaluate(f.read(), os.path.basename(path), base_dir)
    if not isinstance(project, dict) or "world" not in project:
        raise LuaError("%s: a project file must return a table with at least `world`, `camera` and `renderer`" % path)
    image = project.get("image") or {}
    for key in ("width", "height"):
        if key in image:
            image[key] = int(image[key])
    world = project["world"]
    if isinstance(world.get("objects"), dict) and not world["objects"]:
        world["objects"] = []
    return project, base_dir
