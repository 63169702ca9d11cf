"""A small interpreter for the GLSL subset the reference's shader bodies use (glbuild/glsllib/*.glsl and the strings its
AppendShaderBody methods emit): float / int / bool / vec2-4 / bvec / mat2-4 / arrays, swizzles (read and write), if / else, for,
the ternary, compound assignment, user functions (by value), the usual builtins. Test infrastructure: it lets the CPU suite evaluate
the reference's OWN second statement of every formula -- the GLSL its GPU path compiles -- beside the oracle (tests/
test_glsl_crosscheck.py). Arithmetic is binary64: the comparison tolerance is the reference's own CPU <-> GPU 5e-3 (gsdf_test.go:529)."""
import math
import re

import numpy as np

TYPES = {"float", "int", "bool", "vec2", "vec3", "vec4", "bvec2", "bvec3", "bvec4", "ivec2", "ivec3", "ivec4", "mat2", "mat3", "mat4", "void", "uint"}
TOK = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+|\d+)[fFuU]?|([A-Za-z_]\w*)|(\+\+|--|<=|>=|==|!=|&&|\|\||\+=|-=|\*=|/=|[-+*/%<>=!?:,;(){}\[\].]))")


class GLSLError(Exception):
    pass


def preprocess(src):
    """Comments out; object-like #define / #undef applied textually, in order (the bodies use `#define Pi 3.14...`, `#define v buf`)."""
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    out, macros = [], {}
    for line in src.split("\n"):
        m = re.match(r"\s*#\s*define\s+(\w+)\s+(.*?)\s*$", line)
        if m:
            macros[m.group(1)] = m.group(2)
            continue
        m = re.match(r"\s*#\s*undef\s+(\w+)", line)
        if m:
            macros.pop(m.group(1), None)
            continue
        if line.lstrip().startswith("#"):
            continue
        for k, v in macros.items():
            line = re.sub(r"\b%s\b" % re.escape(k), v, line)
        out.append(line)
    return "\n".join(out)


def tokenize(src):
    toks, i = [], 0
    src = src.rstrip()
    while i < len(src):
        m = TOK.match(src, i)
        if not m:
            if src[i:].strip() == "":
                break
            raise GLSLError("bad token at %r" % src[i:i + 30])
        i = m.end()
        if m.group(1) is not None:
            t = m.group(1)
            toks.append(("num", float(t) if re.search(r"[.eE]", t) else int(t)))
        elif m.group(2) is not None:
            toks.append(("id", m.group(2)))
        else:
            toks.append(("op", m.group(3)))
    toks.append(("eof", None))
    return toks


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k]

    def next(self):
        tok = self.t[self.i]
        self.i += 1
        return tok

    def accept(self, v):
        if self.peek()[1] == v and self.peek()[0] in ("op", "id"):
            self.i += 1
            return True
        return False

    def expect(self, v):
        if not self.accept(v):
            raise GLSLError("expected %r, got %r (token %d)" % (v, self.peek(), self.i))

    def is_type(self, k=0):
        tok = self.peek(k)
        return tok[0] == "id" and tok[1] in TYPES

    # ---- top level
    def unit(self):
        funcs = {}
        while self.peek()[0] != "eof":
            while self.accept("const") or self.accept("highp") or self.accept("in"):
                pass
            rtype = self.next()[1]
            name = self.next()[1]
            self.expect("(")
            params = []
            while not self.accept(")"):
                while self.peek()[1] in ("in", "const", "highp"):
                    self.next()
                if self.peek()[1] in ("out", "inout"):
                    raise GLSLError("out / inout parameters are not supported")
                ptype = self.next()[1]
                pname = self.next()[1]
                params.append((ptype, pname))
                self.accept(",")
            body = self.block()
            funcs[name] = (rtype, params, body)
        return funcs

    def block(self):
        self.expect("{")
        st = []
        while not self.accept("}"):
            st.append(self.statement())
        return ("block", st)

    def statement(self):
        tok = self.peek()
        if tok == ("op", "{"):
            return self.block()
        if tok == ("op", ";"):
            self.next()
            return ("block", [])
        if tok == ("id", "if"):
            self.next()
            self.expect("(")
            c = self.expr()
            self.expect(")")
            a = self.statement()
            b = self.statement() if self.accept("else") else None
            return ("if", c, a, b)
        if tok == ("id", "for"):
            self.next()
            self.expect("(")
            init = self.simple() if not self.accept(";") else ("block", [])
            cond = self.expr() if self.peek() != ("op", ";") else ("num", 1)
            self.expect(";")
            it = []
            while self.peek() != ("op", ")"):
                it.append(self.assign())
                self.accept(",")
            self.expect(")")
            return ("for", init, cond, it, self.statement())
        if tok == ("id", "return"):
            self.next()
            e = None if self.peek() == ("op", ";") else self.expr()
            self.expect(";")
            return ("return", e)
        return self.simple()

    def simple(self):
        """declaration or expression statement, terminated by ';'"""
        while self.peek()[1] in ("const", "highp"):
            self.next()
        if self.is_type() and self.peek(1)[0] == "id":
            typ = self.next()[1]
            decls = []
            while True:
                name = self.next()[1]
                size = None
                if self.accept("["):
                    size = None if self.peek() == ("op", "]") else self.expr()
                    self.expect("]")
                init = self.assign() if self.accept("=") else None
                decls.append((name, size, init))
                if not self.accept(","):
                    break
            self.expect(";")
            return ("decl", typ, decls)
        e = self.expr()
        self.expect(";")
        return ("expr", e)

    # ---- expressions
    def expr(self):
        e = self.assign()
        while self.accept(","):
            e = ("seq", e, self.assign())
        return e

    def assign(self):
        lhs = self.ternary()
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("=", "+=", "-=", "*=", "/="):
            self.next()
            return ("assign", tok[1], lhs, self.assign())
        return lhs

    def ternary(self):
        c = self.binary(0)
        if self.accept("?"):
            a = self.assign()
            self.expect(":")
            return ("tern", c, a, self.assign())
        return c

    LEVELS = [("||",), ("&&",), ("==", "!="), ("<", ">", "<=", ">="), ("+", "-"), ("*", "/", "%")]

    def binary(self, lvl):
        if lvl == len(self.LEVELS):
            return self.unary()
        e = self.binary(lvl + 1)
        while self.peek()[0] == "op" and self.peek()[1] in self.LEVELS[lvl]:
            op = self.next()[1]
            e = ("bin", op, e, self.binary(lvl + 1))
        return e

    def unary(self):
        tok = self.peek()
        if tok[0] == "op" and tok[1] in ("-", "+", "!"):
            self.next()
            return ("un", tok[1], self.unary())
        if tok[0] == "op" and tok[1] in ("++", "--"):
            self.next()
            t = self.unary()
            return ("assign", "+=" if tok[1] == "++" else "-=", t, ("num", 1))
        return self.postfix()

    def postfix(self):
        e = self.primary()
        while True:
            if self.accept("["):
                i = self.expr()
                self.expect("]")
                e = ("index", e, i)
            elif self.accept("."):
                name = self.next()[1]
                if name == "length" and self.accept("("):
                    self.expect(")")
                    e = ("len", e)
                else:
                    e = ("member", e, name)
            elif self.peek()[0] == "op" and self.peek()[1] in ("++", "--"):
                op = self.next()[1]
                e = ("postinc", e, 1 if op == "++" else -1)
            else:
                return e

    def primary(self):
        tok = self.next()
        if tok[0] == "num":
            return ("num", tok[1])
        if tok == ("op", "("):
            e = self.expr()
            self.expect(")")
            return e
        if tok[0] == "id":
            name = tok[1]
            if name in ("true", "false"):
                return ("num", name == "true")
            if name in TYPES and self.peek() == ("op", "["):  # array constructor: vec2[](a, b, ...)  /  vec2[3](...)
                self.next()
                if self.peek() != ("op", "]"):
                    self.expr()
                self.expect("]")
                self.expect("(")
                return ("array", name, self.args())
            if self.accept("("):
                return ("call", name, self.args())
            return ("var", name)
        raise GLSLError("unexpected token %r" % (tok,))

    def args(self):
        a = []
        while not self.accept(")"):
            a.append(self.assign())
            self.accept(",")
        return a


class Return(Exception):
    def __init__(self, v):
        self.v = v


SW = {c: i for i, c in enumerate("xyzw")}
SW.update({c: i for i, c in enumerate("rgba")})
SW.update({c: i for i, c in enumerate("stpq")})


def _isvec(v):
    return isinstance(v, np.ndarray) and v.ndim == 1


def _ismat(v):
    return isinstance(v, np.ndarray) and v.ndim == 2


def _cw(f):
    """componentwise builtin with scalar broadcast"""
    def g(*a):
        if any(isinstance(x, np.ndarray) for x in a):
            n = max(len(x) for x in a if isinstance(x, np.ndarray))
            cols = [x if isinstance(x, np.ndarray) else np.full(n, float(x)) for x in a]
            return np.array([f(*[float(c[i]) for c in cols]) for i in range(n)])
        return f(*[float(x) for x in a])
    return g


def _sign(x):
    return 0.0 if x == 0 else math.copysign(1.0, x)


def _pow(x, y):
    if x < 0 or (x == 0 and y <= 0):
        return float("nan")  # undefined in GLSL
    return x ** y


def _round(x):
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)  # half away from zero (GLSL: implementation's choice; the CPU path uses math.Round)


def _safe(f):
    def g(*a):
        try:
            return f(*a)
        except (ValueError, ZeroDivisionError, OverflowError):
            return float("nan")
    return g


BUILTINS = {
    "abs": _cw(abs), "sign": _cw(_sign), "floor": _cw(math.floor), "ceil": _cw(math.ceil), "round": _cw(_round), "fract": _cw(lambda x: x - math.floor(x)),
    "sqrt": _cw(_safe(math.sqrt)), "inversesqrt": _cw(_safe(lambda x: 1 / math.sqrt(x))), "sin": _cw(math.sin), "cos": _cw(math.cos), "tan": _cw(math.tan),
    "acos": _cw(_safe(math.acos)), "asin": _cw(_safe(math.asin)), "exp": _cw(_safe(math.exp)), "log": _cw(_safe(math.log)),
    "pow": _cw(_safe(_pow)), "mod": _cw(_safe(lambda x, y: x - y * math.floor(x / y))),
    "min": _cw(min), "max": _cw(max), "clamp": _cw(lambda x, a, b: min(max(x, a), b)), "mix": _cw(lambda x, y, a: x * (1 - a) + y * a),
    "step": _cw(lambda e, x: 0.0 if x < e else 1.0),
    "atan": lambda *a: _cw(math.atan2)(*a) if len(a) == 2 else _cw(math.atan)(*a),
    "length": lambda v: float(math.sqrt(float(np.dot(v, v)))) if _isvec(v) else abs(float(v)),
    "dot": lambda a, b: float(np.dot(a, b)) if _isvec(a) else float(a) * float(b),
    "distance": lambda a, b: float(math.sqrt(float(np.dot(a - b, a - b)))),
    "normalize": lambda v: v / math.sqrt(float(np.dot(v, v))),
    "cross": lambda a, b: np.cross(a, b),
    "all": lambda v: bool(np.all(v)), "any": lambda v: bool(np.any(v)), "not": lambda v: np.logical_not(v),
}


class Interp:
    def __init__(self, src):
        self.funcs = Parser(tokenize(preprocess(src))).unit()
        self.calls = 0

    # ---- values
    def construct(self, typ, args):
        if typ in ("float", "int", "uint", "bool"):
            v = args[0]
            v = v[0] if isinstance(v, np.ndarray) else v
            return float(v) if typ == "float" else (int(v) if typ != "bool" else bool(v))
        n = int(typ[-1])
        if typ.startswith("mat"):
            flat = []
            for a in args:
                flat += list(np.ravel(a, order="F")) if isinstance(a, np.ndarray) else [float(a)]
            if len(flat) == 1:
                return np.eye(n) * flat[0]
            if len(flat) != n * n:
                raise GLSLError("%s from %d values" % (typ, len(flat)))
            return np.array(flat, dtype=np.float64).reshape(n, n).T.copy()  # column major: m[col][row]; stored [row, col]
        flat = []
        for a in args:
            flat += list(a) if isinstance(a, np.ndarray) else [a]
        if len(flat) == 1:
            flat = flat * n
        if len(flat) < n:
            raise GLSLError("%s from %d values" % (typ, len(flat)))
        dt = np.bool_ if typ.startswith("bvec") else np.float64
        return np.array(flat[:n], dtype=dt)

    def default(self, typ):
        if typ in ("float",):
            return 0.0
        if typ in ("int", "uint"):
            return 0
        if typ == "bool":
            return False
        return self.construct(typ, [0.0])

    # ---- evaluation
    def call(self, name, args):
        self.calls += 1
        if name in self.funcs:
            rtype, params, body = self.funcs[name]
            if len(params) != len(args):
                raise GLSLError("%s: %d arguments for %d parameters" % (name, len(args), len(params)))
            env = [{}]
            for (ptype, pname), a in zip(params, args):
                env[0][pname] = a.copy() if isinstance(a, np.ndarray) else (float(a) if ptype == "float" else a)
            try:
                self.exec(body, env)
            except Return as r:
                return r.v
            return None
        if name in TYPES:
            return self.construct(name, args)
        if name in BUILTINS:
            return BUILTINS[name](*args)
        raise GLSLError("unknown function %s" % name)

    def lookup(self, env, name):
        for sc in reversed(env):
            if name in sc:
                return sc
        raise GLSLError("undeclared %s" % name)

    def exec(self, st, env):
        k = st[0]
        if k == "block":
            env.append({})
            try:
                for s in st[1]:
                    self.exec(s, env)
            finally:
                env.pop()
        elif k == "decl":
            _, typ, decls = st
            for name, size, init in decls:
                if init is not None:
                    v = self.eval(init, env)
                    if isinstance(v, np.ndarray):
                        v = v.copy()
                    elif typ == "float":
                        v = float(v)
                    elif typ == "int":
                        v = int(v)
                elif size is not None:
                    v = [self.default(typ) for _ in range(int(self.eval(size, env)))]
                else:
                    v = self.default(typ)
                env[-1][name] = v
        elif k == "expr":
            self.eval(st[1], env)
        elif k == "if":
            if self.truth(self.eval(st[1], env)):
                self.exec(st[2], env)
            elif st[3] is not None:
                self.exec(st[3], env)
        elif k == "for":
            env.append({})
            try:
                self.exec(st[1], env) if st[1][0] != "decl" else self.exec_flat(st[1], env)
                n = 0
                while self.truth(self.eval(st[2], env)):
                    self.exec(st[4], env)
                    for it in st[3]:
                        self.eval(it, env)
                    n += 1
                    if n > 100000:
                        raise GLSLError("runaway loop")
            finally:
                env.pop()
        elif k == "return":
            raise Return(None if st[1] is None else self.eval(st[1], env))
        else:
            raise GLSLError("statement %s" % k)

    def exec_flat(self, st, env):
        self.exec(st, env)  # a declaration lands in env[-1]: the loop's own scope

    @staticmethod
    def truth(v):
        if isinstance(v, np.ndarray):
            raise GLSLError("vector used as a condition")
        return bool(v)

    def eval(self, e, env):
        k = e[0]
        if k == "num":
            return e[1]
        if k == "var":
            return self.lookup(env, e[1])[e[1]]
        if k == "seq":
            self.eval(e[1], env)
            return self.eval(e[2], env)
        if k == "un":
            v = self.eval(e[2], env)
            return (not self.truth(v)) if e[1] == "!" else (-v if e[1] == "-" else v)
        if k == "bin":
            op = e[1]
            if op == "&&":
                return self.truth(self.eval(e[2], env)) and self.truth(self.eval(e[3], env))
            if op == "||":
                return self.truth(self.eval(e[2], env)) or self.truth(self.eval(e[3], env))
            return self.binop(op, self.eval(e[2], env), self.eval(e[3], env))
        if k == "tern":
            return self.eval(e[2], env) if self.truth(self.eval(e[1], env)) else self.eval(e[3], env)
        if k == "call":
            return self.call(e[1], [self.eval(a, env) for a in e[2]])
        if k == "array":
            return [self.eval(a, env) for a in e[2]]
        if k == "len":
            return len(self.eval(e[1], env))
        if k == "index":
            b, i = self.eval(e[1], env), int(self.eval(e[2], env))
            if _ismat(b):
                return b[:, i].copy()  # m[i] is column i
            v = b[i]
            return float(v) if isinstance(b, np.ndarray) and b.dtype != np.bool_ else v
        if k == "member":
            b = self.eval(e[1], env)
            idx = [SW[c] for c in e[2]]
            if len(idx) == 1:
                v = b[idx[0]]
                return bool(v) if b.dtype == np.bool_ else float(v)
            return b[idx].copy()
        if k == "postinc":
            old = self.eval(e[1], env)
            self.store(e[1], old + e[2], env)
            return old
        if k == "assign":
            op, lhs = e[1], e[2]
            v = self.eval(e[3], env)
            if op != "=":
                v = self.binop(op[0], self.eval(lhs, env), v)
            self.store(lhs, v, env)
            return v
        raise GLSLError("expression %s" % k)

    def store(self, lhs, v, env):
        if lhs[0] == "var":
            sc = self.lookup(env, lhs[1])
            old = sc[lhs[1]]
            if isinstance(v, np.ndarray):
                v = v.copy()
            elif isinstance(old, float):
                v = float(v)
            elif isinstance(old, int) and not isinstance(old, bool):
                v = int(v)
            sc[lhs[1]] = v
        elif lhs[0] == "member":
            base = self.eval(lhs[1], env).copy()
            idx = [SW[c] for c in lhs[2]]
            if len(idx) == 1:
                base[idx[0]] = v
            else:
                base[idx] = v
            self.store(lhs[1], base, env)
        elif lhs[0] == "index":
            base = self.eval(lhs[1], env)
            i = int(self.eval(lhs[2], env))
            if isinstance(base, list):
                base[i] = v.copy() if isinstance(v, np.ndarray) else v
            elif _ismat(base):
                b = base.copy()
                b[:, i] = v
                self.store(lhs[1], b, env)
            else:
                b = base.copy()
                b[i] = v
                self.store(lhs[1], b, env)
        else:
            raise GLSLError("not assignable: %s" % lhs[0])

    @staticmethod
    def binop(op, a, b):
        if op in ("<", ">", "<=", ">=", "==", "!="):
            if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
                if op in ("==", "!="):
                    eq = bool(np.all(np.asarray(a) == np.asarray(b)))
                    return eq if op == "==" else not eq
                raise GLSLError("relational operator on vectors")
            return {"<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b, "==": a == b, "!=": a != b}[op]
        if op == "*" and (_ismat(a) or _ismat(b)):
            if _ismat(a) and (_isvec(b) or _ismat(b)):
                return a @ b
            if _isvec(a) and _ismat(b):
                return a @ b
            return a * b  # matrix times scalar
        if op == "+":
            return a + b
        if op == "-":
            return a - b
        if op == "*":
            return a * b
        if op == "/":
            if isinstance(a, int) and isinstance(b, int) and not isinstance(a, bool):
                return int(a / b) if b != 0 else 0
            with np.errstate(divide="ignore", invalid="ignore"):
                if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
                    return np.asarray(a, dtype=np.float64) / np.asarray(b, dtype=np.float64)
                if b == 0:
                    return float("nan") if a == 0 else math.copysign(float("inf"), a) * (math.copysign(1.0, b) if isinstance(b, float) else 1.0)
                return a / b
        if op == "%":
            return a % b
        raise GLSLError("operator %s" % op)
