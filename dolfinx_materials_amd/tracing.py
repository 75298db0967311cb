"""Turns a Python ``yield_stress(p)`` callable into C expressions for R(p) and dR/dp.

jaxmat takes the hardening law as an arbitrary Python callable and lets ``jax.jit`` trace and
compile it on the first pass (reference ``tests/test_FeFp_jax.py:14-19``,
``dolfinx_materials/jaxmat.py:152-155, :214-216``).  The counterpart here: the callable is run ONCE
on a symbolic scalar (:class:`Sym`) that records the arithmetic -- Python operators plus numpy
ufuncs through ``__array_ufunc__`` -- the recorded expression is differentiated symbolically, and
both are emitted as C and compiled into the fused gfx950 kernels by ``_lib.load_custom``
(``-DDXM_CUSTOM_HARDENING``).  Closed-over Python numbers become literals of the expression, exactly
as they become constants of the jitted XLA computation in the reference.

Supported: ``+ - * / **``, unary minus, ``abs``, and the numpy ufuncs in :data:`UFUNCS`
(``exp expm1 log log1p sqrt cbrt tanh sinh cosh sin cos arctan square reciprocal power maximum
minimum`` ...).  Anything else (comparisons, ``float(p)``, ``math.exp``) raises a ``TypeError`` that
says what to use instead.
"""
from __future__ import annotations

import math

import numpy as np

# ----------------------------------------------------------------------------------------------
# expression nodes: tuples ("const", v) | ("var",) | (op, a[, b]) with smart constructors that fold
# constants, so that e.g. d/dp [c1 (1 - exp(-b p))] comes out as (c1 b) exp(-b p): one rounding,
# like the hand-written Voce kernel.
# ----------------------------------------------------------------------------------------------
VAR = ("var",)


def const(v):
    return ("const", float(v))


def is_const(n, v=None):
    return n[0] == "const" and (v is None or n[1] == v)


def _coeff(n):
    """n = c * rest with a numeric c (rest None for a pure constant)."""
    if n[0] == "const":
        return n[1], None
    if n[0] == "mul" and n[1][0] == "const":
        return n[1][1], n[2]
    return 1.0, n


def _scaled(c, rest):
    if rest is None:
        return const(c)
    if c == 0.0:
        return const(0.0)
    if c == 1.0:
        return rest
    return ("mul", const(c), rest)


def add(a, b):
    if is_const(a) and is_const(b):
        return const(a[1] + b[1])
    if is_const(a, 0.0):
        return b
    if is_const(b, 0.0):
        return a
    return ("add", a, b)


def sub(a, b):
    if is_const(a) and is_const(b):
        return const(a[1] - b[1])
    if is_const(b, 0.0):
        return a
    if is_const(a, 0.0):
        return neg(b)
    return ("sub", a, b)


def neg(a):
    c, r = _coeff(a)
    return _scaled(-c, r)


def mul(a, b):
    ca, ra = _coeff(a)
    cb, rb = _coeff(b)
    c = ca * cb
    if ra is None or rb is None:
        return _scaled(c, ra if rb is None else rb)
    return _scaled(c, ("mul", ra, rb))


def div(a, b):
    if is_const(a) and is_const(b):
        return const(a[1] / b[1])
    if is_const(a, 0.0):
        return const(0.0)
    if is_const(b):
        if b[1] != 0.0 and math.frexp(b[1])[0] in (0.5, -0.5):   # power of two: the reciprocal is exact
            return mul(a, const(1.0 / b[1]))
        return ("div", a, b)
    ca, ra = _coeff(a)
    cb, rb = _coeff(b)
    return _scaled(ca / cb, ("div", const(1.0) if ra is None else ra, rb))


def power(a, b):
    if is_const(a) and is_const(b):
        return const(a[1] ** b[1])
    if is_const(b, 1.0):
        return a
    if is_const(b, 0.0):
        return const(1.0)
    if is_const(b, 2.0):
        return mul(a, a)
    return ("pow", a, b)


def fn(name, a):
    if is_const(a):
        return const(_NUMPY_FN[name](a[1]))
    return ("fn", name, a)


def select(a, b, x, y):
    """(a >= b ? x : y)"""
    if x == y:
        return x
    if is_const(a) and is_const(b):
        return x if a[1] >= b[1] else y
    return ("sel", a, b, x, y)


_NUMPY_FN = {
    "exp": np.exp, "expm1": np.expm1, "log": np.log, "log1p": np.log1p, "sqrt": np.sqrt, "cbrt": np.cbrt,
    "tanh": np.tanh, "sinh": np.sinh, "cosh": np.cosh, "sin": np.sin, "cos": np.cos, "atan": np.arctan,
    "fabs": np.abs,
}


def diff(n):
    """d n / d p."""
    op = n[0]
    if op == "const":
        return const(0.0)
    if op == "var":
        return const(1.0)
    if op == "add":
        return add(diff(n[1]), diff(n[2]))
    if op == "sub":
        return sub(diff(n[1]), diff(n[2]))
    if op == "mul":
        return add(mul(diff(n[1]), n[2]), mul(n[1], diff(n[2])))
    if op == "div":
        a, b = n[1], n[2]
        return sub(div(diff(a), b), div(mul(a, diff(b)), mul(b, b)))
    if op == "pow":
        a, b = n[1], n[2]
        if is_const(b):
            return mul(mul(b, power(a, const(b[1] - 1.0))), diff(a))
        if is_const(a):
            return mul(mul(n, const(math.log(a[1]))), diff(b))
        return mul(n, add(mul(diff(b), fn("log", a)), div(mul(b, diff(a)), a)))
    if op == "sel":
        return select(n[1], n[2], diff(n[3]), diff(n[4]))
    if op == "fn":
        name, u = n[1], n[2]
        du = diff(u)
        if name in ("exp", "expm1"):
            return mul(fn("exp", u), du)
        if name == "log":
            return div(du, u)
        if name == "log1p":
            return div(du, add(const(1.0), u))
        if name == "sqrt":
            return div(du, mul(const(2.0), n))
        if name == "cbrt":
            return div(du, mul(const(3.0), mul(n, n)))
        if name == "tanh":
            return mul(sub(const(1.0), mul(n, n)), du)
        if name == "sinh":
            return mul(fn("cosh", u), du)
        if name == "cosh":
            return mul(fn("sinh", u), du)
        if name == "sin":
            return mul(fn("cos", u), du)
        if name == "cos":
            return neg(mul(fn("sin", u), du))
        if name == "atan":
            return div(du, add(const(1.0), mul(u, u)))
        if name == "fabs":
            return mul(select(u, const(0.0), const(1.0), const(-1.0)), du)
    raise TypeError(f"cannot differentiate node {op!r}")


def _lit(v):
    if v != v or v in (math.inf, -math.inf):
        raise ValueError("the traced hardening law contains a non-finite constant")
    if v == int(v) and abs(v) < 2**53:
        return f"{int(v)}.0" if v >= 0 else f"(-{int(-v)}.0)"
    h = float(v).hex()  # exact C99 / C++17 hexadecimal floating literal
    return h if v >= 0 else f"({h})"


def emit_c(n):
    op = n[0]
    if op == "const":
        return _lit(n[1])
    if op == "var":
        return "p"
    if op in ("add", "sub", "mul", "div"):
        sym = {"add": "+", "sub": "-", "mul": "*", "div": "/"}[op]
        if op == "mul" and is_const(n[1], -1.0):
            return f"(-{emit_c(n[2])})"
        if op == "mul":   # individually rounded, never contracted into an FMA (dxm_common.hpp: DXM_MUL)
            return f"DXM_MUL({emit_c(n[1])}, {emit_c(n[2])})"
        return f"({emit_c(n[1])} {sym} {emit_c(n[2])})"
    if op == "pow":
        return f"pow({emit_c(n[1])}, {emit_c(n[2])})"
    if op == "fn":
        return f"{n[1]}({emit_c(n[2])})"
    if op == "sel":
        return f"(({emit_c(n[1])} >= {emit_c(n[2])}) ? {emit_c(n[3])} : {emit_c(n[4])})"
    raise TypeError(op)


def evaluate(n, p):
    """numpy evaluation of a node at ``p`` (array or scalar): the host-side check of a trace."""
    op = n[0]
    if op == "const":
        return np.full(np.shape(p), n[1]) if np.ndim(p) else n[1]
    if op == "var":
        return np.asarray(p, dtype=np.float64) if np.ndim(p) else float(p)
    if op == "add":
        return evaluate(n[1], p) + evaluate(n[2], p)
    if op == "sub":
        return evaluate(n[1], p) - evaluate(n[2], p)
    if op == "mul":
        return evaluate(n[1], p) * evaluate(n[2], p)
    if op == "div":
        return evaluate(n[1], p) / evaluate(n[2], p)
    if op == "pow":
        return np.power(evaluate(n[1], p), evaluate(n[2], p))
    if op == "fn":
        return _NUMPY_FN[n[1]](evaluate(n[2], p))
    if op == "sel":
        return np.where(evaluate(n[1], p) >= evaluate(n[2], p), evaluate(n[3], p), evaluate(n[4], p))
    raise TypeError(op)


# ----------------------------------------------------------------------------------------------
# the symbolic scalar handed to the user's callable
# ----------------------------------------------------------------------------------------------
def _node(x):
    if isinstance(x, Sym):
        return x.node
    if isinstance(x, (bool, int, float, np.integer, np.floating)):
        return const(x)
    if isinstance(x, np.ndarray) and x.ndim == 0:
        return const(x.item())
    raise TypeError(f"a traced yield_stress(p) can only combine p with Python / numpy scalars, got {type(x).__name__}")


_UNARY = {
    "exp": "exp", "expm1": "expm1", "log": "log", "log1p": "log1p", "sqrt": "sqrt", "cbrt": "cbrt", "tanh": "tanh",
    "sinh": "sinh", "cosh": "cosh", "sin": "sin", "cos": "cos", "arctan": "atan", "absolute": "fabs", "fabs": "fabs",
}
_BINARY = {"add": add, "subtract": sub, "multiply": mul, "divide": div, "true_divide": div, "power": power,
           "float_power": power}
#: names of the numpy ufuncs a traced callable may apply to p
UFUNCS = sorted(list(_UNARY) + list(_BINARY) + ["negative", "positive", "square", "reciprocal", "maximum", "minimum"])


class Sym:
    """Symbolic stand-in for the cumulated plastic strain ``p`` while the callable is traced."""

    __array_priority__ = 1000.0

    def __init__(self, node):
        self.node = node

    # numpy ufuncs (np.exp(p), np.float64(2) * p, np.maximum(p, 0) ...)
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            return NotImplemented
        name = ufunc.__name__
        a = [_node(x) for x in inputs]
        if name in _UNARY:
            return Sym(fn(_UNARY[name], a[0]))
        if name in _BINARY:
            return Sym(_BINARY[name](a[0], a[1]))
        if name == "negative":
            return Sym(neg(a[0]))
        if name == "positive":
            return Sym(a[0])
        if name == "square":
            return Sym(mul(a[0], a[0]))
        if name == "reciprocal":
            return Sym(div(const(1.0), a[0]))
        if name == "maximum":
            return Sym(select(a[0], a[1], a[0], a[1]))
        if name == "minimum":
            return Sym(select(a[1], a[0], a[0], a[1]))
        raise TypeError(f"numpy.{name} is not supported inside a traced yield_stress(p); supported ufuncs: {', '.join(UFUNCS)}")

    def __add__(self, o): return Sym(add(self.node, _node(o)))
    def __radd__(self, o): return Sym(add(_node(o), self.node))
    def __sub__(self, o): return Sym(sub(self.node, _node(o)))
    def __rsub__(self, o): return Sym(sub(_node(o), self.node))
    def __mul__(self, o): return Sym(mul(self.node, _node(o)))
    def __rmul__(self, o): return Sym(mul(_node(o), self.node))
    def __truediv__(self, o): return Sym(div(self.node, _node(o)))
    def __rtruediv__(self, o): return Sym(div(_node(o), self.node))
    def __pow__(self, o): return Sym(power(self.node, _node(o)))
    def __rpow__(self, o): return Sym(power(_node(o), self.node))
    def __neg__(self): return Sym(neg(self.node))
    def __pos__(self): return self
    def __abs__(self): return Sym(fn("fabs", self.node))

    def _no(self, what):
        raise TypeError(f"{what} is not traceable: write yield_stress(p) with + - * / ** and numpy ufuncs "
                        "(np.exp, np.sqrt, np.maximum, ...), or pass a materials.CustomHardening")

    def __float__(self): self._no("float(p) (e.g. math.exp(p))")
    def __int__(self): self._no("int(p)")
    def __bool__(self): self._no("branching on p (if p > ...)")
    def __lt__(self, o): self._no("comparing p")
    __le__ = __gt__ = __ge__ = __lt__
    def __array__(self, *a, **k): self._no("converting p to an array (jax.numpy / non-ufunc numpy functions)")

    # jnp-style method spellings some laws use
    def exp(self): return Sym(fn("exp", self.node))
    def sqrt(self): return Sym(fn("sqrt", self.node))


def trace(func):
    """Run ``func`` on a symbolic ``p``; returns ``(R_node, dR_node)``."""
    out = func(Sym(VAR))
    R = _node(out)
    return R, diff(R)


# ----------------------------------------------------------------------------------------------
# jax: a callable written with jax.numpy (what the reference's users have, tests/test_FeFp_jax.py:14-15) is traced by
# jax itself -- jax.make_jaxpr(yield_stress)(0.0) -- and its primitives are translated into the node tuples above
# ----------------------------------------------------------------------------------------------
_JAX_UNARY = {"exp": "exp", "expm1": "expm1", "log": "log", "log1p": "log1p", "sqrt": "sqrt", "cbrt": "cbrt", "tanh": "tanh",
              "sinh": "sinh", "cosh": "cosh", "sin": "sin", "cos": "cos", "atan": "atan", "abs": "fabs"}
_JAX_BINARY = {"add": add, "sub": sub, "mul": mul, "div": div, "pow": power}
_JAX_COMPARE = {"ge", "gt", "le", "lt"}


def from_jaxpr(closed):
    """Node tuple of a scalar -> scalar ``ClosedJaxpr`` (``jax.make_jaxpr(f)(0.0)``).  Primitives: ``add sub mul div neg
    exp expm1 log log1p pow integer_pow sqrt rsqrt cbrt tanh sinh cosh sin cos atan abs square max min select_n`` with
    ``ge gt le lt`` predicates, ``convert_element_type`` / ``copy`` (identities for a scalar) and nested ``pjit`` /
    ``custom_jvp_call`` bodies.  Only the data model of a jaxpr is used (``jaxpr.invars / constvars / eqns / outvars``,
    ``eqn.primitive.name / invars / outvars / params``, ``Literal.val``, ``closed.consts``): jax itself is not imported."""
    jaxpr = getattr(closed, "jaxpr", closed)
    consts = list(getattr(closed, "consts", ()))
    if len(jaxpr.invars) != 1 or len(jaxpr.outvars) != 1:
        raise TypeError("yield_stress must map one scalar (p) to one scalar")
    return _eval_jaxpr(jaxpr, consts, [VAR])[0]


def _scalar_const(v):
    a = np.asarray(v)
    if a.size != 1:
        raise TypeError("a traced yield_stress(p) can only close over scalars")
    return const(float(a.reshape(())))


def _eval_jaxpr(jaxpr, consts, args):
    env = {}

    def read(v):
        if hasattr(v, "val"):   # jax.core.Literal
            return _scalar_const(v.val)
        return env[v]

    for var, val in zip(jaxpr.constvars, consts):
        env[var] = _scalar_const(val)
    for var, val in zip(jaxpr.invars, args):
        env[var] = val
    for eqn in jaxpr.eqns:
        name = eqn.primitive.name
        ins = [read(v) for v in eqn.invars]
        if name in _JAX_UNARY:
            out = fn(_JAX_UNARY[name], ins[0])
        elif name in _JAX_BINARY:
            out = _JAX_BINARY[name](ins[0], ins[1])
        elif name == "neg":
            out = neg(ins[0])
        elif name == "integer_pow":
            out = power(ins[0], const(eqn.params["y"]))
        elif name == "square":
            out = mul(ins[0], ins[0])
        elif name == "rsqrt":
            out = div(const(1.0), fn("sqrt", ins[0]))
        elif name == "max":
            out = select(ins[0], ins[1], ins[0], ins[1])
        elif name == "min":
            out = select(ins[1], ins[0], ins[0], ins[1])
        elif name in _JAX_COMPARE:
            out = ("cmp", name, ins[0], ins[1])     # consumed by select_n only
        elif name == "select_n":
            pred, on_false, on_true = ins
            if pred[0] != "cmp":
                raise TypeError("select_n on something that is not a comparison of traced values")
            _, op, a, b = pred
            if op == "ge":
                out = select(a, b, on_true, on_false)
            elif op == "le":
                out = select(b, a, on_true, on_false)
            elif op == "lt":     # a < b  ==  not (a >= b)
                out = select(a, b, on_false, on_true)
            else:                # a > b  ==  not (b >= a)
                out = select(b, a, on_false, on_true)
        elif name in ("convert_element_type", "copy", "copy_p", "stop_gradient", "reshape", "squeeze", "broadcast_in_dim"):
            out = ins[0]        # identities for a 0-d value
        elif name in ("pjit", "closed_call", "core_call", "custom_jvp_call", "custom_vjp_call", "remat", "checkpoint"):
            inner = eqn.params.get("jaxpr") or eqn.params.get("call_jaxpr") or eqn.params.get("fun_jaxpr")
            inner_jaxpr = getattr(inner, "jaxpr", inner)
            outs = _eval_jaxpr(inner_jaxpr, list(getattr(inner, "consts", ())), ins)
            for var, val in zip(eqn.outvars, outs):
                env[var] = val
            continue
        else:
            raise TypeError(f"jax primitive '{name}' is not supported inside a traced yield_stress(p); supported: "
                            f"{', '.join(sorted(list(_JAX_UNARY) + list(_JAX_BINARY) + ['neg', 'integer_pow', 'square', 'rsqrt', 'max', 'min', 'select_n']))}")
        env[eqn.outvars[0]] = out
    outs = [read(v) for v in jaxpr.outvars]
    for o in outs:
        if o[0] == "cmp":
            raise TypeError("yield_stress(p) returns a comparison, not a stress")
    return outs


def trace_jax(func):
    """``(R_node, dR_node)`` of a callable written with ``jax.numpy``; needs jax."""
    import jax

    prev = jax.config.read("jax_enable_x64") if hasattr(jax.config, "read") else None
    try:
        jax.config.update("jax_enable_x64", True)   # closed-over constants and literals in double precision
        closed = jax.make_jaxpr(func)(0.0)
    finally:
        if prev is not None:
            jax.config.update("jax_enable_x64", prev)
    R = from_jaxpr(closed)
    return R, diff(R)


def _jax_available():
    import importlib.util

    return importlib.util.find_spec("jax") is not None


class TracedLaw:
    """What the tracer produces: C expressions plus host evaluators (used by tests and for sig0)."""

    def __init__(self, func, nodes=None):
        self.func = func
        if nodes is not None:
            self.R_node, self.dR_node = nodes
            self.how = "given"
        else:
            try:
                self.R_node, self.dR_node = trace(func)
                self.how = "numpy"
            except TypeError:
                # not written with numpy ufuncs and operators: a jax.numpy callable (the reference's usual form) is
                # traced by jax itself where jax exists
                if not _jax_available():
                    raise
                self.R_node, self.dR_node = trace_jax(func)
                self.how = "jax"
        self.expr_R, self.expr_dR = emit_c(self.R_node), emit_c(self.dR_node)
        self.sig0 = float(evaluate(self.R_node, 0.0))
        # R(0) = 0 is a legitimate law (a power law rising from zero): the kernels floor their Newton tolerance at
        # 2e-8 mu for it, and materials.CustomHardening accepts it
        if not (self.sig0 >= 0.0) or not math.isfinite(self.sig0):
            raise ValueError(f"yield_stress(0) = {self.sig0}: the initial yield stress must be finite and non-negative")
        # the trace must reproduce the callable itself (guards against value-dependent Python control flow)
        pts = np.array([0.0, 1e-4, 1e-3, 1e-2, 1e-1])
        try:
            direct = np.array([float(func(float(x))) for x in pts])
        except Exception:
            direct = None
        if direct is not None:
            mine = evaluate(self.R_node, pts)
            # (a jax callable evaluates in single precision unless jax_enable_x64 is set)
            if not np.allclose(mine, direct, rtol=1e-12 if self.how != "jax" else 1e-5, atol=0.0):
                raise ValueError("the traced expression does not reproduce yield_stress(p) on sample points "
                                 "(value-dependent Python control flow cannot be traced)")
        slope = evaluate(self.dR_node, pts)
        if np.any(~np.isfinite(slope)) or np.any(slope < 0.0):
            import warnings

            warnings.warn("yield_stress(p) is softening (dR/dp < 0) on sample points: the local Newton assumes a "
                          "non-decreasing law", RuntimeWarning)

    def R(self, p):
        return evaluate(self.R_node, np.asarray(p, dtype=np.float64))

    def dR(self, p):
        return evaluate(self.dR_node, np.asarray(p, dtype=np.float64))
