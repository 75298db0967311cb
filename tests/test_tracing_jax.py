"""``tracing.from_jaxpr``: a ``yield_stress(p)`` written with ``jax.numpy`` -- the form the reference's own test and demos
use (``tests/test_FeFp_jax.py:14-15``, ``plane_elastoplasticity.py:60-66``) -- is traced by jax (``jax.make_jaxpr``) and its
primitives are translated into the expression nodes the numpy tracer produces.  jax is not installed in the build container
nor on the GPU box, so the translator is exercised here through objects with the data model of a jaxpr (the attributes
``from_jaxpr`` reads and nothing else); the last tests run the real thing where jax imports."""
import numpy as np
import pytest

from dolfinx_materials_amd import tracing as T


class Var:
    pass


class Lit:
    def __init__(self, val):
        self.val = val


class _Prim:
    def __init__(self, name):
        self.name = name


class Eqn:
    def __init__(self, prim, invars, outvars, **params):
        self.primitive, self.invars, self.outvars, self.params = _Prim(prim), invars, outvars, params


class Jaxpr:
    def __init__(self, constvars, invars, eqns, outvars):
        self.constvars, self.invars, self.eqns, self.outvars = constvars, invars, eqns, outvars


class Closed:
    def __init__(self, jaxpr, consts=()):
        self.jaxpr, self.consts = jaxpr, list(consts)


def voce_jaxpr(sig0=500.0, sigu=750.0, b=1000.0):
    """What jax makes of ``sig0 + (sigu - sig0) * (1 - jnp.exp(-b * p))``: mul, exp, sub, mul, add with the closed-over
    Python numbers as literals (weakly typed scalars get a convert_element_type in between on some versions)."""
    p, a, a2, e, c, d, out = (Var() for _ in range(7))
    eqns = [Eqn("mul", [Lit(-b), p], [a]), Eqn("convert_element_type", [a], [a2], new_dtype="float64", weak_type=False),
            Eqn("exp", [a2], [e]), Eqn("sub", [Lit(1.0), e], [c]), Eqn("mul", [Lit(sigu - sig0), c], [d]), Eqn("add", [Lit(sig0), d], [out])]
    return Closed(Jaxpr([], [p], eqns, [out]))


def test_voce_law_from_a_jaxpr_equals_the_numpy_trace():
    sig0, sigu, b = 500.0, 750.0, 1000.0

    def yield_stress(p):   # the reference's function with np for jnp
        return sig0 + (sigu - sig0) * (1 - np.exp(-b * p))

    numpy_law = T.TracedLaw(yield_stress)
    R = T.from_jaxpr(voce_jaxpr(sig0, sigu, b))
    assert T.emit_c(R) == numpy_law.expr_R and T.emit_c(T.diff(R)) == numpy_law.expr_dR
    law = T.TracedLaw(yield_stress, nodes=(R, T.diff(R)))
    assert law.sig0 == 500.0 and law.expr_dR == "DXM_MUL(250000.0, exp(DXM_MUL((-1000.0), p)))"


def test_constants_of_the_closed_jaxpr_and_nested_calls():
    # R(p) = c0 + pjit(lambda x: c1 * x**3)(p) with c0 = 200 a captured constant of the outer jaxpr
    p, c0, q, out = Var(), Var(), Var(), Var()
    x, c1, x3, y = Var(), Var(), Var(), Var()
    inner = Closed(Jaxpr([c1], [x], [Eqn("integer_pow", [x], [x3], y=3), Eqn("mul", [c1, x3], [y])], [y]), [np.float64(4.0)])
    outer = Closed(Jaxpr([c0], [p], [Eqn("pjit", [p], [q], jaxpr=inner, name="f"), Eqn("add", [c0, q], [out])], [out]), [np.array(200.0)])
    R = T.from_jaxpr(outer)
    pts = np.array([0.0, 0.1, 0.5])
    assert np.allclose(T.evaluate(R, pts), 200.0 + 4.0 * pts**3) and np.allclose(T.evaluate(T.diff(R), pts), 12.0 * pts**2)


@pytest.mark.parametrize("op,expect", [("gt", lambda p: np.where(p > 0.01, 300.0 + 1e3 * p, 310.0)), ("ge", lambda p: np.where(p >= 0.01, 300.0 + 1e3 * p, 310.0)),
                                       ("lt", lambda p: np.where(p < 0.01, 300.0 + 1e3 * p, 310.0)), ("le", lambda p: np.where(p <= 0.01, 300.0 + 1e3 * p, 310.0))])
def test_where_through_comparison_and_select_n(op, expect):
    # jnp.where(p OP 0.01, 300 + 1e3 p, 310): select_n(pred, on_false, on_true)
    p, pred, lin, lin2, out = (Var() for _ in range(5))
    eqns = [Eqn(op, [p, Lit(0.01)], [pred]), Eqn("mul", [Lit(1e3), p], [lin]), Eqn("add", [Lit(300.0), lin], [lin2]),
            Eqn("select_n", [pred, Lit(310.0), lin2], [out])]
    R = T.from_jaxpr(Closed(Jaxpr([], [p], eqns, [out])))
    pts = np.array([0.0, 0.005, 0.01, 0.02])
    assert np.array_equal(T.evaluate(R, pts), expect(pts))
    assert "?" in T.emit_c(R)


def test_max_min_sqrt_tanh_and_pow():
    p, a, b, c, d, e, out = (Var() for _ in range(7))
    eqns = [Eqn("max", [p, Lit(1e-3)], [a]), Eqn("sqrt", [a], [b]), Eqn("tanh", [b], [c]), Eqn("pow", [a, Lit(0.3)], [d]),
            Eqn("min", [d, Lit(0.9)], [e]), Eqn("add", [c, e], [out])]
    R = T.from_jaxpr(Closed(Jaxpr([], [p], eqns, [out])))
    pts = np.array([0.0, 1e-4, 0.02, 0.5, 2.0])
    m = np.maximum(pts, 1e-3)
    assert np.allclose(T.evaluate(R, pts), np.tanh(np.sqrt(m)) + np.minimum(m**0.3, 0.9), rtol=1e-15)


def test_what_cannot_be_translated_says_so():
    p, out = Var(), Var()
    with pytest.raises(TypeError, match="erf_inv"):
        T.from_jaxpr(Closed(Jaxpr([], [p], [Eqn("erf_inv", [p], [out])], [out])))
    with pytest.raises(TypeError, match="one scalar"):
        T.from_jaxpr(Closed(Jaxpr([], [p, Var()], [], [p])))
    v = Var()
    with pytest.raises(TypeError, match="scalars"):
        T.from_jaxpr(Closed(Jaxpr([v], [p], [Eqn("add", [p, v], [out])], [out]), [np.ones(3)]))


def test_a_law_rising_from_zero_is_accepted():
    """the kernels floor their Newton tolerance for R(0) = 0 and CustomHardening accepts it; so does the tracer."""
    law = T.TracedLaw(lambda p: 800.0 * p**0.5 + 0.0)
    assert law.sig0 == 0.0
    with pytest.raises(ValueError):
        T.TracedLaw(lambda p: -1.0 + p)


# ---- with jax itself -------------------------------------------------------------------------------------------------
def test_the_references_yield_stress_written_with_jnp_is_traced_verbatim():
    jax = pytest.importorskip("jax")
    import jax.numpy as jnp

    import dolfinx_materials_amd.materials as jm

    sig0 = 500.0
    sigu = 750.0
    b = 1000.0

    def yield_stress(p):   # tests/test_FeFp_jax.py:14-15, verbatim
        return sig0 + (sigu - sig0) * (1 - jnp.exp(-b * p))

    def yield_stress_np(p):
        return sig0 + (sigu - sig0) * (1 - np.exp(-b * p))

    law, ref = T.TracedLaw(yield_stress), T.TracedLaw(yield_stress_np)
    assert law.how == "jax" and law.expr_R == ref.expr_R and law.expr_dR == ref.expr_dR
    beh = jm.FeFpJ2Plasticity(elasticity=jm.LinearElasticIsotropic(E=70e3, nu=0.3), yield_stress=yield_stress)
    assert beh.custom_hardening.expr_R == "(" + ref.expr_R + ")"
    del jax
