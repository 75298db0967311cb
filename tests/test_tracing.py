"""A Python ``yield_stress(p)`` callable -- what the reference's only test on this path hands to
``jm.FeFpJ2Plasticity`` (``tests/test_FeFp_jax.py:14-19``) -- is traced into C expressions for R and
dR/dp and compiled into the fused kernels."""
import math

import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd import _lib, tracing
from oracle import constitutive_np as onp

from helpers import E, NU, SIG0_V, SIGU_V, B_V, SIG0_F, SIGU_F, B_F, j2_history, fefp_path


def reference_yield_stress():
    """The function of tests/test_FeFp_jax.py:7-15, verbatim apart from jnp -> np."""
    sig0 = 500.0

    b = 1000
    sigu = 750.0

    def yield_stress(p):
        return sig0 + (sigu - sig0) * (1 - np.exp(-b * p))

    return yield_stress


def test_reference_voce_callable_traces_to_the_hand_written_expressions():
    law = tracing.TracedLaw(reference_yield_stress())
    assert law.expr_R == "(500.0 + DXM_MUL(250.0, (1.0 - exp(DXM_MUL((-1000.0), p)))))"
    assert law.expr_dR == "DXM_MUL(250000.0, exp(DXM_MUL((-1000.0), p)))"   # one folded coefficient, like (sigu - sig0) * b * exp(-b p)
    assert law.sig0 == 500.0
    p = np.linspace(0.0, 0.02, 11)
    v = onp.VoceHardening(500.0, 750.0, 1000.0)
    assert np.array_equal(law.R(p), v.R(p)) and np.allclose(law.dR(p), v.dR(p), rtol=1e-15)


LAWS = [
    lambda p: 250.0 + 600.0 * ((p + 1e-3) ** 0.3 - 1e-3**0.3),
    lambda p: 300 * np.sqrt(1 + p / 0.01) + 20 * np.tanh(5 * p) / np.float64(2),
    lambda p: 100.0 + 50.0 * np.log1p(40.0 * p) + 3.0 * np.expm1(0.5 * p) - 2.0 * np.cos(p) + np.arctan(p) / 3,
    lambda p: 200.0 + np.maximum(1e3 * p, 2e3 * p - 5.0) + abs(p) * 10.0 + np.minimum(p, 0.5) ** 2,
    lambda p: (400.0 + 1e3 * p) / (1.0 + 0.1 * p) + 2.0 ** (p * 3.0) + np.cbrt(1.0 + p) + np.square(p) + np.sinh(p) - np.cosh(p),
    lambda p: 150.0 + 1.0 / (0.01 + p) * (-1.0) + 100.0 + (p + 0.3) ** (p + 1.0) - np.negative(p) + np.reciprocal(2.0 + p) + np.sin(p),
]


@pytest.mark.parametrize("k", range(len(LAWS)))
def test_traced_value_and_symbolic_derivative(k):
    f = LAWS[k]
    law = tracing.TracedLaw(f)
    p = np.array([1e-4, 3e-3, 0.02, 0.11, 0.7])   # away from the kinks of max / min / abs
    assert np.allclose(law.R(p), [float(f(float(x))) for x in p], rtol=1e-14)
    h = 1e-6
    fd = (law.R(p + h) - law.R(p - h)) / (2 * h)
    assert np.allclose(law.dR(p), fd, rtol=2e-7, atol=1e-7)
    # the emitted C is an expression in p only: evaluate it with Python's math as a cross-check of emit_c
    env = {k_: getattr(math, k_) for k_ in ("exp", "expm1", "log", "log1p", "sqrt", "tanh", "sinh", "cosh", "sin", "cos", "atan", "fabs", "pow")}
    env["cbrt"] = np.cbrt
    env["_mul"] = lambda a_, b_: a_ * b_

    def c_eval(expr, x):
        import re

        py = re.sub(r"\(\(([^?]*?)\) \? ([^:]*?) : ([^)]*?)\)", r"((\2) if (\1) else (\3))", expr)   # not needed for the nested-free cases below
        py = re.sub(r"DXM_MUL\(", "_mul(", py)
        py = re.sub(r"(0x[0-9a-f.]+p[+-]?\d+)", lambda m_: repr(float.fromhex(m_.group(1))), py)
        return eval(py, {"__builtins__": {}}, dict(env, p=x))

    if "?" not in law.expr_R and "?" not in law.expr_dR:
        for x in p:
            assert math.isclose(c_eval(law.expr_R, float(x)), float(law.R(x)), rel_tol=1e-13)
            assert math.isclose(c_eval(law.expr_dR, float(x)), float(law.dR(x)), rel_tol=1e-13, abs_tol=1e-300)


def test_untraceable_callables_are_rejected_with_a_reason():
    with pytest.raises(TypeError, match="not traceable"):
        tracing.TracedLaw(lambda p: 250.0 + math.exp(p))
    with pytest.raises(TypeError, match="not traceable"):
        tracing.TracedLaw(lambda p: 250.0 if p > 0 else 200.0)
    with pytest.raises(TypeError, match="not supported"):
        tracing.TracedLaw(lambda p: 250.0 + np.arcsinh(p))
    with pytest.raises(ValueError, match="yield_stress\\(0\\)"):
        tracing.TracedLaw(lambda p: -5.0 + p)
    with pytest.warns(RuntimeWarning, match="softening"):
        tracing.TracedLaw(lambda p: 250.0 - 10.0 * p)


def test_traced_law_compiles_for_gfx950_without_a_gpu():
    beh = jm.FeFpJ2Plasticity(elasticity=jm.LinearElasticIsotropic(E=E, nu=NU), yield_stress=reference_yield_stress())
    assert beh.law == _lib.LAW_FEFP_J2_VOCE and beh.custom_hardening is not None
    assert beh.params() == [E, NU, 500.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    lib = _lib.load_custom(beh.custom_hardening.expr_R, beh.custom_hardening.expr_dR)
    assert lib.dxm_has_custom_hardening() == 1
    beh2 = jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), LAWS[3])   # ternaries and pow compile too
    assert _lib.load_custom(beh2.custom_hardening.expr_R, beh2.custom_hardening.expr_dR).dxm_has_custom_hardening() == 1


@pytest.mark.gpu
def test_traced_voce_callable_reproduces_the_builtin_voce_kernels_bit_for_bit():
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    el = jm.LinearElasticIsotropic(E=E, nu=NU)

    def ys(p):
        return SIG0_V + (SIGU_V - SIG0_V) * (1 - np.exp(-B_V * p))

    n = 5000
    a = JAXMaterial(jm.vonMisesIsotropicHardening(el, ys))
    b = JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V)))
    a.set_data_manager(n)
    b.set_data_manager(n)
    for eps in j2_history(n, sig0=SIG0_V):
        ra, rb = a.integrate(eps), b.integrate(eps)
        assert a.last_stats == b.last_stats and b.last_stats["n_not_converged"] == 0
        for x, y in zip(ra, rb):
            assert np.array_equal(np.asarray(x), np.asarray(y))
        a.data_manager.update()
        b.data_manager.update()
    assert b.last_stats["n_plastic"] == 0 and a.get_final_state_dict()["p"].max() > 1e-3

    def ysf(p):
        return SIG0_F + (SIGU_F - SIG0_F) * (1 - np.exp(-B_F * p))

    n = 700
    a = JAXMaterial(jm.FeFpJ2Plasticity(elasticity=el, yield_stress=ysf))
    b = JAXMaterial(jm.FeFpJ2Plasticity(elasticity=el, yield_stress=jm.VoceHardening(SIG0_F, SIGU_F, B_F)))
    a.set_data_manager(n)
    b.set_data_manager(n)
    for F in fefp_path(n)[2::4]:
        ra, rb = a.integrate(F), b.integrate(F)
        assert a.last_stats == b.last_stats
        for x, y in zip(ra, rb):
            assert np.array_equal(np.asarray(x), np.asarray(y))
        a.data_manager.update()
        b.data_manager.update()
    assert b.last_stats["n_plastic"] == n


@pytest.mark.gpu
def test_reference_fefp_test_runs_with_only_the_imports_changed():
    """tests/test_FeFp_jax.py:6-33 with `dolfinx_materials.jaxmat` -> `dolfinx_materials_amd.jaxmat`,
    `jaxmat.materials` -> `dolfinx_materials_amd.materials` and jnp -> np; every step is checked against the
    oracle (the reference test itself asserts nothing)."""
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    Nbatch = 10
    behavior = jm.FeFpJ2Plasticity(elasticity=jm.LinearElasticIsotropic(E=70e3, nu=0.3), yield_stress=reference_yield_stress())
    material = JAXMaterial(behavior)
    material.set_data_manager(Nbatch)
    eps = 2e-2
    Nsteps = 20
    dt = 0
    st = onp.fefp_initial_state(Nbatch)
    cp, p = st["cpinv"], st["p"]
    hard = onp.VoceHardening(500.0, 750.0, 1000.0)
    for t in np.linspace(0, 1.0, Nsteps)[1:]:
        F = np.zeros((Nbatch, 9))
        F[:, 0] = 1 + eps * t
        F[:, [1, 2]] = 1 - eps / 2 * t
        P, isv, Ct = material.integrate(F, dt)
        ref = onp.fefp_update(F, cp, p, 70e3, 0.3, hard)
        assert np.abs(np.asarray(P) - ref["P"]).max() < 1e-11 * np.abs(ref["P"]).max()
        assert np.abs(np.asarray(Ct) - ref["Ct"]).max() < 1e-11 * np.abs(ref["Ct"]).max()
        assert np.abs(np.asarray(isv)[:, 0] - ref["p"]).max() < 1e-13
        material.data_manager.update()
        cp, p = ref["cpinv"], ref["p"]
    assert p.min() > 1e-2 and np.asarray(P).shape == (Nbatch, 9) and np.asarray(Ct).shape == (Nbatch, 9, 9)
