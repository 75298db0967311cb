"""GPU parity of the finite-strain FeFp J2 kernel against the CPU oracle (parity with jaxmat is
unpinned: tests/test_FeFp_jax.py has no assertions; see oracle/__init__.py)."""
import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd.jaxmat import JAXMaterial
from oracle import constitutive_np as onp

from helpers import E, NU, SIG0_F, SIGU_F, B_F, fefp_path

pytestmark = pytest.mark.gpu
TIGHT = 1e-11
HARD = onp.VoceHardening(SIG0_F, SIGU_F, B_F)


def make(n):
    m = JAXMaterial(jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(SIG0_F, SIGU_F, B_F)))
    m.set_data_manager(n)
    return m


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_identity_gradient_gives_zero_stress_and_elastic_tangent():
    n = 130
    m = make(n)
    F = np.zeros((n, 9))
    F[:, :3] = 1.0
    P, isv, Ct = m.integrate(F)
    assert np.abs(P).max() == 0.0 and not np.isnan(Ct).any()
    ref = onp.fefp_update(F, onp.fefp_initial_state(n)["cpinv"], np.zeros(n), E, NU, HARD)
    assert relerr(Ct, ref["Ct"]) < TIGHT
    assert np.array_equal(isv[:, 0], np.zeros(n)) and np.allclose(isv[:, 1:4], 1.0) and np.allclose(isv[:, 4:], 0.0)
    assert m.last_stats["n_plastic"] == 0 and m.last_stats["n_nan"] == 0


@pytest.mark.parametrize("n", [10, 63, 64, 65, 1000])
def test_reference_smoke_path_matches_oracle(n):
    """The driver sequence of tests/test_FeFp_jax.py:21-33 (set_data_manager, 19 x integrate +
    data_manager.update), with perturbed copies of the path for the other points."""
    m = make(n)
    st = onp.fefp_initial_state(n)
    cp, p = st["cpinv"], st["p"]
    saw_plastic = False
    for F in fefp_path(n):
        P, isv, Ct = m.integrate(F, 0)
        ref = onp.fefp_update(F, cp, p, E, NU, HARD)
        safe = np.abs(ref["f_trial"]) > 1e-9 * SIG0_F
        assert P.shape == (n, 9) and isv.shape == (n, 7) and Ct.shape == (n, 9, 9)
        assert relerr(P[safe], ref["P"][safe]) < TIGHT
        assert relerr(Ct[safe], ref["Ct"][safe]) < TIGHT
        assert np.abs(isv[safe, 0] - ref["p"][safe]).max() < 1e-16 + TIGHT * max(ref["p"].max(), 1e-300)
        assert relerr(isv[safe, 1:], ref["be_bar"][safe]) < TIGHT
        assert m.last_stats["n_nan"] == 0 and m.last_stats["n_not_converged"] == 0
        saw_plastic |= bool(ref["plastic"].any())
        m.data_manager.update()
        cp, p = ref["cpinv"], ref["p"]
    assert saw_plastic
    be = onp.mandel_to_tensor(m.get_final_state_dict()["be_bar"])
    assert np.abs(np.linalg.det(be) - 1.0).max() < 1e-12  # isochoric elastic left Cauchy-Green


def test_set_initial_state_with_prior_plastic_state():
    n = 500
    rng = np.random.default_rng(3)
    Fn = onp.tensor_to_nsym(np.eye(3) + 0.03 * rng.standard_normal((n, 3, 3)))
    r0 = onp.fefp_update(Fn, onp.fefp_initial_state(n)["cpinv"], np.zeros(n), E, NU, HARD)
    assert r0["plastic"].mean() > 0.5
    m = make(n)
    m.set_initial_state_dict({"F": Fn, "be_bar": r0["be_bar"], "p": r0["p"]})
    F = onp.tensor_to_nsym(onp.nsym_to_tensor(Fn) + 0.01 * rng.standard_normal((n, 3, 3)))
    P, isv, Ct = m.integrate(F)
    ref = onp.fefp_update(F, r0["cpinv"], r0["p"], E, NU, HARD)
    safe = np.abs(ref["f_trial"]) > 1e-9 * SIG0_F
    assert relerr(P[safe], ref["P"][safe]) < 1e-10
    assert relerr(Ct[safe], ref["Ct"][safe]) < 1e-10
    assert relerr(isv[safe, 1:], ref["be_bar"][safe]) < 1e-10


def test_objectivity():
    """Superposed rigid rotation: P(QF) = Q P(F), be_bar(QF) = Q be_bar Q^T, same p."""
    n = 256
    rng = np.random.default_rng(9)
    F = np.eye(3) + 0.04 * rng.standard_normal((n, 3, 3))
    A = rng.standard_normal((n, 3, 3))
    Q, _ = np.linalg.qr(A)
    Q *= np.sign(np.linalg.det(Q))[:, None, None]
    m1, m2 = make(n), make(n)
    P1, isv1, _ = m1.integrate(onp.tensor_to_nsym(F))
    P2, isv2, _ = m2.integrate(onp.tensor_to_nsym(Q @ F))
    assert relerr(onp.nsym_to_tensor(P2), Q @ onp.nsym_to_tensor(P1)) < 1e-11
    assert np.abs(isv2[:, 0] - isv1[:, 0]).max() < 1e-14
    be1, be2 = onp.mandel_to_tensor(isv1[:, 1:]), onp.mandel_to_tensor(isv2[:, 1:])
    assert relerr(be2, Q @ be1 @ Q.transpose(0, 2, 1)) < 1e-11


def test_large_deformation_single_step():
    """Very large single increments (|F - I| ~ 0.25, dp up to ~2): the 2x2 local Newton still
    converges everywhere and the kernel follows the oracle."""
    rng = np.random.default_rng(0)
    F = np.eye(3) + 0.25 * rng.standard_normal((4000, 3, 3))
    F = F[np.linalg.det(F) > 0.2]
    n = len(F)
    m = make(n)
    P, isv, Ct = m.integrate(onp.tensor_to_nsym(F))
    ref = onp.fefp_update(onp.tensor_to_nsym(F), onp.fefp_initial_state(n)["cpinv"], np.zeros(n), E, NU, HARD)
    assert m.last_stats["n_not_converged"] == 0 and m.last_stats["n_nan"] == 0 and not ref["notconv"].any()
    safe = np.abs(ref["f_trial"]) > 1e-9 * SIG0_F
    assert relerr(P[safe], ref["P"][safe]) < 1e-10
    assert relerr(Ct[safe], ref["Ct"][safe]) < 1e-9
    assert relerr(isv[safe, 0], ref["p"][safe]) < 1e-10
    be = onp.mandel_to_tensor(isv[:, 1:])
    assert np.abs(np.linalg.det(be) - 1).max() < 1e-12


def test_fefp_with_linear_hardening_matches_oracle():
    """FeFpJ2Plasticity accepts either hardening law (law id 4: R = sig0 + H p)."""
    n = 700
    hard = onp.LinearHardening(400.0, 2e3)
    m = JAXMaterial(jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(400.0, 2e3)))
    m.set_data_manager(n)
    st = onp.fefp_initial_state(n)
    cp, p = st["cpinv"], st["p"]
    for k, F in enumerate(fefp_path(n)):
        if k % 3:
            continue
        P, isv, Ct = m.integrate(F)
        ref = onp.fefp_update(F, cp, p, E, NU, hard)
        safe = np.abs(ref["f_trial"]) > 1e-9 * 400.0
        assert relerr(P[safe], ref["P"][safe]) < TIGHT and relerr(Ct[safe], ref["Ct"][safe]) < TIGHT
        assert np.abs(isv[safe, 0] - ref["p"][safe]).max() < 1e-16 + TIGHT * max(ref["p"].max(), 1e-300)
        assert m.last_stats["n_not_converged"] == 0
        m.data_manager.update()
        cp, p = ref["cpinv"], ref["p"]
    assert ref["plastic"].all()


def test_inverted_points_are_reported_not_computed():
    """det F <= 0: the reference's arithmetic gives NaN (J^(-2/3)) and QuadratureMap.update asserts on it
    (quadrature_map.py:322-324); the kernel must poison exactly those points, count them, and leave the others alone."""
    n = 1000
    rng = np.random.default_rng(11)
    F = np.eye(3)[None] + 0.05 * rng.standard_normal((n, 3, 3))
    bad = np.arange(0, n, 37)
    F[bad, 0, :] *= -1.0                      # reflected: det F < 0
    F[5] = 0.0
    F[5, 0, 0] = F[5, 1, 1] = 1.0             # flat: det F = 0
    bad = np.union1d(bad, [5])
    F9 = onp.tensor_to_nsym(F)
    m = JAXMaterial(jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(SIG0_F, SIGU_F, B_F)))
    m.set_data_manager(n)
    P, _, Ct = m.integrate(F9)                # (QuadratureMap's own `assert not np.isnan` then fires, as upstream)
    assert m.last_stats["n_nan"] == len(bad)
    P, Ct = np.asarray(P), np.asarray(Ct)
    assert np.isnan(Ct[bad]).all() and np.isfinite(Ct[good_ := np.setdiff1d(np.arange(n), bad)]).all()
    good = np.setdiff1d(np.arange(n), bad)
    assert np.isnan(P[bad]).all() and np.isfinite(P[good]).all()
    st = onp.fefp_initial_state(n)
    ref = onp.fefp_update(F9[good], st["cpinv"][good], st["p"][good], E, NU, onp.VoceHardening(SIG0_F, SIGU_F, B_F))
    assert relerr(P[good], ref["P"]) < TIGHT


@pytest.mark.parametrize("kind", ["voce", "linear"])
def test_hip_uniaxial_kirchhoff_stress_paths_follow_the_model_known_answer(kind):
    """The FeFp KERNEL against a material-point known answer that does not go through the oracle
    (``helpers.fefp_uniaxial_known_answer``): 257 points on their own uniaxial Kirchhoff-stress paths, lateral stretch by Newton on
    ``P_yy = 0`` with the kernel's own 9x9 tangent (several ``integrate`` calls from one s0, then ``advance``), 8 increments up to 8 %
    stretch: ``tau_xx = R(p)`` at the plastic points, ``det(be_bar) = 1`` with ``be_bar`` carrying exactly the deviatoric stress,
    ``kappa/2 (J^2 - 1) = tau_xx / 3`` -- at 1e-9, both hardening laws (Voce: tests/test_FeFp_jax.py:14-15)."""
    from helpers import fefp_uniaxial_known_answer

    n = 257
    if kind == "voce":
        hard, R = jm.VoceHardening(SIG0_F, SIGU_F, B_F), (lambda q: SIG0_F + (SIGU_F - SIG0_F) * (1.0 - np.exp(-B_F * q)))
    else:
        hard, R = jm.LinearHardening(SIG0_F, 2e3), (lambda q: SIG0_F + 2e3 * q)
    m = JAXMaterial(jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=E, nu=NU), hard))
    m.set_data_manager(n)

    def step(F9):
        P, isv, Ct = m.integrate(F9)
        assert m.last_stats["n_not_converged"] == 0 and m.last_stats["n_nan"] == 0
        isv = np.asarray(isv)
        return np.array(P), np.array(Ct), isv[:, 0].copy(), isv[:, 1:7].copy()

    nplastic, pmax = fefp_uniaxial_known_answer(step, m.data_manager.update, R, n=n)
    assert nplastic > 200 and pmax > 1e-2
    m.close()
