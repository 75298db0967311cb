"""CPU tests that pin the oracle (numpy + plain C) against the reference's golden vectors, the
in-tree closed-form spec and AD / finite-difference tangents.  No GPU, no product code."""
import os

import numpy as np
import pytest

from oracle import constitutive_np as onp
from oracle import oracle_c
from oracle.ref_import import import_reference, reference_available

from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, j2_history, random_j2_state

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
HARDS = {
    "linear": (onp.LinearHardening(SIG0_LIN, H_LIN), (0, SIG0_LIN, H_LIN, 0.0)),
    "voce": (onp.VoceHardening(SIG0_V, SIGU_V, B_V), (1, SIG0_V, SIGU_V, B_V)),
}


def test_elastic_oracle_matches_reference_golden():
    g = np.load(os.path.join(GOLDEN, "elastic_ref.npz"))
    for fn in (onp.elastic_iso, oracle_c.elastic_iso):
        sig, Ct = fn(g["eps"], float(g["E"]), float(g["nu"]))
        assert np.allclose(sig, g["sig"], rtol=1e-10, atol=0)
        assert np.allclose(Ct, g["Ct"], rtol=1e-10, atol=1e-9)


@pytest.mark.skipif(not reference_available(), reason="reference tree only exists in the build container")
def test_elastic_oracle_matches_live_reference():
    import warnings

    _, pm = import_reference()
    rng = np.random.default_rng(5)
    eps = 1e-3 * rng.standard_normal((40, 6))
    mat = pm.LinearElasticIsotropic(210e3, 0.25)
    mat.set_data_manager(40)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sig, isv, Ct = mat.integrate(eps)
    so, Co = onp.elastic_iso(eps, 210e3, 0.25)
    assert np.allclose(so, sig, rtol=1e-10, atol=0) and np.allclose(Co, Ct, rtol=1e-10, atol=1e-9)
    assert isv.shape == (40, 0) and Ct.shape == (40, 6, 6)


def test_j2_coefficient_form_equals_mfront_literal_form():
    """j2_update (eps_p state, Ct = c1 1x1 + c2 I + c3 nxn) == literal transcription of
    IsotropicLinearHardeningPlasticity.mfront:49-77 (eel state, IxI/Id/M tensors)."""
    n = 400
    eps = j2_history(n, seed=3)[2]
    epsp_n, p_n = random_j2_state(n)
    r = onp.j2_update(eps, epsp_n, p_n, E, NU, HARDS["linear"][0])
    sig, eel, p, Dt = onp.j2_update_mfront_form(eps, -epsp_n, p_n, E, NU, H_LIN, SIG0_LIN)
    assert 0.3 < r["plastic"].mean() < 0.95
    assert np.allclose(r["sig"], sig, rtol=1e-13, atol=1e-10)
    assert np.allclose(r["Ct"], Dt, rtol=1e-13, atol=1e-9)
    assert np.allclose(r["p"], p, rtol=1e-13, atol=1e-18)
    assert np.allclose(r["epsp"], eps - eel, rtol=1e-12, atol=1e-18)


@pytest.mark.parametrize("kind", ["linear", "voce"])
def test_tangent_matches_central_differences(kind):
    hard = HARDS[kind][0]
    n = 300
    eps = j2_history(n, seed=11, sig0=hard.sig0)[2]
    epsp_n, p_n = random_j2_state(n, sig0=hard.sig0)
    r = onp.j2_update(eps, epsp_n, p_n, E, NU, hard)
    h = 1e-8
    fd = np.zeros((n, 6, 6))
    for j in range(6):
        d = np.zeros(6)
        d[j] = h
        fd[:, :, j] = (onp.j2_update(eps + d, epsp_n, p_n, E, NU, hard)["sig"] - onp.j2_update(eps - d, epsp_n, p_n, E, NU, hard)["sig"]) / (2 * h)
    safe = np.abs(r["f_trial"]) > 1e-4 * hard.sig0  # FD straddles the kink otherwise
    assert np.abs(fd[safe] - r["Ct"][safe]).max() < 2e-8 * np.abs(r["Ct"]).max()
    assert np.abs(r["Ct"] - r["Ct"].transpose(0, 2, 1)).max() < 1e-9  # associative J2: symmetric


@pytest.mark.parametrize("kind", ["linear", "voce"])
def test_tangent_matches_forward_mode_ad(kind):
    """Same construction as the reference: vmap(jacfwd(constitutive_update, has_aux=True))
    (jaxmat.py:147-151), with torch.func on a per-point restatement whose local Newton is
    differentiated through (fixed iteration count), i.e. the derivative of the algorithm."""
    torch = pytest.importorskip("torch")
    from torch.func import jacfwd, vmap

    hard, (k, s0, h1, h2) = HARDS[kind]
    lmbda, mu = onp.lame(E, NU)
    one = torch.tensor([1.0, 1, 1, 0, 0, 0], dtype=torch.float64)

    def R(p):
        return s0 + h1 * p if k == 0 else s0 + (h1 - s0) * (1 - torch.exp(-h2 * p))

    def dR(p):
        return torch.as_tensor(h1, dtype=torch.float64) if k == 0 else (h1 - s0) * h2 * torch.exp(-h2 * p)

    def update(eps, epsp_n, p_n):
        eel = eps - epsp_n
        tr = eel[:3].sum()
        se = 2 * mu * (eel - tr / 3 * one)
        seq = torch.sqrt(1.5 * (se * se).sum())
        dp = torch.zeros((), dtype=torch.float64)
        for _ in range(12):
            r = seq - 3 * mu * dp - R(p_n + dp)
            dp = dp - r / (-3 * mu - dR(p_n + dp))
        plastic = (seq - R(p_n)) > 0
        dp = torch.where(plastic, dp, torch.zeros_like(dp))
        nrm = 1.5 * se / torch.where(plastic, seq, torch.ones_like(seq))
        eel = eel - dp * nrm
        sig = lmbda * eel[:3].sum() * one + 2 * mu * eel
        return sig, (epsp_n + dp * nrm, p_n + dp)

    n = 64
    eps = j2_history(n, seed=21, sig0=hard.sig0)[2]
    epsp_n, p_n = random_j2_state(n, sig0=hard.sig0)
    Ct_ad, (epsp_ad, p_ad) = vmap(jacfwd(update, argnums=0, has_aux=True))(
        torch.from_numpy(eps), torch.from_numpy(epsp_n), torch.from_numpy(p_n)
    )
    r = onp.j2_update(eps, epsp_n, p_n, E, NU, hard)
    safe = np.abs(r["f_trial"]) > 1e-6 * hard.sig0
    assert np.abs(Ct_ad.numpy()[safe] - r["Ct"][safe]).max() < 1e-9 * np.abs(r["Ct"]).max()
    assert np.allclose(p_ad.numpy()[safe], r["p"][safe], rtol=1e-10, atol=1e-16)
    assert np.allclose(epsp_ad.numpy()[safe], r["epsp"][safe], rtol=1e-9, atol=1e-15)


@pytest.mark.parametrize("kind", ["linear", "voce"])
def test_return_lands_on_yield_surface_and_unloading_is_elastic(kind):
    hard = HARDS[kind][0]
    n = 500
    h = j2_history(n, seed=8, sig0=hard.sig0)
    _, mu = onp.lame(E, NU)
    r = onp.j2_update(h[2], np.zeros((n, 6)), np.zeros(n), E, NU, hard)
    s = r["sig"].copy()
    s[:, :3] -= s[:, :3].mean(axis=1)[:, None]
    seq = np.sqrt(1.5 * (s * s).sum(axis=1))
    pl = r["plastic"]
    assert np.allclose(seq[pl], hard.R(r["p"][pl]), rtol=1e-12)          # f(sigma, p) = 0
    assert (seq[~pl] <= hard.R(r["p"][~pl]) + 1e-9).all()
    assert np.allclose(r["epsp"][:, :3].sum(axis=1), 0, atol=1e-16)       # isochoric flow
    assert np.allclose(np.sqrt(2 / 3 * (r["epsp"] ** 2).sum(axis=1)), r["p"], rtol=1e-12, atol=1e-18)
    r2 = onp.j2_update(h[3], r["epsp"], r["p"], E, NU, hard)             # unloading
    assert not r2["plastic"].any()
    C = onp.elastic_matrix(E, NU)
    assert np.allclose(r2["sig"] - r["sig"], (h[3] - h[2]) @ C.T, atol=1e-9 * np.abs(r["sig"]).max())
    assert np.allclose(r2["Ct"], C[None], atol=1e-9)


@pytest.mark.parametrize("kind", ["linear", "voce"])
def test_c_oracle_equals_numpy_oracle(kind):
    hard, (k, s0, h1, h2) = HARDS[kind]
    n = 3001
    eps = j2_history(n, seed=5, sig0=hard.sig0)[2]
    epsp_n, p_n = random_j2_state(n, sig0=hard.sig0)
    r = onp.j2_update(eps, epsp_n, p_n, E, NU, hard)
    for nt in (1, 3):
        c = oracle_c.j2(eps, epsp_n, p_n, E, NU, k, s0, h1, h2, nthreads=nt)
        safe = np.abs(r["f_trial"]) > 1e-9 * hard.sig0
        for key in ("sig", "epsp", "p", "Ct"):
            assert np.abs(c[key][safe] - r[key][safe]).max() <= 1e-13 * max(np.abs(r[key]).max(), 1e-300), key
        assert c["n_not_converged"] == 0 and abs(c["n_plastic"] - r["plastic"].sum()) <= (~safe).sum()


def test_protocol_golden_from_reference_plumbing():
    """The (eps_p, p)-state oracle replays the sequence that was recorded through the reference's
    own Material/DataManager machinery (tests/golden/make_golden.py::make_protocol)."""
    g = np.load(os.path.join(GOLDEN, "protocol_ref.npz"))
    hard = onp.LinearHardening(250.0, 5e3)
    n = g["eps_hat"].shape[0]
    s0 = dict(stress=np.zeros((n, 6)), p=np.zeros(n), epsp=np.zeros((n, 6)))
    s1 = {k: v.copy() for k, v in s0.items()}
    for k, (op, sc) in enumerate(zip(g["script"], g["scale"])):
        if op == "integrate":
            r = onp.j2_update(sc * g["eps_hat"], s0["epsp"], s0["p"], 70e3, 0.3, hard)
            s1 = dict(stress=r["sig"], p=r["p"], epsp=r["epsp"])
            assert np.allclose(r["sig"], g[f"flux_{k}"], rtol=1e-12, atol=1e-9)
            assert np.allclose(np.hstack([r["p"][:, None], r["epsp"]]), g[f"isv_{k}"], rtol=1e-12, atol=1e-18)
            assert np.allclose(r["Ct"], g[f"Ct_{k}"], rtol=1e-12, atol=1e-8)
        elif op == "update":
            s0 = {k2: v.copy() for k2, v in s1.items()}
        else:
            s1 = {k2: v.copy() for k2, v in s0.items()}
        for key in ("stress", "p", "epsp"):
            assert np.allclose(s0[key].reshape(n, -1), g[f"s0_{key}_{k}"], rtol=1e-12, atol=1e-9 if key == "stress" else 1e-18)
            assert np.allclose(s1[key].reshape(n, -1), g[f"s1_{key}_{k}"], rtol=1e-12, atol=1e-9 if key == "stress" else 1e-18)


def test_uniaxial_known_answer():
    """tests/mfront/test_elastoplasticity.py:31-36: final sigma[:3] = 2/sqrt(3) [sig0, 0, sig0/2]."""
    g = np.load(os.path.join(GOLDEN, "j2_uniaxial_kat.npz"))
    hard = onp.LinearHardening(float(g["sig0"]), float(g["H"]))
    epsp, p = np.zeros((1, 6)), np.zeros(1)
    for eps, sig in zip(g["strain"][1:], g["stress"][1:]):
        r = onp.j2_update(eps[None], epsp, p, float(g["E"]), float(g["nu"]), hard)
        assert np.allclose(r["sig"][0], sig, rtol=1e-12, atol=1e-9)
        epsp, p = r["epsp"], r["p"]
    assert np.allclose(r["sig"][0, :3], g["expected"], rtol=1e-2, atol=1e-8)


@pytest.mark.parametrize("kind", ["voce", "linear"])
def test_uniaxial_stress_path_follows_the_closed_form_of_the_hardening_law(kind):
    """Known answer anchored on the reference's own law formula (`sig0 + (sigu - sig0) (1 - exp(-b p))`, tests/test_FeFp_jax.py:14-15,
    plane_elastoplasticity.py:60-71): under uniaxial stress the flow direction is constant, so the backward-Euler radial return is
    exact for any step size and  sigma_xx = R(p),  eps_xx = sigma_xx / E + p,  eps_yy = eps_zz = -nu sigma_xx / E - p / 2.
    The lateral strain is found by Newton on sigma_yy = 0 with the oracle's own consistent tangent (which is thereby exercised);
    numpy and C oracle."""
    from scipy.optimize import brentq

    hard, (k, s0, h1, h2) = HARDS[kind]
    R = (lambda q: s0 + (h1 - s0) * (1.0 - np.exp(-h2 * q))) if kind == "voce" else (lambda q: s0 + h1 * q)
    epsp, p = np.zeros((1, 6)), np.zeros(1)
    et = 0.0
    for exx in np.linspace(0.0, 3e-2, 13)[1:]:
        for _ in range(30):     # lateral strain: sigma_yy(e_t) = 0, d sigma_yy / d e_t = Ct[1,1] + Ct[1,2]
            eps = np.array([[exx, et, et, 0.0, 0.0, 0.0]])
            r = onp.j2_update(eps, epsp, p, E, NU, hard)
            syy = r["sig"][0, 1]
            if abs(syy) < 1e-11 * s0:
                break
            et -= syy / (r["Ct"][0, 1, 1] + r["Ct"][0, 1, 2])
        assert abs(syy) < 1e-11 * s0
        pc = 0.0 if exx <= s0 / E else brentq(lambda q: R(q) / E + q - exx, 0.0, exx, xtol=1e-16, rtol=1e-15)
        sc = E * exx if pc == 0.0 else R(pc)
        assert abs(r["sig"][0, 0] - sc) < 1e-10 * sc and np.abs(r["sig"][0, 1:]).max() < 1e-10 * s0
        assert abs(r["p"][0] - pc) < 1e-13 + 1e-10 * pc and abs(et - (-NU * sc / E - pc / 2)) < 1e-13
        rc = oracle_c.j2(eps, epsp, p, E, NU, k, s0, h1, h2)
        assert np.allclose(rc["sig"], r["sig"], rtol=0, atol=1e-10 * s0) and np.allclose(rc["p"], r["p"], rtol=0, atol=1e-14)
        epsp, p = r["epsp"], r["p"]
    assert pc > 1e-2      # well into the saturating part of the Voce curve


# ---------------------------------------------------------------------------------------------
# FeFp (parity unpinned; the oracle is pinned by construction checks only)
# ---------------------------------------------------------------------------------------------
from helpers import SIG0_F, SIGU_F, B_F, fefp_path  # noqa: E402

HARD_F = onp.VoceHardening(SIG0_F, SIGU_F, B_F)


def test_fefp_tangent_matches_central_differences_and_be_bar_is_isochoric():
    n = 120
    st = onp.fefp_initial_state(n)
    cp, p = st["cpinv"], st["p"]
    path = fefp_path(n)
    for k, F in enumerate(path):
        r = onp.fefp_update(F, cp, p, E, NU, HARD_F, tangent=(k in (0, 6, 18)))
        if "Ct" in r:
            h = 1e-7
            fd = np.zeros((n, 9, 9))
            for j in range(9):
                e = np.zeros(9)
                e[j] = h
                fd[:, :, j] = (onp.fefp_update(F + e, cp, p, E, NU, HARD_F, tangent=False)["P"] - onp.fefp_update(F - e, cp, p, E, NU, HARD_F, tangent=False)["P"]) / (2 * h)
            safe = np.abs(r["f_trial"]) > 1e-3 * SIG0_F
            assert np.abs(fd[safe] - r["Ct"][safe]).max() < 2e-8 * np.abs(r["Ct"]).max()
        assert not r["notconv"].any() and r["iters"].max() <= 8
        be = onp.mandel_to_tensor(r["be_bar"])
        assert np.abs(onp._det3(be) - 1).max() < 1e-13
        cp, p = r["cpinv"], r["p"]
    assert r["plastic"].all()
    # uniaxial points of tests/test_FeFp_jax.py: saturated Voce stress, Kirchhoff tau_eq = R(p)
    tau = onp.nsym_to_tensor(r["P"][:10]) @ onp.nsym_to_tensor(path[-1][:10]).transpose(0, 2, 1)
    s = tau - np.trace(tau, axis1=1, axis2=2)[:, None, None] / 3 * np.eye(3)
    assert np.allclose(np.sqrt(1.5 * (s * s).sum((1, 2))), HARD_F.R(r["p"][:10]), rtol=1e-11)


def test_fefp_small_strain_limit_is_j2_voce():
    n = 200
    rng = np.random.default_rng(2)
    scale = 1e-6
    # tiny strains with a yield stress scaled down so that the step is plastic
    hard = onp.VoceHardening(SIG0_F * 1e-4, SIGU_F * 1e-4, B_F)
    H = scale * rng.standard_normal((n, 3, 3))
    eps = onp.tensor_to_mandel(0.5 * (H + H.transpose(0, 2, 1)))
    r_ss = onp.j2_update(eps, np.zeros((n, 6)), np.zeros(n), E, NU, hard)
    r_fs = onp.fefp_update(onp.tensor_to_nsym(np.eye(3) + H), onp.fefp_initial_state(n)["cpinv"], np.zeros(n), E, NU, hard)
    assert r_ss["plastic"].mean() > 0.5
    P = onp.nsym_to_tensor(r_fs["P"])
    sig = onp.mandel_to_tensor(r_ss["sig"])
    assert np.abs(P - sig).max() < 50 * scale * np.abs(sig).max()  # O(|H|) relative difference
    assert np.abs(r_fs["p"] - r_ss["p"]).max() < 50 * scale * r_ss["p"].max()


def test_fefp_objectivity_of_the_oracle():
    n = 64
    rng = np.random.default_rng(10)
    F = np.eye(3) + 0.04 * rng.standard_normal((n, 3, 3))
    Q, _ = np.linalg.qr(rng.standard_normal((n, 3, 3)))
    Q *= np.sign(np.linalg.det(Q))[:, None, None]
    cp = onp.fefp_initial_state(n)["cpinv"]
    r1 = onp.fefp_update(onp.tensor_to_nsym(F), cp, np.zeros(n), E, NU, HARD_F, tangent=False)
    r2 = onp.fefp_update(onp.tensor_to_nsym(Q @ F), cp, np.zeros(n), E, NU, HARD_F, tangent=False)
    assert np.allclose(onp.nsym_to_tensor(r2["P"]), Q @ onp.nsym_to_tensor(r1["P"]), rtol=0, atol=1e-10 * np.abs(r1["P"]).max())
    assert np.allclose(r1["p"], r2["p"], rtol=1e-11, atol=1e-18)
    assert np.allclose(r1["cpinv"], r2["cpinv"], rtol=0, atol=1e-13)  # material tensor: unchanged


def test_fefp_self_golden_reference_smoke_path():
    """Self-golden (parity unpinned): the oracle's stresses along the exact path of
    tests/test_FeFp_jax.py:21-33 (Nbatch=10, 19 steps), committed as tests/golden/fefp_self.npz."""
    g = np.load(os.path.join(GOLDEN, "fefp_self.npz"))
    st = onp.fefp_initial_state(10)
    cp, p = st["cpinv"], st["p"]
    for k, F in enumerate(fefp_path(10, pert=0.0)):
        r = onp.fefp_update(F, cp, p, E, NU, HARD_F, tangent=False)
        assert np.allclose(r["P"], g["P"][k], rtol=1e-12, atol=1e-9)
        assert np.allclose(r["p"], g["p"][k], rtol=1e-11, atol=1e-18)
        cp, p = r["cpinv"], r["p"]


def test_fefp_c_oracle_equals_numpy_oracle():
    n = 257
    st = onp.fefp_initial_state(n)
    cp, p = st["cpinv"], st["p"]
    for k, F in enumerate(fefp_path(n)):
        r = onp.fefp_update(F, cp, p, E, NU, HARD_F)
        c = oracle_c.fefp(F, cp, p, E, NU, SIG0_F, SIGU_F, B_F, nthreads=2)
        safe = np.abs(r["f_trial"]) > 1e-9 * SIG0_F
        for key in ("P", "Ct", "be_bar", "cpinv", "p"):
            assert np.abs(c[key][safe] - r[key][safe]).max() <= 1e-12 * max(np.abs(r[key]).max(), 1e-300), (k, key)
        assert c["n_not_converged"] == 0
        cp, p = r["cpinv"], r["p"]


def test_elastic_known_answer_nu_zero():
    """tests/mfront/test_initialization.py:113-132: sigma[:3] = E * [1e-3, 0, 0] for nu = 0."""
    eps = np.array([[1e-3, 0, 0, 0, 0, 0.0]])
    for fn in (onp.elastic_iso, oracle_c.elastic_iso):
        sig, Ct = fn(eps, 70e3, 0.0)
        assert np.allclose(sig[0, :3], 70e3 * np.array([1e-3, 0, 0]), rtol=1e-14, atol=1e-12)
        assert np.allclose(Ct[0], 70e3 * np.eye(6), rtol=1e-14)


def test_voce_return_mapping_against_an_independent_root_finder():
    """The scalar return-mapping equation solved by scipy's bracketing root finder (no Newton, no
    shared code) gives the same plastic multiplier as the oracle's Newton."""
    from scipy.optimize import brentq

    hard = HARDS["voce"][0]
    n = 300
    eps = j2_history(n, seed=13, sig0=hard.sig0)[2]
    epsp_n, p_n = random_j2_state(n, sig0=hard.sig0)
    r = onp.j2_update(eps, epsp_n, p_n, E, NU, hard)
    _, mu = onp.lame(E, NU)
    e = eps - epsp_n
    e[:, :3] -= e[:, :3].mean(axis=1)[:, None]
    seq = np.sqrt(1.5) * 2 * mu * np.sqrt((e * e).sum(axis=1))
    for i in np.nonzero(r["plastic"])[0][:100]:
        f = lambda dp: seq[i] - 3 * mu * dp - hard.R(p_n[i] + dp)  # noqa: E731
        dp = brentq(f, 0.0, seq[i] / (3 * mu), xtol=1e-18, rtol=1e-15)
        assert abs(dp - (r["p"][i] - p_n[i])) < 1e-12 * max(dp, 1e-12)


def test_fefp_solution_satisfies_the_full_tensorial_system():
    """The oracle solves a reduced 2x2 system in (dp, Ie).  Its result must satisfy the original
    seven equations of the formulation (plastic consistency, flow rule for dev(be_bar), and
    det(be_bar) = 1) that the reduction was derived from."""
    n = 400
    rng = np.random.default_rng(11)
    F = np.eye(3) + 0.06 * rng.standard_normal((n, 3, 3))
    st = onp.fefp_initial_state(n)
    r0 = onp.fefp_update(onp.tensor_to_nsym(np.eye(3) + 0.03 * rng.standard_normal((n, 3, 3))), st["cpinv"], st["p"], E, NU, HARD_F, tangent=False)
    r = onp.fefp_update(onp.tensor_to_nsym(F), r0["cpinv"], r0["p"], E, NU, HARD_F, tangent=False)
    assert r["plastic"].mean() > 0.8
    _, mu = onp.lame(E, NU)
    G = onp.mandel_to_tensor(r0["cpinv"])
    J = onp._det3(F)
    btr = (J ** (-2 / 3))[:, None, None] * (F @ G @ F.transpose(0, 2, 1))
    be = onp.mandel_to_tensor(r["be_bar"])
    dp = r["p"] - r0["p"]
    I3 = np.eye(3)
    dev = lambda A: A - (np.trace(A, axis1=1, axis2=2) / 3)[:, None, None] * I3  # noqa: E731
    s = mu * dev(be)
    seq = np.sqrt(1.5 * (s * s).sum((1, 2)))
    pl = r["plastic"]
    # (1) consistency f = 0 on plastic points, f <= 0 and dp = 0 on elastic ones
    assert np.abs(seq[pl] - HARD_F.R(r["p"][pl])).max() < 1e-9 * SIG0_F
    assert (seq[~pl] <= HARD_F.R(r["p"][~pl]) + 1e-9).all() and np.all(dp[~pl] == 0)
    # (2) flow rule: dev(be - be_trial) + 2 dp tr(be)/3 n = 0 with n = 3 s / (2 seq)
    nrm = 1.5 * s / seq[:, None, None]
    res = dev(be - btr) + (2 * dp * np.trace(be, axis1=1, axis2=2) / 3)[:, None, None] * nrm
    assert np.abs(res[pl]).max() < 1e-13
    # (3) isochoric elastic left Cauchy-Green tensor
    assert np.abs(onp._det3(be) - 1).max() < 1e-13
    # (4) stress: tau = kappa/2 (J^2 - 1) 1 + s,  P = tau F^-T
    lm, _ = onp.lame(E, NU)
    kappa = lm + 2 * mu / 3
    tau = 0.5 * kappa * (J * J - 1)[:, None, None] * I3 + s
    P = tau @ np.linalg.inv(F).transpose(0, 2, 1)
    assert np.abs(onp.nsym_to_tensor(r["P"]) - P).max() < 1e-9


def test_fefp_c_oracle_linear_hardening():
    n = 129
    hard = onp.LinearHardening(400.0, 2e3)
    st = onp.fefp_initial_state(n)
    cp, p = st["cpinv"], st["p"]
    for F in fefp_path(n)[::4]:
        r = onp.fefp_update(F, cp, p, E, NU, hard)
        c = oracle_c.fefp(F, cp, p, E, NU, 400.0, 2e3, kind=0)
        safe = np.abs(r["f_trial"]) > 1e-9 * 400.0
        for key in ("P", "Ct", "p"):
            assert np.abs(c[key][safe] - r[key][safe]).max() <= 1e-12 * max(np.abs(r[key]).max(), 1e-300), key
        cp, p = r["cpinv"], r["p"]


@pytest.mark.parametrize("kind", ["voce", "linear"])
def test_fefp_uniaxial_kirchhoff_stress_path_known_answer(kind):
    """ORACLE, finite strain: 257 uniaxial Kirchhoff-stress paths (helpers.fefp_uniaxial_known_answer): tau_xx = R(p) at the plastic
    points, det(be_bar) = 1 with be_bar carrying exactly the deviatoric stress, kappa/2 (J^2 - 1) = tau_xx / 3 -- statements about the
    MODEL that hold for any integration algorithm, at 1e-9, for both hardening laws (tests/test_FeFp_jax.py:14-15 for Voce)."""
    from helpers import B_F, SIG0_F, SIGU_F, fefp_uniaxial_known_answer

    n = 257
    if kind == "voce":
        hard, R = onp.VoceHardening(SIG0_F, SIGU_F, B_F), (lambda q: SIG0_F + (SIGU_F - SIG0_F) * (1.0 - np.exp(-B_F * q)))
    else:
        hard, R = onp.LinearHardening(SIG0_F, 2e3), (lambda q: SIG0_F + 2e3 * q)
    st = onp.fefp_initial_state(n)
    state = {"cpinv": st["cpinv"], "p": st["p"], "next": None}

    def step(F9):
        r = onp.fefp_update(F9, state["cpinv"], state["p"], E, NU, hard)
        assert not r["notconv"].any()
        state["next"] = (r["cpinv"], r["p"])
        return r["P"], r["Ct"], r["p"], r["be_bar"]

    def advance():
        state["cpinv"], state["p"] = state["next"]

    nplastic, pmax = fefp_uniaxial_known_answer(step, advance, R, n=n)
    assert nplastic > 200 and pmax > 1e-2
