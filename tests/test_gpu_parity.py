"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Tolerances: the contract
is rtol 1e-10 elastic / 1e-6 plastic (BASELINE.json north_star); both sides are fp64 and the
tests hold them to 1e-12 relative to the field scale, away from the yield kink."""
import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd.jaxmat import JAXMaterial
from dolfinx_materials_amd.python_materials import LinearElasticIsotropic
from oracle import constitutive_np as onp

from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, j2_history, random_j2_state

pytestmark = pytest.mark.gpu

RTOL_ELASTIC = 1e-10
RTOL_PLASTIC = 1e-6
TIGHT = 1e-12


def relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def make_j2(kind):
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if kind == "linear":
        return JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))), onp.LinearHardening(SIG0_LIN, H_LIN)
    return JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V))), onp.VoceHardening(SIG0_V, SIGU_V, B_V)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097])
def test_elastic_matches_oracle(n):
    rng = np.random.default_rng(0)
    eps = 1e-3 * rng.standard_normal((n, 6))
    mat = LinearElasticIsotropic(E, NU)
    mat.set_data_manager(n)
    sig, isv, Ct = mat.integrate(eps)
    so, Co = onp.elastic_iso(eps, E, NU)
    assert sig.shape == (n, 6) and isv.shape == (n, 0) and Ct.shape == (n, 6, 6)
    assert relerr(sig, so) < RTOL_ELASTIC
    assert relerr(Ct, Co) < RTOL_ELASTIC
    assert mat.last_stats["n_nan"] == 0 and mat.last_stats["n_plastic"] == 0


def test_elastic_golden_reference():
    """Against vectors produced by the reference itself (tests/golden/make_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "elastic_ref.npz"))
    mat = LinearElasticIsotropic(float(g["E"]), float(g["nu"]))
    mat.set_data_manager(g["eps"].shape[0])
    sig, _, Ct = mat.integrate(g["eps"])
    assert np.allclose(sig, g["sig"], rtol=RTOL_ELASTIC, atol=0)
    assert np.allclose(Ct, g["Ct"], rtol=RTOL_ELASTIC, atol=1e-9)


@pytest.mark.parametrize("kind", ["linear", "voce"])
@pytest.mark.parametrize("n", [1, 64, 1000, 70001])
def test_j2_single_step_random_state(kind, n):
    mat, hard = make_j2(kind)
    mat.set_data_manager(n)
    epsp_n, p_n = random_j2_state(n, sig0=hard.sig0)
    eps = j2_history(n, seed=99, sig0=hard.sig0)[2]
    mat.set_initial_state_dict({"p": p_n, "epsp": epsp_n})
    s0 = mat.get_initial_state_dict()
    assert np.array_equal(s0["p"][:, 0], p_n) and np.array_equal(s0["epsp"], epsp_n)
    sig, isv, Ct = mat.integrate(eps)
    ref = onp.j2_update(eps, epsp_n, p_n, E, NU, hard)
    # points sitting numerically on the yield surface may legitimately take either branch
    safe = np.abs(ref["f_trial"]) > 1e-9 * hard.sig0
    assert safe.mean() > 0.99
    assert relerr(sig[safe], ref["sig"][safe]) < TIGHT
    assert relerr(isv[safe, 0], ref["p"][safe]) < TIGHT
    assert np.abs(isv[safe, 1:] - ref["epsp"][safe]).max() < TIGHT * np.abs(ref["epsp"]).max()
    assert relerr(Ct[safe], ref["Ct"][safe]) < TIGHT
    assert relerr(sig[safe], ref["sig"][safe]) < RTOL_PLASTIC  # the stated contract
    st = mat.last_stats
    assert st["n_nan"] == 0 and st["n_not_converged"] == 0
    assert abs(st["n_plastic"] - int(ref["plastic"].sum())) <= int((~safe).sum())
    if kind == "voce":
        assert 1 <= st["max_local_iters"] <= 25
    fin = mat.get_final_state_dict()
    assert np.array_equal(fin["stress"], sig) and np.array_equal(fin["p"][:, 0], isv[:, 0])
    assert np.array_equal(fin["epsp"], isv[:, 1:])


@pytest.mark.parametrize("kind", ["linear", "voce"])
def test_j2_history_load_unload(kind):
    """4-increment load/unload history with advance() between increments."""
    n = 5000
    mat, hard = make_j2(kind)
    mat.set_data_manager(n)
    epsp = np.zeros((n, 6))
    p = np.zeros(n)
    fracs = []
    for eps in j2_history(n, sig0=hard.sig0):
        sig, isv, Ct = mat.integrate(eps)
        # a second Newton-iteration-like call from the same s0 must give the same answer
        sig2, isv2, Ct2 = mat.integrate(eps)
        assert np.array_equal(sig, sig2) and np.array_equal(Ct, Ct2)
        ref = onp.j2_update(eps, epsp, p, E, NU, hard)
        safe = np.abs(ref["f_trial"]) > 1e-9 * hard.sig0
        assert relerr(sig[safe], ref["sig"][safe]) < TIGHT
        assert relerr(Ct[safe], ref["Ct"][safe]) < TIGHT
        assert np.abs(isv[safe, 0] - ref["p"][safe]).max() < 1e-15 + TIGHT * ref["p"].max()
        fracs.append(ref["plastic"].mean())
        mat.data_manager.update()
        fin = mat.get_final_state_dict()
        ini = mat.get_initial_state_dict()
        assert np.array_equal(fin["p"], ini["p"]) and np.array_equal(fin["epsp"], ini["epsp"])
        epsp, p = ref["epsp"], ref["p"]
        # carry the device state (not the oracle's) forward only through the device
    assert fracs[2] > 0.5 and fracs[3] == 0.0  # mostly plastic at peak, elastic unloading


def test_revert_restores_initial_state():
    n = 300
    mat, hard = make_j2("linear")
    mat.set_data_manager(n)
    h = j2_history(n)
    mat.integrate(h[2])
    mat.data_manager.update()
    ini = mat.get_initial_state_dict()
    mat.integrate(h[2] * 1.5)
    changed = mat.get_final_state_dict()
    assert not np.array_equal(changed["p"], ini["p"])
    mat.data_manager.revert()
    fin = mat.get_final_state_dict()
    assert np.array_equal(fin["p"], ini["p"]) and np.array_equal(fin["epsp"], ini["epsp"])


def test_empty_batch():
    mat, _ = make_j2("linear")
    mat.set_data_manager(0)
    sig, isv, Ct = mat.integrate(np.zeros((0, 6)))
    assert sig.shape == (0, 6) and isv.shape == (0, 7) and Ct.shape == (0, 6, 6)


def test_unknown_state_field_rejected():
    mat, _ = make_j2("linear")
    mat.set_data_manager(4)
    with pytest.raises(AssertionError):
        mat.set_initial_state_dict({"nonsense": np.zeros((4, 1))})


def test_update_material_property_reaches_kernel():
    n = 256
    mat, hard = make_j2("linear")
    mat.set_data_manager(n)
    eps = j2_history(n)[2]
    mat.update_material_property("yield_stress.sig0", 300.0)
    sig, _, _ = mat.integrate(eps)
    ref = onp.j2_update(eps, np.zeros((n, 6)), np.zeros(n), E, NU, onp.LinearHardening(300.0, H_LIN))
    assert relerr(sig, ref["sig"]) < TIGHT


def test_large_batch_properties():
    """Full-size (1e7 points) size-independent properties: elastic unloading increments are
    linear in the strain increment; tangent is symmetric; p is non-decreasing."""
    n = 2_000_000
    mat, hard = make_j2("linear")
    mat.set_data_manager(n)
    h = j2_history(n)
    for eps in h[:3]:
        sig, isv, Ct = mat.integrate(eps)
        p_prev = mat.get_initial_state_dict()["p"][:, 0]
        assert (isv[:, 0] >= p_prev - 1e-18).all()
        assert np.abs(Ct - Ct.transpose(0, 2, 1)).max() < 1e-9
        mat.data_manager.update()
    sig3 = sig.copy()
    sig4, isv4, Ct4 = mat.integrate(h[3])
    C = onp.elastic_matrix(E, NU)
    assert np.allclose(sig4 - sig3, (h[3] - h[2]) @ C.T, rtol=0, atol=1e-9 * np.abs(sig3).max())
    assert mat.last_stats["n_plastic"] == 0


@pytest.mark.parametrize("kind", ["elastic", "linear", "voce"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1001, 70001])
def test_symmetric_packed_tangent_equals_upper_triangle_of_full(kind, n):
    """the 21-entry packed layout carries exactly the upper triangle of the
    full 6x6 tangent, everything else (stress, state) unchanged."""
    from dolfinx_materials_amd.conventions import pack_sym_tangent, unpack_sym_tangent

    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if kind == "elastic":
        beh, sig0 = jm.ElasticBehavior(el), SIG0_LIN
    elif kind == "linear":
        beh, sig0 = jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN)), SIG0_LIN
    else:
        beh, sig0 = jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V)), SIG0_V
    full, sym = JAXMaterial(beh), JAXMaterial(beh, tangent_layout="sym")
    full.set_data_manager(n)
    sym.set_data_manager(n)
    hard = None if kind == "elastic" else (onp.LinearHardening(SIG0_LIN, H_LIN) if kind == "linear" else onp.VoceHardening(SIG0_V, SIGU_V, B_V))
    epsp, p = np.zeros((n, 6)), np.zeros(n)
    for eps in j2_history(n, seed=31, sig0=sig0)[:3]:
        sf, isvf, cf = full.integrate(eps)
        ss, isvs, cs = sym.integrate(eps)
        assert cs.shape == (n, 21)
        assert np.array_equal(sf, ss) and np.array_equal(isvf, isvs)
        assert np.array_equal(cs, pack_sym_tangent(cf))
        assert np.array_equal(unpack_sym_tangent(cs), cf)   # the full block is exactly symmetric
        # ... and both are the oracle's tangent (the packed layout is checked against the CPU restatement, not
        # only against the other HIP kernel)
        if hard is None:
            ref_ct = np.broadcast_to(onp.elastic_matrix(E, NU), (n, 6, 6))
            safe = np.ones(n, dtype=bool)
        else:
            ref = onp.j2_update(eps, epsp, p, E, NU, hard)
            ref_ct, safe = ref["Ct"], np.abs(ref["f_trial"]) > 1e-9 * sig0
            epsp, p = ref["epsp"], ref["p"]
        assert np.abs(unpack_sym_tangent(cs)[safe] - ref_ct[safe]).max() <= 1e-12 * np.abs(ref_ct).max()
        full.data_manager.update()
        sym.data_manager.update()


@pytest.mark.parametrize("kind", ["linear", "voce"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1001, 70001])
def test_coefficient_tangent_layout_rebuilds_the_full_block_bit_for_bit(kind, n):
    """`tangent_layout="coef"`: the nine numbers (c1, c2, c3, n) of Ct = c1 1x1 + c2 I + c3 n x n per point; the block
    rebuilt from them equals the full-layout kernel's (1 ulp: numpy has no fused multiply-add) and the oracle's."""
    from dolfinx_materials_amd.conventions import tangent_from_coefficients

    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    hard_d, hard, sig0 = ((jm.LinearHardening(SIG0_LIN, H_LIN), onp.LinearHardening(SIG0_LIN, H_LIN), SIG0_LIN) if kind == "linear"
                          else (jm.VoceHardening(SIG0_V, SIGU_V, B_V), onp.VoceHardening(SIG0_V, SIGU_V, B_V), SIG0_V))
    beh = jm.vonMisesIsotropicHardening(el, hard_d)
    full, coef = JAXMaterial(beh), JAXMaterial(beh, tangent_layout="coef")
    full.set_data_manager(n)
    coef.set_data_manager(n)
    epsp, p = np.zeros((n, 6)), np.zeros(n)
    for eps in j2_history(n, seed=41, sig0=sig0)[:3]:
        sf, isvf, cf = full.integrate(eps)
        sc, isvc, cc = coef.integrate(eps)
        assert cc.shape == (n, 9) and np.array_equal(sf, sc) and np.array_equal(isvf, isvc)
        rebuilt = tangent_from_coefficients(cc)
        assert np.abs(rebuilt - cf).max() <= 2e-16 * np.abs(cf).max() * 4
        ref = onp.j2_update(eps, epsp, p, E, NU, hard)
        safe = np.abs(ref["f_trial"]) > 1e-9 * sig0
        assert np.abs(rebuilt[safe] - ref["Ct"][safe]).max() <= 1e-12 * np.abs(ref["Ct"]).max()
        epsp, p = ref["epsp"], ref["p"]
        full.data_manager.update()
        coef.data_manager.update()


@pytest.mark.parametrize("kind,n", [("linear", 1001), ("voce", 70_003)])
def test_pack4_tangent_layout_with_the_stress_rebuilds_the_full_block(kind, n):
    """`tangent_layout="pack4"`: (c1, c2, c3, w) per point; with the stress of the same update the block is
    c1 1x1 + c2 I + c3 n x n, n = dev(stress) w (`conventions.tangent_from_pack4`: numpy rounds the product and the sum of
    an entry separately, so 1 ulp of the c3 term from the kernel's fused form), and 1e-12 from the oracle."""
    from dolfinx_materials_amd.conventions import tangent_from_pack4

    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    hard_d, hard, sig0 = ((jm.LinearHardening(SIG0_LIN, H_LIN), onp.LinearHardening(SIG0_LIN, H_LIN), SIG0_LIN) if kind == "linear"
                          else (jm.VoceHardening(SIG0_V, SIGU_V, B_V), onp.VoceHardening(SIG0_V, SIGU_V, B_V), SIG0_V))
    beh = jm.vonMisesIsotropicHardening(el, hard_d)
    full, pack = JAXMaterial(beh), JAXMaterial(beh, tangent_layout="pack4")
    full.set_data_manager(n)
    pack.set_data_manager(n)
    assert pack.tangent_size == 4
    epsp, p = np.zeros((n, 6)), np.zeros(n)
    for eps in j2_history(n, seed=43, sig0=sig0)[:3]:
        sf, isvf, cf = full.integrate(eps)
        sp, isvp, cp = pack.integrate(eps)
        assert cp.shape == (n, 4) and np.array_equal(sf, sp) and np.array_equal(isvf, isvp)
        rebuilt = tangent_from_pack4(sp, cp)
        assert np.abs(rebuilt - cf).max() <= 2e-16 * np.abs(cf).max() * 4
        ref = onp.j2_update(eps, epsp, p, E, NU, hard)
        safe = np.abs(ref["f_trial"]) > 1e-9 * sig0
        assert np.abs(rebuilt[safe] - ref["Ct"][safe]).max() <= 1e-12 * np.abs(ref["Ct"]).max()
        assert not cp[~ref["plastic"] & safe][:, 2:].any()      # elastic points: c3 = 0, w = 0
        epsp, p = ref["epsp"], ref["p"]
        full.data_manager.update()
        pack.data_manager.update()
    full.close()
    pack.close()


def test_coefficient_layout_rejected_where_it_has_no_meaning():
    from dolfinx_materials_amd import _lib

    for beh in (jm.ElasticBehavior(jm.LinearElasticIsotropic(E=E, nu=NU)),
                jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(500.0, 750.0, 1e3))):
        with pytest.raises(_lib.DxmError):
            JAXMaterial(beh, tangent_layout="coef").set_data_manager(8)


def test_symmetric_layout_rejected_for_fefp():
    from dolfinx_materials_amd import _lib

    m = JAXMaterial(jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(500.0, 750.0, 1e3)), tangent_layout="sym")
    with pytest.raises(_lib.DxmError, match="not symmetric"):
        m.set_data_manager(8)


def test_elastic_known_answer_nu_zero():
    """tests/mfront/test_initialization.py:113-132: sigma[:3] = E * [1e-3, 0, 0] for nu = 0, also
    after update_material_property (the reference test changes E through the QuadratureMap)."""
    mat = LinearElasticIsotropic(70e3, 0.0)
    mat.set_data_manager(4)
    eps = np.tile(np.array([1e-3, 0, 0, 0, 0, 0.0]), (4, 1))
    sig, _, Ct = mat.integrate(eps)
    assert np.allclose(sig[:, :3], 70e3 * np.array([1e-3, 0, 0]), rtol=1e-14, atol=1e-12)
    mat.update_material_property("elasticity.E", 210e3)
    sig, _, Ct = mat.integrate(eps)
    assert np.allclose(sig[:, :3], 210e3 * np.array([1e-3, 0, 0]), rtol=1e-14, atol=1e-12)
    assert np.allclose(Ct[0], 210e3 * np.eye(6), rtol=1e-14)


@pytest.mark.parametrize("kind", ["voce", "linear"])
def test_hip_uniaxial_stress_paths_follow_the_closed_form_of_the_hardening_law(kind):
    """The KERNELS against a known answer that does not go through the oracle: 257 points, each on its own uniaxial-stress path
    (axial strain ramped to 0.3 ... 3 %), lateral strain found per point by Newton on sigma_yy = 0 with the kernel's own consistent
    tangent (several `integrate` calls from one s0, then `advance`: the cadence of a global Newton loop).  Constant flow direction =>
    the radial return is exact for any step:  sigma_xx = R(p),  eps_xx = sigma_xx / E + p,  eps_yy = -nu sigma_xx / E - p / 2,
    with R the law of tests/test_FeFp_jax.py:14-15 / the linear law of the MFront spec."""
    from scipy.optimize import brentq

    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if kind == "voce":
        s0, hard = SIG0_V, jm.VoceHardening(SIG0_V, SIGU_V, B_V)
        R = lambda q: SIG0_V + (SIGU_V - SIG0_V) * (1.0 - np.exp(-B_V * q))   # noqa: E731
    else:
        s0, hard = SIG0_LIN, jm.LinearHardening(SIG0_LIN, H_LIN)
        R = lambda q: SIG0_LIN + H_LIN * q   # noqa: E731
    n = 257
    m = JAXMaterial(jm.vonMisesIsotropicHardening(el, hard))
    m.set_data_manager(n)
    emax = np.linspace(3e-3, 3e-2, n)
    et = np.zeros(n)
    for t in np.linspace(0.0, 1.0, 9)[1:]:
        exx = t * emax
        for it in range(40):
            eps = np.zeros((n, 6))
            eps[:, 0], eps[:, 1], eps[:, 2] = exx, et, et
            sig, isv, ct = m.integrate(eps)
            syy = sig[:, 1].copy()
            if np.abs(syy).max() < 1e-10 * s0:
                break
            et = et - syy / (ct[:, 1, 1] + ct[:, 1, 2])
        assert np.abs(syy).max() < 1e-10 * s0 and it < 12, (t, it)
        pc = np.array([0.0 if e <= s0 / E else brentq(lambda q, e=e: R(q) / E + q - e, 0.0, e, xtol=1e-16, rtol=1e-15) for e in exx])
        sc = np.where(pc == 0.0, E * exx, R(pc))
        assert np.abs(sig[:, 0] - sc).max() < 1e-9 * s0 and np.abs(sig[:, 1:]).max() < 1e-9 * s0
        assert np.abs(np.asarray(isv)[:, 0] - pc).max() < 1e-12 and np.abs(et - (-NU * sc / E - pc / 2)).max() < 1e-12
        assert m.last_stats["n_not_converged"] == 0 and m.last_stats["n_nan"] == 0
        m.data_manager.update()
    assert (pc > 0).sum() > 200 and pc.max() > 1e-2
    m.close()
