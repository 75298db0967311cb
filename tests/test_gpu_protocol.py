"""GPU: the reference-plumbing golden sequence and the uniaxial known answer, replayed through
HIPMaterial (C ABI -> HIP kernels)."""
import os

import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd.jaxmat import JAXMaterial

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _j2(E, nu, sig0, H):
    return JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=nu), jm.LinearHardening(sig0, H)))


def test_protocol_sequence_matches_reference_plumbing_golden():
    """integrate / update / revert sequence recorded through the reference's own
    Material.integrate + DataManager (generic.py:176-216): same fluxes, ISV block layout
    (p first, then epsp), tangents and s0/s1 dictionaries after every call."""
    g = np.load(os.path.join(GOLDEN, "protocol_ref.npz"))
    n = g["eps_hat"].shape[0]
    mat = _j2(70e3, 0.3, 250.0, 5e3)
    mat.set_data_manager(n)
    for k, (op, sc) in enumerate(zip(g["script"], g["scale"])):
        if op == "integrate":
            flux, isv, Ct = mat.integrate(sc * g["eps_hat"])
            assert np.allclose(flux, g[f"flux_{k}"], rtol=1e-12, atol=1e-9)
            assert np.allclose(isv, g[f"isv_{k}"], rtol=1e-12, atol=1e-18)
            assert np.allclose(Ct, g[f"Ct_{k}"], rtol=1e-12, atol=1e-8)
        elif op == "update":
            mat.data_manager.update()
        else:
            mat.data_manager.revert()
        s0, s1 = mat.get_initial_state_dict(), mat.get_final_state_dict()
        for key in ("stress", "p", "epsp"):
            tol = 1e-9 if key == "stress" else 1e-18
            assert np.allclose(s0[key], g[f"s0_{key}_{k}"], rtol=1e-12, atol=tol), (k, op, key)
            assert np.allclose(s1[key], g[f"s1_{key}_{k}"], rtol=1e-12, atol=tol), (k, op, key)


def test_uniaxial_tension_known_answer():
    """Material-point replay of tests/mfront/test_elastoplasticity.py:14-36: the final stress is
    2/sqrt(3) [sig0, 0, sig0/2] (rtol 1e-2, the reference test's own tolerance) and every step
    matches the committed path to 1e-12."""
    g = np.load(os.path.join(GOLDEN, "j2_uniaxial_kat.npz"))
    mat = _j2(float(g["E"]), float(g["nu"]), float(g["sig0"]), float(g["H"]))
    mat.set_data_manager(1)
    for eps, sig in zip(g["strain"][1:], g["stress"][1:]):
        s, _, _ = mat.integrate(eps[None])
        assert np.allclose(s[0], sig, rtol=1e-12, atol=1e-9)
        mat.data_manager.update()
    assert np.allclose(s[0, :3], g["expected"], rtol=1e-2, atol=1e-8)
