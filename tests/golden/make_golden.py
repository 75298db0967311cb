"""Generates the committed golden vectors under tests/golden/ (run in the build container only:
it imports the reference from /root/reference, which never travels to the GPU box).

    python tests/golden/make_golden.py

Fixtures
  elastic_ref.npz    LinearElasticIsotropic(70e3, 0.3) run through the reference's own
                     Material.integrate (python_materials/elasticity.py:21-24 via
                     generic.py:176-189): eps (64,6) seed 0 -> sig, Ct.        [reference output]
  protocol_ref.npz   a stateful J2 law (literal transcription of
                     tests/mfront/IsotropicLinearHardeningPlasticity.mfront:49-77 as a per-point
                     `constitutive_update(eps, state, dt)`) driven through the reference's
                     Material / DataManager / MaterialStateManager machinery
                     (generic.py:103-295): integrate, update, integrate, integrate, revert ...
                     Records what the reference plumbing returns at each call: flux, ISV block
                     (hstack order), tangent, and the s0/s1 dictionaries.  [reference plumbing,
                     build's law]
  j2_uniaxial_kat.npz  material-point replay of tests/mfront/test_elastoplasticity.py:14-36
                     (E=70e3, nu=0.3, H=1e-6, sig0=250, 50 steps to eps_xx=2e-2, eps_zz=0,
                     sig_yy=0): strain path and stresses from the oracle; the known answer
                     2/sqrt(3)*[sig0, 0, sig0/2] (rtol 1e-2) is asserted at generation time.
  fefp_self.npz      SELF-golden (parity unpinned): the build's FeFp oracle along the path of
                     tests/test_FeFp_jax.py:21-33 (Nbatch=10, 19 steps).
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import constitutive_np as onp  # noqa: E402
from oracle.ref_import import import_reference  # noqa: E402

generic, python_materials = import_reference()
warnings.simplefilter("ignore")


def make_elastic():
    E, nu = 70e3, 0.3
    rng = np.random.default_rng(0)
    eps = 1e-3 * rng.standard_normal((64, 6))
    mat = python_materials.LinearElasticIsotropic(E, nu)
    mat.set_data_manager(64)
    sig, isv, Ct = mat.integrate(eps)
    assert isv.shape == (64, 0)
    np.savez(os.path.join(HERE, "elastic_ref.npz"), E=E, nu=nu, eps=eps, sig=np.array(sig), Ct=np.array(Ct))


class J2Reference(generic.Material):
    """Per-point law in the reference's own plug-in form (docs/jax.md:46-50 signature)."""

    def __init__(self, E, nu, sig0, H):
        super().__init__()
        self.E, self.nu, self.sig0, self.H = E, nu, sig0, H

    @property
    def gradients(self):
        return {"strain": 6}

    @property
    def fluxes(self):
        return {"stress": 6}

    @property
    def internal_state_variables(self):
        return {"p": 1, "epsp": 6}

    def constitutive_update(self, eps, state, dt):
        epsp_n = state["epsp"]
        p_n = state["p"]
        sig, eel, p, Dt = onp.j2_update_mfront_form(
            eps[None, :], -epsp_n[None, :], np.atleast_1d(p_n), self.E, self.nu, self.H, self.sig0
        )
        state["strain"] = eps
        state["stress"] = sig[0]
        state["p"] = p
        state["epsp"] = eps - eel[0]
        return Dt[0], state


def make_protocol():
    E, nu, sig0, H = 70e3, 0.3, 250.0, 5e3
    n = 16
    _, mu = onp.lame(E, nu)
    epsy = sig0 / (2 * mu) * np.sqrt(2 / 3)
    rng = np.random.default_rng(42)
    d = rng.standard_normal((n, 6))
    d /= np.linalg.norm(d, axis=1)[:, None]
    eps_hat = d * rng.uniform(0.5, 4.0, n)[:, None] * epsy
    mat = J2Reference(E, nu, sig0, H)
    mat.set_data_manager(n)
    script = ["integrate", "update", "integrate", "integrate", "revert", "integrate", "update", "integrate", "update"]
    scale = [0.4, None, 0.8, 1.0, None, 1.0, None, 0.5, None]
    rec = {"eps_hat": eps_hat, "script": np.array(script), "scale": np.array([np.nan if s is None else s for s in scale])}
    for k, (op, sc) in enumerate(zip(script, scale)):
        if op == "integrate":
            flux, isv, Ct = mat.integrate(sc * eps_hat)
            rec[f"flux_{k}"] = np.array(flux)
            rec[f"isv_{k}"] = np.array(isv)
            rec[f"Ct_{k}"] = np.array(Ct)
        elif op == "update":
            mat.data_manager.update()
        else:
            mat.data_manager.revert()
        s0 = mat.get_initial_state_dict()
        s1 = mat.get_final_state_dict()
        for key in ("stress", "p", "epsp"):
            rec[f"s0_{key}_{k}"] = np.array(s0[key])
            rec[f"s1_{key}_{k}"] = np.array(s1[key])
    np.savez(os.path.join(HERE, "protocol_ref.npz"), **rec)


def make_uniaxial():
    E, nu, sig0, H = 70e3, 0.3, 250.0, 1e-6
    hard = onp.LinearHardening(sig0, H)
    Exx = np.linspace(0, 2e-2, 51)
    epsp = np.zeros((1, 6))
    p = np.zeros(1)
    eyy = 0.0
    path, sigs = [np.zeros(6)], [np.zeros(6)]
    for exx in Exx[1:]:
        # plane-strain-like uniaxial tension: eps_zz = 0, solve sig_yy = 0 for eps_yy with the
        # consistent tangent (what the FE solve of tests/uniaxial_tension.py does globally)
        for _ in range(50):
            eps = np.array([[exx, eyy, 0, 0, 0, 0.0]])
            r = onp.j2_update(eps, epsp, p, E, nu, hard)
            if abs(r["sig"][0, 1]) < 1e-10:
                break
            eyy -= r["sig"][0, 1] / r["Ct"][0, 1, 1]
        epsp, p = r["epsp"], r["p"]
        path.append(eps[0].copy())
        sigs.append(r["sig"][0].copy())
    sigs = np.array(sigs)
    expected = 2 / np.sqrt(3) * np.array([sig0, 0, sig0 / 2])
    assert np.allclose(sigs[-1, :3], expected, rtol=1e-2, atol=1e-8), sigs[-1]
    np.savez(os.path.join(HERE, "j2_uniaxial_kat.npz"), E=E, nu=nu, sig0=sig0, H=H, strain=np.array(path), stress=sigs, expected=expected)


def make_fefp_self():
    """Self-golden (parity unpinned: tests/test_FeFp_jax.py has no assertions and jaxmat is
    absent): stresses and p of the build's own FeFp oracle along the exact driver path of
    tests/test_FeFp_jax.py:21-33."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import E, NU, SIG0_F, SIGU_F, B_F, fefp_path

    hard = onp.VoceHardening(SIG0_F, SIGU_F, B_F)
    st = onp.fefp_initial_state(10)
    cp, p = st["cpinv"], st["p"]
    Ps, ps, Fs = [], [], []
    for F in fefp_path(10, pert=0.0):
        r = onp.fefp_update(F, cp, p, E, NU, hard, tangent=False)
        Ps.append(r["P"]); ps.append(r["p"]); Fs.append(F)
        cp, p = r["cpinv"], r["p"]
    np.savez(os.path.join(HERE, "fefp_self.npz"), F=np.array(Fs), P=np.array(Ps), p=np.array(ps))


if __name__ == "__main__":
    make_fefp_self()
    make_elastic()
    make_protocol()
    make_uniaxial()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
