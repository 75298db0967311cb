"""Generates jaxmat_*.npz: golden vectors of the reference's JAX path for BASELINE configs 2-4.

Runs ONLY where the third-party arithmetic the reference delegates to is importable
(``jax``, ``equinox``, ``jaxmat >= 0.0.1``: ``setup.cfg:19-20``, imports ``dolfinx_materials/jaxmat.py:12-16``).
None of them exists in the build container or on the GPU box (no network), so the parity of the J2-Voce and
FeFp laws against jaxmat is UNPINNED until somebody runs this script and commits its output:

    pip install jax jaxmat            # anywhere with network access
    python tests/golden/make_jaxmat_golden.py [/path/to/dolfinx_materials/checkout]
    git add tests/golden/jaxmat_*.npz

``tests/test_jaxmat_golden.py`` picks the files up (CPU: oracle vs jaxmat; GPU: HIP kernels vs jaxmat) and
is skipped while they are absent.

Two recordings per law, because of a reference bug on this path (``jaxmat.py:135-138``: the converted state is
discarded, every ``JAXMaterial.integrate`` starts from ``behavior.init_state``):
  ``*_adapter``   what the reference's own ``JAXMaterial.integrate`` returns call by call (needs the checkout;
                  ``dolfinx.common.Timer`` is stubbed when dolfinx is missing): every step from the virgin state;
  ``*_stateful``  ``jit(vmap(jacfwd(behavior.constitutive_update, has_aux=True)))`` -- the construction of
                  ``jaxmat.py:147-155`` -- driven with the state carried from step to step (the intended semantics,
                  ``generic.py:176-216``).
Inputs are the seeded histories of ``tests/helpers.py``; parameters as in
``tests/test_FeFp_jax.py:7-15`` and ``demos/jax/elastoplasticity/plane_elastoplasticity.py:60-71``.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    try:
        import jax
        import jax.numpy as jnp
        import jaxmat.materials as jm
    except Exception as exc:  # the normal case in the build container
        print(f"jaxmat / jax not importable here ({exc!r}): nothing generated; parity stays unpinned")
        return 1
    jax.config.update("jax_enable_x64", True)
    from helpers import E, NU, SIG0_V, SIGU_V, B_V, SIG0_F, SIGU_F, B_F, SIG0_LIN, H_LIN, j2_history, fefp_path

    ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    JAXMaterial = None
    if os.path.isdir(os.path.join(ref_root, "dolfinx_materials")):
        try:
            import dolfinx.common  # noqa: F401
        except Exception:  # generic.py:2 imports a Timer it never uses
            import contextlib

            d, dc = types.ModuleType("dolfinx"), types.ModuleType("dolfinx.common")
            dc.Timer = lambda name: contextlib.nullcontext()
            d.common = dc
            sys.modules.setdefault("dolfinx", d)
            sys.modules.setdefault("dolfinx.common", dc)
        sys.path.insert(0, ref_root)
        try:
            from dolfinx_materials.jaxmat import JAXMaterial
        except Exception as exc:
            print(f"reference adapter not importable ({exc!r}): only the *_stateful recordings are written")

    el = jm.LinearElasticIsotropic(E=E, nu=NU)

    def record(name, behavior, history, n):
        out = {"n": n, "E": E, "nu": NU, "gradients": np.stack([np.asarray(h) for h in history])}
        # ---- stateful: the construction of jaxmat.py:147-155 with the state carried forward
        def upd(g, state, dt):
            grad = getattr(state, "strain" if g.shape[-1] == 6 else "F").__class__(array=g)
            flux, new_state = behavior.constitutive_update(grad, state, dt)
            return flux.array, (flux.array, new_state)

        batched = jax.jit(jax.vmap(jax.jacfwd(upd, argnums=0, has_aux=True), in_axes=(0, 0, None)))
        state = behavior.init_state(n)
        fl, ct, pp = [], [], []
        for g in history:
            Ct, (flux, state) = batched(jnp.asarray(g), state, 0.0)
            fl.append(np.asarray(flux)); ct.append(np.asarray(Ct)); pp.append(np.asarray(state.internal.p))
        out.update(stateful_flux=np.stack(fl), stateful_Ct=np.stack(ct), stateful_p=np.stack(pp))
        if hasattr(state.internal, "be_bar"):
            out["stateful_be_bar_last"] = np.asarray(state.internal.be_bar.array)
        # ---- adapter: the reference's JAXMaterial as it is
        if JAXMaterial is not None:
            mat = JAXMaterial(behavior)
            mat.set_data_manager(n)
            fl, ct, iv = [], [], []
            for g in history:
                flux, isv, Ct = mat.integrate(jnp.asarray(g), 0)
                mat.data_manager.update()
                fl.append(np.asarray(flux)); ct.append(np.asarray(Ct)); iv.append(np.asarray(isv))
            out.update(adapter_flux=np.stack(fl), adapter_Ct=np.stack(ct), adapter_isv=np.stack(iv),
                       adapter_isv_names=np.array(mat.internal_state_variable_names))
        np.savez_compressed(os.path.join(HERE, f"jaxmat_{name}.npz"), **out)
        print("wrote", f"jaxmat_{name}.npz", {k: getattr(v, "shape", v) for k, v in out.items()})

    n = 256
    record("j2_voce", jm.vonMisesIsotropicHardening(elasticity=el, yield_stress=jm.VoceHardening(sig0=SIG0_V, sigu=SIGU_V, b=B_V)),
           j2_history(n, sig0=SIG0_V), n)

    def lin(p):   # linear hardening as a callable (jaxmat has no LinearHardening class in every version)
        return SIG0_LIN + H_LIN * p

    record("j2_linear", jm.vonMisesIsotropicHardening(elasticity=el, yield_stress=lin), j2_history(n), n)
    n = 64
    record("fefp_voce", jm.FeFpJ2Plasticity(elasticity=el, yield_stress=jm.VoceHardening(sig0=SIG0_F, sigu=SIGU_F, b=B_F)), fefp_path(n), n)
    return 0


if __name__ == "__main__":
    sys.exit(main())
