#!/usr/bin/env python3
"""Stand-alone batch use of a material, without any FE code: the call sequence of the reference's
``tests/test_FeFp_jax.py:6-33`` (``set_data_manager`` -> loop { ``integrate`` ->
``data_manager.update()`` }) with the MI355X engine.  Only the two imports differ from the
reference script (and the Voce law is a ``jm.VoceHardening`` object instead of a Python function).

    python examples/standalone_batch.py [Nbatch]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import dolfinx_materials_amd.materials as jm  # reference: import jaxmat.materials as jm
from dolfinx_materials_amd.jaxmat import JAXMaterial  # reference: from dolfinx_materials.jaxmat import JAXMaterial


def main(Nbatch=10):
    E, nu, sig0, b, sigu = 70e3, 0.3, 500.0, 1000, 750.0
    elastic_model = jm.LinearElasticIsotropic(E=E, nu=nu)
    # exactly what the reference passes (tests/test_FeFp_jax.py:14-15), with np for jnp: the callable is traced and
    # compiled into the kernels on construction; jm.VoceHardening(sig0, sigu, b) is the built-in equivalent
    def yield_stress(p):
        return sig0 + (sigu - sig0) * (1 - np.exp(-b * p))

    behavior = jm.FeFpJ2Plasticity(elasticity=elastic_model, yield_stress=yield_stress)
    material = JAXMaterial(behavior)
    material.set_data_manager(Nbatch)

    eps, Nsteps, dt = 2e-2, 20, 0
    for t in np.linspace(0, 1.0, Nsteps)[1:]:
        F = np.zeros((Nbatch, 9))
        F[:, 0] = 1 + eps * t
        F[:, [1, 2]] = 1 - eps / 2 * t
        P, isv, Ct = material.integrate(F, dt)
        material.data_manager.update()
        print(f"t={t:.3f}  P11={P[0, 0]:9.4f}  p={isv[0, 0]:.6f}  plastic points={material.last_stats['n_plastic']}")
    # The same last increment with the state passed in and handed back EXPLICITLY -- ``sig, new_state = material.constitutive_update(
    # eps, state, dt)`` of docs/jax.md:46-50 at one point, ``batched_constitutive_update`` (jaxmat.py:147-155) for a batch: the
    # material's own s0 / s1 are not touched (here they already hold the accepted last increment).
    state = material.natural_state(Nbatch)                     # F = I, be_bar = I, p = 0
    for t in np.linspace(0, 1.0, Nsteps)[1:]:
        F = np.zeros((Nbatch, 9))
        F[:, 0] = 1 + eps * t
        F[:, [1, 2]] = 1 - eps / 2 * t
        Ct2, state = material.batched_constitutive_update(F, state, dt)
    P0, state0 = material.constitutive_update(F[0], {k: v[0] for k, v in material.natural_state(1).items()}, dt)   # one virgin point, one step
    print(f"explicit state: P11={state['PK1'][0, 0]:9.4f}  p={state['p'][0, 0]:.6f}  (max |P - P_explicit| = {np.abs(state['PK1'] - P).max():.2e})")
    # P and Ct own their (page-locked) memory and outlive the material; isv is a lazy view of its state: materialise it
    return P, np.array(isv), Ct


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
