#!/usr/bin/env python3
"""Hand-run campaign on top of the committed fuzz seeds (tests/test_gpu_fuzz.py): 120 further parameter draws of the J2
sweep, and an adversarial batch for the tangent direction n = dev(sigma) w -- strains whose hydrostatic part is 1e1 ... 1e7
times the deviatoric one (the deviator of the stress then loses digits to cancellation, as the trial deviator of the strain
does in any implementation)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import test_gpu_fuzz as t
bad = []
for s in range(8, 128):
    try:
        t.test_j2_random_parameters_and_nonproportional_histories(s)
    except AssertionError as e:
        bad.append((s, str(e)[:200]))
print("j2 seeds 8..127 failures:", bad)
# adversarial: hydrostatic part of the strain >> deviatoric part (the tangent direction is now formed from dev(sigma))
import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd.jaxmat import JAXMaterial
from oracle import constitutive_np as onp
E, NU, SIG0, H = 70e3, 0.3, 250.0, 5e3
mu = E / 2 / (1 + NU)
ey = SIG0 / (2 * mu) * np.sqrt(2 / 3)
n = 50_000
rng = np.random.default_rng(5)
for ratio in (1e1, 1e3, 1e5, 1e7):
    d = rng.standard_normal((n, 6)); d[:, :3] -= d[:, :3].mean(1)[:, None]; d /= np.linalg.norm(d, axis=1)[:, None]
    eps = d * (rng.uniform(1.0, 3.0, n) * ey)[:, None]
    eps[:, :3] += (ratio * ey)
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0, H)))
    m.set_data_manager(n)
    sig, isv, ct = m.integrate(eps)
    ref = onp.j2_update(eps, np.zeros((n, 6)), np.zeros(n), E, NU, onp.LinearHardening(SIG0, H))
    safe = np.abs(ref["f_trial"]) > 1e-9 * SIG0
    e_sig = np.abs(sig[safe] - ref["sig"][safe]).max() / np.abs(ref["sig"]).max()
    e_ct = np.abs(ct[safe] - ref["Ct"][safe]).max() / np.abs(ref["Ct"]).max()
    print(f"hydrostatic/deviatoric strain {ratio:.0e}: plastic {ref['plastic'].mean():.2f}  rel err stress {e_sig:.1e}  tangent {e_ct:.1e}  symmetric {np.array_equal(ct, ct.transpose(0,2,1))}")
    m.close()
