"""Shared synthetic inputs for the parity tests (seeded)."""
import numpy as np

from oracle import constitutive_np as onp

E, NU = 70e3, 0.3
SIG0_LIN, H_LIN = 250.0, 5e3          # tests/mfront/test_initialization.py:47-52
SIG0_V, SIGU_V, B_V = 350.0, 500.0, 1e3  # demos/jax/elastoplasticity/plane_elastoplasticity.py:60-71


def eps_yield(sig0=SIG0_LIN):
    _, mu = onp.lame(E, NU)
    return sig0 / (2 * mu) * np.sqrt(2.0 / 3.0)


def j2_history(n, seed=1234, sig0=SIG0_LIN, amp=4.0):
    """K=4 increments: proportional loading eps_k = (k/3) eps_hat, k=1..3, then unloading to
    0.5 eps_hat (cfg 2 of SURVEY.md 8(d))."""
    rng = np.random.default_rng(seed)
    d = rng.standard_normal((n, 6))
    d /= np.linalg.norm(d, axis=1)[:, None]
    s = rng.uniform(0.0, amp, n) * eps_yield(sig0)
    eps_hat = d * s[:, None]
    return [eps_hat / 3.0, eps_hat * (2.0 / 3.0), eps_hat, 0.5 * eps_hat]


def random_j2_state(n, seed=7, sig0=SIG0_LIN):
    rng = np.random.default_rng(seed)
    epsp = 0.3 * eps_yield(sig0) * rng.standard_normal((n, 6))
    epsp[:, :3] -= epsp[:, :3].mean(axis=1)[:, None]
    p = rng.uniform(0.0, 2e-3, n)
    return epsp, p
