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


# FeFp (tests/test_FeFp_jax.py:7-15)
SIG0_F, SIGU_F, B_F = 500.0, 750.0, 1000.0


def fefp_path(n, nsteps=19, eps=2e-2, seed=4321, pert=0.2, n_exact=10):
    """F_k = I + t_k (eps diag(1,-1/2,-1/2) + pert eps G), t_k = k/nsteps, G ~ N(0,1)^{3x3};
    the first `n_exact` points have G = 0, i.e. exactly the path of tests/test_FeFp_jax.py:28-30
    (F = diag(1 + eps t, 1 - eps t/2, 1 - eps t/2)).  Returns a list of (n,9) arrays."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, 3, 3))
    G[: min(n_exact, n)] = 0.0
    out = []
    for k in range(1, nsteps + 1):
        t = k / nsteps
        F = np.zeros((n, 3, 3))
        F[:] = np.eye(3)
        F += t * (eps * np.diag([1.0, -0.5, -0.5]) + pert * eps * G)
        out.append(onp.tensor_to_nsym(F))
    return out
