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


KUHN = [(0, 1, 2, 6), (0, 2, 3, 6), (0, 3, 7, 6), (0, 7, 4, 6), (0, 4, 5, 6), (0, 5, 1, 6)]   # hex8 -> 6 tets along 0-6


def simplex_host_gradient(coords, geom_conn, dofmap, u, dphi):
    """Plain numpy displacement gradient H (ncells, nqp, 3, 3) of a Lagrange field on straight-sided simplices (the
    checker of dxm_mesh_create_simplex): H = sum_m u_m (x) A^-T dphi[q, m], zero-padded to 3x3 for triangles."""
    tdim = geom_conn.shape[1] - 1
    X = coords[geom_conn][:, :, :tdim]                       # (c, tdim+1, tdim)
    A = (X[:, 1:] - X[:, :1]).transpose(0, 2, 1)             # dX_a / dxi_d
    Ai = np.linalg.inv(A)                                    # dxi_d / dX_a
    U = u.reshape(-1, tdim)[dofmap]                          # (c, nd, tdim)
    g = np.einsum("qmd,cda->cqma", dphi, Ai)                 # dN_m / dX_a at point q
    H = np.zeros((len(geom_conn), dphi.shape[0], 3, 3))
    H[:, :, :tdim, :tdim] = np.einsum("cmi,cqma->cqia", U, g)
    return H


def mandel_strain(H):
    """(…, 3, 3) displacement gradients -> (…, 6) Mandel strains [utils.py:146-165]."""
    e = 0.5 * (H + np.swapaxes(H, -1, -2))
    r = np.sqrt(2.0)
    return np.stack([e[..., 0, 0], e[..., 1, 1], e[..., 2, 2], r * e[..., 0, 1], r * e[..., 0, 2], r * e[..., 1, 2]], axis=-1)


def deformation_gradient9(H):
    """(…, 3, 3) displacement gradients -> (…, 9) F = 1 + H in the order [11,22,33,12,21,13,31,23,32] (utils.py:168-190)."""
    F = H + np.eye(3)
    return np.stack([F[..., 0, 0], F[..., 1, 1], F[..., 2, 2], F[..., 0, 1], F[..., 1, 0], F[..., 0, 2], F[..., 2, 0],
                     F[..., 1, 2], F[..., 2, 1]], axis=-1)


def triangle_grid(n, distort=0.2, seed=0):
    """Unit square cut into 2 n^2 triangles with interior vertices moved randomly: (coords (nv,2), cells (nc,3))."""
    xs = np.linspace(0.0, 1.0, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    rng = np.random.default_rng(seed)
    inner = (coords > 1e-12).all(axis=1) & (coords < 1 - 1e-12).all(axis=1)
    coords[inner] += distort / n * rng.uniform(-1, 1, (int(inner.sum()), 2))
    idx = lambda i, j: i * (n + 1) + j   # noqa: E731
    cells = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            cells += [(a, b, c), (a, c, d)]
    return coords, np.array(cells, dtype=np.int32)


# ---- host <-> device copies of the TESTS themselves ---------------------------------------------------------------
# torch's `tensor.cpu()` / `torch.from_numpy(a).to(device)` hand pageable memory to the GPU runtime, whose cache of
# on-the-fly page-locked ranges goes stale when numpy / torch recycle host addresses: about one suite run in 25 died
# with "Memory access fault by GPU ... Write access to a read-only page" inside such a call.
# The tests therefore stage through page-locked tensors, like the library does.
def to_device(a, dev="cuda:0"):
    """numpy array -> fp64 / int tensor on the device through a page-locked staging tensor."""
    import torch

    a = np.ascontiguousarray(a)
    pin = torch.empty(a.shape, dtype=torch.from_numpy(a[:0] if a.ndim else a.reshape(1)[:0]).dtype, pin_memory=True)
    pin.numpy()[...] = a
    out = pin.to(dev)
    torch.cuda.synchronize()
    return out


def to_host(t):
    """device tensor -> numpy array through a page-locked staging tensor."""
    import torch

    pin = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pin.copy_(t)
    torch.cuda.synchronize()
    return pin.numpy().copy()


def fefp_uniaxial_known_answer(step, advance, R, n=257, nsteps=8, stretch=(1.004, 1.08), sig0=SIG0_F):
    """Material-point known answer for the finite-strain law that does not go through the oracle's formulas: ``n`` points, each on
    its own UNIAXIAL KIRCHHOFF-STRESS path.  ``F = diag(lam, lt, lt)`` with ``lam`` ramped to ``stretch[0] ... stretch[1]`` in
    ``nsteps`` increments; the lateral stretch ``lt`` is found per point by Newton on ``P_yy = 0`` with the update's own 9x9
    tangent (``dP_yy / d lt = Ct[yy, yy] + Ct[yy, zz]``: several updates from one initial state, then ``advance`` -- the cadence of
    a global Newton loop).  ``step(F9) -> (P (n,9), Ct (n,9,9), p (n,), be_bar (n,6))``.

    What must hold after every increment, whatever the integration algorithm (model: ``tau = kappa/2 (J^2 - 1) 1 + mu dev(be_bar)``,
    J2 yield on Kirchhoff stress with ``R(p)``, isochoric ``be_bar``; DESIGN.md "FeFp"):
      * ``tau = P F^T`` is uniaxial; at a plastic point ``tau_xx = R(p)`` (tests/test_FeFp_jax.py:14-15 for Voce);
      * ``det(be_bar) = 1`` and ``be_bar = diag(a, a^-1/2, a^-1/2)`` with ``mu (a - a^-1/2) = tau_xx``;
      * the volumetric relation ``kappa/2 (J^2 - 1) = tau_xx / 3`` for ``J = lam lt^2``;
      * an elastic point has ``a = J^(-2/3) lam^2`` (no plastic stretch yet).
    Returns (number of plastic points at the end, largest p)."""
    lmbda, mu = onp.lame(E, NU)
    kappa = lmbda + 2.0 * mu / 3.0
    lam_end = np.linspace(stretch[0], stretch[1], n)
    lt = np.ones(n)
    p_prev = np.zeros(n)
    for t in np.linspace(0.0, 1.0, nsteps + 1)[1:]:
        lam = 1.0 + t * (lam_end - 1.0)
        for it in range(60):
            F9 = np.zeros((n, 9))
            F9[:, 0], F9[:, 1], F9[:, 2] = lam, lt, lt
            P, Ct, p, be = step(F9)
            pyy = P[:, 1].copy()
            if np.abs(pyy).max() < 1e-11 * sig0:
                break
            lt = lt - pyy / (Ct[:, 1, 1] + Ct[:, 1, 2])
        assert np.abs(pyy).max() < 1e-11 * sig0 and it < 15, (t, it, np.abs(pyy).max())
        txx = P[:, 0] * lam                                      # tau = P F^T, F diagonal
        assert np.abs(P[:, 2]).max() < 1e-10 * sig0 and np.abs(P[:, 3:]).max() < 1e-10 * sig0
        plastic = p > p_prev + 1e-15
        assert np.all(p >= p_prev - 1e-16)
        # yield consistency at the plastic points, elastic closed form at the others
        assert np.abs(txx[plastic] - R(p[plastic])).max(initial=0.0) < 1e-9 * sig0
        assert np.all(txx[~plastic] <= R(p[~plastic]) + 1e-9 * sig0)
        J = lam * lt * lt
        # be_bar: isochoric, uniaxial, and carrying exactly the deviatoric stress
        a = np.ones(n)
        for _ in range(60):                                      # mu (a - a^-1/2) = tau_xx, Newton from a = 1 (monotone)
            a = a - (mu * (a - a ** -0.5) - txx) / (mu * (1.0 + 0.5 * a ** -1.5))
        assert np.abs(mu * (a - a ** -0.5) - txx).max() < 1e-12 * sig0
        det_be = be[:, 0] * be[:, 1] * be[:, 2]                  # (the shear entries vanish on this path)
        assert np.abs(be[:, 3:]).max() < 1e-13 and np.abs(det_be - 1.0).max() < 1e-12
        assert np.abs(be[:, 0] - a).max() < 1e-9 and np.abs(be[:, 1] - a ** -0.5).max() < 1e-9 and np.abs(be[:, 2] - be[:, 1]).max() < 1e-13
        assert np.abs(0.5 * kappa * (J * J - 1.0) - txx / 3.0).max() < 1e-9 * sig0
        el = p == 0.0
        assert np.abs(a[el] - J[el] ** (-2.0 / 3.0) * lam[el] ** 2).max(initial=0.0) < 1e-9
        advance()
        p_prev = p.copy()
    return int((p_prev > 0).sum()), float(p_prev.max())
