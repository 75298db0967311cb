"""Second, independently formulated CPU oracles for the finite-strain FeFp J2 law, in torch (fp64).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED against the real ``jaxmat`` (absent
here): what these functions do is BOUND that risk -- they restate the model in the form the reference's
dependency is built (the prose of
``demos/jax/finite_strain_elastoplasticity/finite_strain_elastoplasticity.py:165-181``), with none of the
algebra of ``oracle/constitutive_np.py::fefp_update`` / the HIP kernel (no radial reduction to two scalars,
no hidden ``Cp^-1`` state, no closed-form tangent):

``fefp_fb7``
    state ``(F_n, be_bar_n, p_n)`` as in jaxmat (``jaxmat.py:166-186``: the state carries the gradient); relative
    deformation gradient ``f = F F_n^-1``, ``f_bar = det(f)^(-1/3) f``, ``be_bar_trial = f_bar be_bar_n f_bar^T``;
    SEVEN unknowns ``(dp, be_bar)`` and the residual
        ``FB(-f_yield, dp) = 0``,   ``FB(a, b) = a + b - sqrt(a^2 + b^2)``  (Fischer-Burmeister complementarity)
        ``dev(be_bar - be_bar_trial) + 2 dp tr(be_bar)/3 n + (det(be_bar) - 1) 1 = 0``,  ``n = d sigma_eq / d s``
    with ``s = mu dev(be_bar)``, ``f_yield = sigma_eq(s) - R(p_n + dp)``; Newton on the 7x7 system with the
    Jacobian by forward-mode AD.  ``tau = s + kappa/2 (J^2 - 1) 1``, ``P = tau F^-T``.
    The tangent is ``jacfwd`` of the whole update w.r.t. ``F`` exactly as the reference builds it
    (``jaxmat.py:147-151``: ``vmap(jacfwd(constitutive_update, argnums=0, has_aux=True))``), the root being
    differentiated by the implicit-function theorem (what ``optimistix.root_find`` does under ``jacfwd``).

``fefp_simo``
    Simo & Hughes (1998) Box 9.1 / Simo (1992), the textbook the reference cites (``docs/references.bib``):
    the radial return that PRESERVES ``tr(be_bar)`` (``det(be_bar) = 1`` then holds only approximately).  A
    different discretisation of the same model: its distance to the ``det = 1`` update is the size of the
    modelling choice, reported in DESIGN.md section 5.
"""
from __future__ import annotations

import math

import numpy as np
import torch

torch.set_default_dtype(torch.float64)
SQ2 = math.sqrt(2.0)
NSYM_IDX = ((0, 0), (1, 1), (2, 2), (0, 1), (1, 0), (0, 2), (2, 0), (1, 2), (2, 1))   # utils.py:168-190


def nsym_to_tensor(v):
    rows = [[None] * 3 for _ in range(3)]
    for k, (i, j) in enumerate(NSYM_IDX):
        rows[i][j] = v[k]
    return torch.stack([torch.stack(r) for r in rows])


def tensor_to_nsym(T):
    return torch.stack([T[i, j] for i, j in NSYM_IDX])


def mandel_to_tensor(v):
    a, b, c = v[3] / SQ2, v[4] / SQ2, v[5] / SQ2
    return torch.stack([torch.stack([v[0], a, b]), torch.stack([a, v[1], c]), torch.stack([b, c, v[2]])])


def tensor_to_mandel(T):
    return torch.stack([T[0, 0], T[1, 1], T[2, 2], SQ2 * T[0, 1], SQ2 * T[0, 2], SQ2 * T[1, 2]])


def dev(T):
    return T - (T[0, 0] + T[1, 1] + T[2, 2]) / 3.0 * torch.eye(3)


def det3(A):
    """Closed-form 3x3 determinant / inverse: plain arithmetic, so that forward-mode AD under vmap never goes
    through batched LU kernels (torch 2.10 returns NaN tangents for some batch rows there)."""
    return (A[0, 0] * (A[1, 1] * A[2, 2] - A[1, 2] * A[2, 1]) - A[0, 1] * (A[1, 0] * A[2, 2] - A[1, 2] * A[2, 0])
            + A[0, 2] * (A[1, 0] * A[2, 1] - A[1, 1] * A[2, 0]))


def inv3(A):
    c = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(3):
            i1, i2, j1, j2 = (i + 1) % 3, (i + 2) % 3, (j + 1) % 3, (j + 2) % 3
            c[j][i] = A[i1, j1] * A[i2, j2] - A[i1, j2] * A[i2, j1]   # transposed cofactor
    return torch.stack([torch.stack(r) for r in c]) / det3(A)


def tr3(T):
    return T[0, 0] + T[1, 1] + T[2, 2]


class Voce:
    def __init__(self, sig0, sigu, b):
        self.sig0, self.sigu, self.b = sig0, sigu, b

    def __call__(self, p):
        return self.sig0 + (self.sigu - self.sig0) * (1.0 - torch.exp(-self.b * p))


class Linear:
    def __init__(self, sig0, H):
        self.sig0, self.H = sig0, H

    def __call__(self, p):
        return self.sig0 + self.H * p


def _lame(E, nu):
    lmbda = E * nu / (1 + nu) / (1 - 2 * nu)
    mu = E / 2 / (1 + nu)
    return lmbda, mu, lmbda + 2 * mu / 3


def _sigma_eq(s):
    return torch.sqrt(1.5 * torch.sum(s * s) + 1e-300)


def _fb(a, b):
    return a + b - torch.sqrt(a * a + b * b)


# ----------------------------------------------------------------------------------------------
# (A) seven-unknown Fischer-Burmeister form
# ----------------------------------------------------------------------------------------------
def _trial(F9, Fn9, ben6):
    F, Fn = nsym_to_tensor(F9), nsym_to_tensor(Fn9)
    f = F @ inv3(Fn)
    fbar = f * det3(f) ** (-1.0 / 3.0)
    return fbar @ mandel_to_tensor(ben6) @ fbar.T


def _residual(x, F9, Fn9, ben6, pn, mu, hard):
    dp, be = x[0], mandel_to_tensor(x[1:])
    be_tr = _trial(F9, Fn9, ben6)
    s = mu * dev(be)
    seq = _sigma_eq(s)
    fy = seq - hard(pn + dp)
    n = 1.5 * s / seq
    r_be = dev(be - be_tr) + 2.0 * dp * tr3(be) / 3.0 * n + (det3(be) - 1.0) * torch.eye(3)
    # the yield residual is scaled to the size of the others (stress -> strain units); a root is a root
    return torch.cat([(_fb(-fy, mu * dp) / mu).reshape(1), tensor_to_mandel(r_be)])


def _pk1(x, F9, mu, kappa):
    F = nsym_to_tensor(F9)
    J = det3(F)
    tau = mu * dev(mandel_to_tensor(x[1:])) + 0.5 * kappa * (J * J - 1.0) * torch.eye(3)
    return tensor_to_nsym(tau @ inv3(F).T)


def fefp_fb7(F9, Fn9, be_n, p_n, E, nu, hard, maxit=60, tol=1e-13, tangent=True):
    """(N,9) F, (N,9) F_n, (N,6) be_bar_n Mandel, (N,) p_n  ->  dict(P (N,9), be_bar (N,6), p (N,), Ct (N,9,9), iters)."""
    F9, Fn9, be_n, p_n = (torch.as_tensor(np.asarray(a, dtype=np.float64)) for a in (F9, Fn9, be_n, p_n))
    N = F9.shape[0]
    _, mu, kappa = _lame(E, nu)
    res = lambda x, F, Fn, b, p: _residual(x, F, Fn, b, p, mu, hard)   # noqa: E731
    vres = torch.func.vmap(res)
    vjac = torch.func.vmap(torch.func.jacfwd(res, argnums=0))
    x = torch.cat([torch.zeros(N, 1), torch.func.vmap(lambda F, Fn, b: tensor_to_mandel(_trial(F, Fn, b)))(F9, Fn9, be_n)], dim=1)
    iters = 0
    for it in range(maxit):
        r = vres(x, F9, Fn9, be_n, p_n)
        nrm = r.abs().amax(dim=1)
        if float(nrm.max()) < tol:
            break
        step = torch.linalg.solve(vjac(x, F9, Fn9, be_n, p_n), -r.unsqueeze(-1)).squeeze(-1)
        # the Fischer-Burmeister function has a kink at the origin: damp steps that leave dp >= 0
        t = torch.ones(N)
        neg = (x[:, 0] + step[:, 0]) < 0
        t[neg] = torch.clamp(0.9 * x[neg, 0] / (-step[neg, 0]).clamp_min(1e-300), min=0.1)
        x = torch.where((nrm > tol).unsqueeze(1), x + t.unsqueeze(1) * step, x)
        iters = it + 1
    out = dict(P=torch.func.vmap(lambda x_, F: _pk1(x_, F, mu, kappa))(x, F9).numpy(), be_bar=x[:, 1:].numpy().copy(),
               p=(p_n + x[:, 0]).numpy(), dp=x[:, 0].numpy().copy(), iters=iters,
               residual=float(vres(x, F9, Fn9, be_n, p_n).abs().max()))
    if tangent:
        # jacfwd of the update w.r.t. F with the root differentiated implicitly:
        #   dP/dF = d_F P + d_x P . dx/dF,   dx/dF = -(d_x r)^-1 d_F r
        pk = lambda xx, FF: _pk1(xx, FF, mu, kappa)   # noqa: E731
        Jx = vjac(x, F9, Fn9, be_n, p_n)
        JF = torch.func.vmap(torch.func.jacfwd(res, argnums=1))(x, F9, Fn9, be_n, p_n)
        dx = -torch.linalg.solve(Jx, JF)
        out["Ct"] = (torch.func.vmap(torch.func.jacfwd(pk, argnums=1))(x, F9)
                     + torch.func.vmap(torch.func.jacfwd(pk, argnums=0))(x, F9) @ dx).numpy()
    return out


# ----------------------------------------------------------------------------------------------
# (B) Simo & Hughes Box 9.1: trace-preserving radial return
# ----------------------------------------------------------------------------------------------
def _simo_point(F9, Fn9, ben6, pn, dgam, mu, kappa):
    """The update for a GIVEN consistency parameter dgam (0 = elastic); differentiable in F and dgam."""
    F = nsym_to_tensor(F9)
    be_tr = _trial(F9, Fn9, ben6)
    Ibar = tr3(be_tr) / 3.0
    s_tr = mu * dev(be_tr)
    nrm = torch.sqrt(torch.sum(s_tr * s_tr) + 1e-300)
    n = s_tr / nrm
    s = s_tr - 2.0 * mu * Ibar * dgam * n
    J = det3(F)
    tau = s + 0.5 * kappa * (J * J - 1.0) * torch.eye(3)
    be = s / mu + Ibar * torch.eye(3)
    return tensor_to_nsym(tau @ inv3(F).T), tensor_to_mandel(be), nrm, Ibar


def fefp_simo(F9, Fn9, be_n, p_n, E, nu, hard, maxit=60, tol=1e-14, tangent=True):
    F9, Fn9, be_n, p_n = (torch.as_tensor(np.asarray(a, dtype=np.float64)) for a in (F9, Fn9, be_n, p_n))
    N = F9.shape[0]
    _, mu, kappa = _lame(E, nu)
    s23 = math.sqrt(2.0 / 3.0)
    _, _, nrm, Ibar = torch.func.vmap(lambda F, Fn, b, p: _simo_point(F, Fn, b, p, torch.zeros(()), mu, kappa))(F9, Fn9, be_n, p_n)
    plastic = nrm - s23 * hard(p_n) > 0
    # scalar consistency equation  g(dgam) = |s_tr| - sqrt(2/3) R(p_n + sqrt(2/3) dgam) - 2 mu Ibar dgam = 0
    g = lambda dg, nrm_, Ib, p: nrm_ - s23 * hard(p + s23 * dg) - 2.0 * mu * Ib * dg   # noqa: E731
    dg = torch.zeros(N)
    for _ in range(maxit):
        val = torch.where(plastic, g(dg, nrm, Ibar, p_n), torch.zeros(N))
        if float(val.abs().max()) < tol * hard.sig0:
            break
        d = torch.func.vmap(torch.func.grad(g))(dg, nrm, Ibar, p_n)
        dg = dg - val / d
    out_P, out_be, _, _ = torch.func.vmap(lambda F, Fn, b, p, d_: _simo_point(F, Fn, b, p, d_, mu, kappa))(F9, Fn9, be_n, p_n, dg)
    out = dict(P=out_P.numpy(), be_bar=out_be.numpy(), p=(p_n + s23 * dg).numpy(), plastic=plastic.numpy())
    if tangent:
        def point(F, Fn, b, p, d_, pl):
            pk = lambda FF, dd: _simo_point(FF, Fn, b, p, dd, mu, kappa)[0]   # noqa: E731
            gg = lambda FF, dd: (lambda o: g(dd, o[2], o[3], p))(_simo_point(FF, Fn, b, p, dd, mu, kappa))   # noqa: E731
            dP_dF = torch.func.jacfwd(pk, argnums=0)(F, d_)
            dP_dd = torch.func.jacfwd(pk, argnums=1)(F, d_)
            ddg = -torch.func.jacfwd(gg, argnums=0)(F, d_) / torch.func.grad(gg, argnums=1)(F, d_)
            return dP_dF + pl * torch.outer(dP_dd, ddg)

        out["Ct"] = torch.func.vmap(point)(F9, Fn9, be_n, p_n, dg, plastic.to(torch.float64)).numpy()
    return out


# ----------------------------------------------------------------------------------------------
# (C) small-strain J2 in the stress-state form recalled from jaxmat: the state is
#     (sigma_n, eps_n, p_n), sigma_trial = sigma_n + C : (eps - eps_n), one unknown dp with a
#     Fischer-Burmeister residual, tangent = jacfwd of the update with the root differentiated implicitly.
#     Independent of oracle/constitutive_np.py::j2_update, which integrates (eps_p, p) in the elastic-strain
#     form of tests/mfront/IsotropicLinearHardeningPlasticity.mfront:49-77 with a closed-form tangent.
# ----------------------------------------------------------------------------------------------
def _j2_stress(dp, eps, eps_n, sig_n, lmbda, mu):
    one = torch.tensor([1.0, 1.0, 1.0, 0.0, 0.0, 0.0])
    de = eps - eps_n
    sig_tr = sig_n + lmbda * torch.sum(de[:3]) * one + 2.0 * mu * de
    s_tr = sig_tr - torch.sum(sig_tr[:3]) / 3.0 * one
    seq = torch.sqrt(1.5 * torch.sum(s_tr * s_tr) + 1e-300)
    return sig_tr - 3.0 * mu * dp * s_tr / seq, seq


def j2_fb(eps, eps_n, sig_n, p_n, E, nu, hard, maxit=60, tol=1e-13):
    eps, eps_n, sig_n, p_n = (torch.as_tensor(np.asarray(a, dtype=np.float64)) for a in (eps, eps_n, sig_n, p_n))
    lmbda, mu, _ = _lame(E, nu)

    def res(dp, e, en, sn, pn):
        _, seq = _j2_stress(dp, e, en, sn, lmbda, mu)
        return _fb(-(seq - 3.0 * mu * dp - hard(pn + dp)), mu * dp) / mu

    vres, vd = torch.func.vmap(res), torch.func.vmap(torch.func.grad(res, argnums=0))
    dp = torch.zeros(eps.shape[0])
    for _ in range(maxit):
        r = vres(dp, eps, eps_n, sig_n, p_n)
        if float(r.abs().max()) < tol:
            break
        dp = torch.clamp(dp - r / vd(dp, eps, eps_n, sig_n, p_n), min=0.0)
    sig = torch.func.vmap(lambda d, e, en, sn: _j2_stress(d, e, en, sn, lmbda, mu)[0])(dp, eps, eps_n, sig_n)
    st = lambda d, e, en, sn: _j2_stress(d, e, en, sn, lmbda, mu)[0]   # noqa: E731
    d_e = torch.func.vmap(torch.func.jacfwd(st, argnums=1))(dp, eps, eps_n, sig_n)
    d_d = torch.func.vmap(torch.func.jacfwd(st, argnums=0))(dp, eps, eps_n, sig_n)
    ddp = -torch.func.vmap(torch.func.jacfwd(res, argnums=1))(dp, eps, eps_n, sig_n, p_n) / vd(dp, eps, eps_n, sig_n, p_n).unsqueeze(1)
    plastic = dp > 0
    Ct = d_e + plastic.to(torch.float64)[:, None, None] * d_d.unsqueeze(2) * ddp.unsqueeze(1)
    return dict(sig=sig.numpy(), p=(p_n + dp).numpy(), Ct=Ct.numpy(), plastic=plastic.numpy())
