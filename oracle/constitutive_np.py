"""Vectorised numpy restatement of the per-Gauss-point constitutive updates.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Conventions (reference ``dolfinx_materials/utils.py:146-212``, ``docs/intro.md:134-175``):
symmetric 2nd-order tensors are Mandel 6-vectors ``[a11,a22,a33,√2 a12,√2 a13,√2 a23]``;
non-symmetric ones are 9-vectors ``[a11,a22,a33,a12,a21,a13,a31,a23,a32]``; the
tangent is ``Ct[i,j] = d flux_i / d gradient_j`` of the *discrete algorithm* at
fixed old state (reference ``dolfinx_materials/jaxmat.py:147-151``).

Every function maps ``(N, dim)`` batches to ``(N, dim)`` batches, like
``batched_constitutive_update`` in the reference (``generic.py:115-117``,
``jaxmat.py:147-155``).
"""
from __future__ import annotations

import numpy as np

SQ2 = np.sqrt(2.0)

# ----------------------------------------------------------------------------------------
# conventions
# ----------------------------------------------------------------------------------------
#: position (row, col) of each entry of the non-symmetric 9-vector (utils.py:168-190)
NSYM_IDX = ((0, 0), (1, 1), (2, 2), (0, 1), (1, 0), (0, 2), (2, 0), (1, 2), (2, 1))


def mandel_to_tensor(v):
    """(N,6) Mandel -> (N,3,3) symmetric (utils.py:193-212 for the (6,) case)."""
    v = np.asarray(v, dtype=np.float64)
    T = np.empty(v.shape[:-1] + (3, 3))
    T[..., 0, 0] = v[..., 0]
    T[..., 1, 1] = v[..., 1]
    T[..., 2, 2] = v[..., 2]
    T[..., 0, 1] = T[..., 1, 0] = v[..., 3] / SQ2
    T[..., 0, 2] = T[..., 2, 0] = v[..., 4] / SQ2
    T[..., 1, 2] = T[..., 2, 1] = v[..., 5] / SQ2
    return T


def tensor_to_mandel(T):
    """(N,3,3) symmetric -> (N,6) Mandel (utils.py:146-165)."""
    T = np.asarray(T, dtype=np.float64)
    v = np.empty(T.shape[:-2] + (6,))
    v[..., 0] = T[..., 0, 0]
    v[..., 1] = T[..., 1, 1]
    v[..., 2] = T[..., 2, 2]
    v[..., 3] = SQ2 * T[..., 0, 1]
    v[..., 4] = SQ2 * T[..., 0, 2]
    v[..., 5] = SQ2 * T[..., 1, 2]
    return v


def nsym_to_tensor(v):
    """(N,9) -> (N,3,3) (utils.py:208-210)."""
    v = np.asarray(v, dtype=np.float64)
    T = np.empty(v.shape[:-1] + (3, 3))
    for k, (i, j) in enumerate(NSYM_IDX):
        T[..., i, j] = v[..., k]
    return T


def tensor_to_nsym(T):
    """(N,3,3) -> (N,9) (utils.py:168-190)."""
    T = np.asarray(T, dtype=np.float64)
    v = np.empty(T.shape[:-2] + (9,))
    for k, (i, j) in enumerate(NSYM_IDX):
        v[..., k] = T[..., i, j]
    return v


def lame(E, nu):
    """python_materials/elasticity.py:12-13."""
    return E * nu / (1 + nu) / (1 - 2 * nu), E / 2 / (1 + nu)


def elastic_matrix(E, nu):
    """python_materials/elasticity.py:15-19: C = 2 mu I6 ; C[:3,:3] += lambda."""
    lmbda, mu = lame(E, nu)
    C = 2 * mu * np.eye(6)
    C[:3, :3] += lmbda
    return C


# ----------------------------------------------------------------------------------------
# law 0: isotropic linear elasticity (python_materials/elasticity.py:21-24)
# ----------------------------------------------------------------------------------------
def elastic_iso(eps, E, nu):
    """sigma = C eps ; Ct = C for every point.  Returns (sig (N,6), Ct (N,6,6))."""
    eps = np.asarray(eps, dtype=np.float64)
    C = elastic_matrix(E, nu)
    sig = eps @ C.T
    Ct = np.broadcast_to(C, (eps.shape[0], 6, 6)).copy()
    return sig, Ct


# ----------------------------------------------------------------------------------------
# hardening laws R(p) and R'(p)
# ----------------------------------------------------------------------------------------
class LinearHardening:
    """R(p) = s0 + H p  (tests/mfront/IsotropicPlasticMisesFlow.mfront:7-11)."""

    def __init__(self, sig0, H):
        self.sig0, self.H = float(sig0), float(H)

    def R(self, p):
        return self.sig0 + self.H * p

    def dR(self, p):
        return np.full_like(np.asarray(p, dtype=np.float64), self.H)


class VoceHardening:
    """R(p) = s0 + (su - s0)(1 - exp(-b p))  (tests/test_FeFp_jax.py:14-15)."""

    def __init__(self, sig0, sigu, b):
        self.sig0, self.sigu, self.b = float(sig0), float(sigu), float(b)

    def R(self, p):
        return self.sig0 + (self.sigu - self.sig0) * (1.0 - np.exp(-self.b * p))

    def dR(self, p):
        return (self.sigu - self.sig0) * self.b * np.exp(-self.b * p)


#: Newton controls shared by the oracle, the C restatement and the HIP kernels
NEWTON_MAXIT = 25
NEWTON_RTOL = 1e-14  # |r| <= rtol * max(stress_scale(sig0, mu), trial stress)


def stress_scale(sig0, mu):
    """Stress the local Newton tolerance is relative to: the initial yield stress, floored at 2e-8 mu so that a law with
    R(0) = 0 (hardening from zero) still has a reachable tolerance (dxmat.hip::build_params)."""
    return max(abs(float(sig0)), 2e-8 * float(mu))


def _solve_dp(seq, p_n, mu, hard, maxit=NEWTON_MAXIT, rtol=NEWTON_RTOL):
    """Scalar return-mapping equation r(dp) = seq - 3 mu dp - R(p_n+dp) = 0 on the
    plastic subset; monotone, started from dp = 0.  For linear hardening
    the first Newton step is the closed form of
    IsotropicLinearHardeningPlasticity.mfront:63-64."""
    dp = np.zeros_like(seq)
    iters = np.zeros(seq.shape, dtype=np.int32)
    if isinstance(hard, LinearHardening):
        dp = (seq - hard.sig0 - hard.H * p_n) / (hard.H + 3 * mu)
        return dp, iters
    active = np.ones(seq.shape, dtype=bool)
    for _ in range(maxit):
        r = seq - 3 * mu * dp - hard.R(p_n + dp)
        active = np.abs(r) > np.maximum(rtol * stress_scale(hard.sig0, mu), rtol * seq)
        if not active.any():
            break
        dr = -3 * mu - hard.dR(p_n + dp)
        dp = np.where(active, dp - r / dr, dp)
        iters += active
    return dp, iters


# ----------------------------------------------------------------------------------------
# laws 1/2: small-strain J2 plasticity, isotropic hardening, implicit Euler radial return
# (tests/mfront/IsotropicLinearHardeningPlasticity.mfront:49-77, (eps_p, p)-state form)
# ----------------------------------------------------------------------------------------
def j2_update(eps, epsp_n, p_n, E, nu, hard):
    """Returns dict(sig (N,6), epsp (N,6), p (N,), Ct (N,6,6), plastic (N,) bool,
    iters (N,) int, f_trial (N,)) and what ``Ct = c1 1x1 + c2 I + c3 n x n`` is made of: coef (N,3) = (c1, c2, c3), n (N,6) the
    flow direction (zero at elastic points) and w (N,) with n = dev(sig) w (the return is radial: dev(sig) = (1 - 3 mu dp / seq)
    s_trial, hence w = 3/2 / (seq - 3 mu dp); zero at elastic points)."""
    eps = np.asarray(eps, dtype=np.float64)
    epsp_n = np.asarray(epsp_n, dtype=np.float64)
    p_n = np.asarray(p_n, dtype=np.float64).reshape(-1)
    N = eps.shape[0]
    lmbda, mu = lame(E, nu)
    one = np.array([1.0, 1.0, 1.0, 0.0, 0.0, 0.0])

    eel = eps - epsp_n  # trial elastic strain  (mfront:52 `eel += deto`)
    tr = eel[:, 0] + eel[:, 1] + eel[:, 2]
    se = 2 * mu * (eel - tr[:, None] / 3.0 * one)  # mfront:53
    seq = np.sqrt(1.5 * np.einsum("ni,ni->n", se, se))  # mfront:54 sigmaeq
    f_trial = seq - hard.R(p_n)  # mfront:55
    plastic = f_trial > 0.0

    dp = np.zeros(N)
    iters = np.zeros(N, dtype=np.int32)
    n = np.zeros((N, 6))
    c1 = np.full(N, lmbda)
    c2 = np.full(N, 2 * mu)
    c3 = np.zeros(N)
    if plastic.any():
        idx = np.nonzero(plastic)[0]
        dp_i, it_i = _solve_dp(seq[idx], p_n[idx], mu, hard)
        dp[idx] = dp_i
        iters[idx] = it_i
        n[idx] = 1.5 * se[idx] / seq[idx, None]  # mfront:61
        beta = dp_i / seq[idx]  # dp * iseq_e
        gamma = 1.0 / (hard.dR(p_n[idx] + dp_i) + 3 * mu)  # cste (H -> R'(p_{n+1}))
        # Dt = lambda IxI + 2mu Id - 4mu^2 [beta (M - n^n) + gamma n^n], M = 3/2 Id - 1/2 IxI
        c1[idx] = lmbda + 2 * mu * mu * beta
        c2[idx] = 2 * mu - 6 * mu * mu * beta
        c3[idx] = 4 * mu * mu * (beta - gamma)

    epsp = epsp_n + dp[:, None] * n
    p = p_n + dp
    eel = eel - dp[:, None] * n  # mfront:65
    tr = eel[:, 0] + eel[:, 1] + eel[:, 2]
    sig = lmbda * tr[:, None] * one + 2 * mu * eel  # mfront:76

    Ct = (
        c1[:, None, None] * np.outer(one, one)[None]
        + c2[:, None, None] * np.eye(6)[None]
        + c3[:, None, None] * n[:, :, None] * n[:, None, :]
    )
    w = np.zeros(N)
    if plastic.any():
        w[idx] = 1.5 / (seq[idx] - 3 * mu * dp[idx])
    return dict(sig=sig, epsp=epsp, p=p, Ct=Ct, plastic=plastic, iters=iters, f_trial=f_trial, coef=np.stack([c1, c2, c3], axis=1), n=n, w=w)


def j2_update_mfront_form(deto, eel_n, p_n, E, nu, H, s0):
    """Literal transcription of the @Integrator block of
    tests/mfront/IsotropicLinearHardeningPlasticity.mfront:49-77 (elastic-strain state,
    strain *increment* input, explicit IxI/Id/M 4th-order tensors).  Used only to pin
    ``j2_update`` (which is the (eps_p,p)-state, coefficient form) against the in-tree spec."""
    lmbda, mu = lame(E, nu)
    IxI = np.zeros((6, 6))
    IxI[:3, :3] = 1.0
    Id = np.eye(6)
    M = 1.5 * (Id - IxI / 3.0)
    N = deto.shape[0]
    eel = eel_n + deto
    p = np.array(p_n, dtype=np.float64).reshape(-1).copy()
    sig = np.empty((N, 6))
    Dt = np.empty((N, 6, 6))
    for i in range(N):
        e = eel[i]
        dev = e - (e[0] + e[1] + e[2]) / 3.0 * np.array([1, 1, 1, 0, 0, 0.0])
        se = 2 * mu * dev
        seq_e = np.sqrt(1.5 * se @ se)
        if seq_e - s0 - H * p[i] > 0:
            iseq_e = 1 / seq_e
            n = 3 * se / (2 * seq_e)
            cste = 1 / (H + 3 * mu)
            dp = (seq_e - s0 - H * p[i]) * cste
            e = e - dp * n
            p[i] += dp
            Dt[i] = lmbda * IxI + 2 * mu * Id - 4 * mu * mu * (
                dp * iseq_e * (M - np.outer(n, n)) + cste * np.outer(n, n)
            )
        else:
            Dt[i] = lmbda * IxI + 2 * mu * Id
        eel[i] = e
        sig[i] = lmbda * (e[0] + e[1] + e[2]) * np.array([1, 1, 1, 0, 0, 0.0]) + 2 * mu * e
    return sig, eel, p, Dt


# ----------------------------------------------------------------------------------------
# law 3: finite-strain FeFp J2 plasticity (gradient F, flux PK1)
# ----------------------------------------------------------------------------------------
# The reference only fixes the interface of this law (gradient "F" (9), flux "PK1" (9), ISVs p
# and be_bar initialised to the identity: jaxmat.py:170-186,
# demos/jax/finite_strain_elastoplasticity/finite_strain_elastoplasticity.py:165-181, driver and
# parameters tests/test_FeFp_jax.py:7-31); the arithmetic lives in the absent third-party
# jaxmat package.  PARITY UNPINNED.  The algorithm below is the build's own documented choice
# (DESIGN.md "FeFp"): Simo's multiplicative J2 model (Simo & Hughes 1998, ch. 9; Simo 1992)
# with the isochoric elastic left Cauchy-Green tensor be_bar as internal variable,
#   psi  = kappa/2 (1/2 (J^2-1) - ln J) + mu/2 (tr be_bar - 3)
#   tau  = kappa/2 (J^2-1) 1 + mu dev(be_bar),        P = tau F^-T
#   f    = sqrt(3/2) |dev tau| - R(p),  R = Voce
#   flow : dev(be_bar) = dev(be_bar_trial) - 2 dp (tr be_bar / 3) n,  n = sqrt(3/2) s/|s|
#          det(be_bar) = 1   (exactly isochoric update instead of Simo's tr-preserving one)
# with be_bar_trial = J^(-2/3) F Cp_bar^-1 F^T.  Because the return is radial in the direction
# s_hat of dev(be_bar_trial), the 7 unknowns (dp, be_bar) reduce to two scalars (dp, Ie) with
# be_bar = Ie 1 + a s_hat:
#   r1 = a_tr - a(dp) - sqrt(6) dp Ie = 0,  a(dp) = sqrt(2/3) R(p_n + dp) / mu
#   r2 = Ie^3 - a^2 Ie / 2 + a^3 det(s_hat) - 1 = 0
# solved by a 2x2 Newton from (0, tr(be_bar_trial)/3).  Persistent state: p and the isochoric
# inverse plastic right Cauchy-Green tensor Cp_bar^-1 = J^(2/3) F^-1 be_bar F^-T (hidden), plus
# be_bar itself as the user-visible ISV.  The tangent dP/dF is the exact derivative of this
# algorithm (implicit differentiation of the 2x2 system), column by column.

SQ32 = np.sqrt(1.5)
SQ6 = np.sqrt(6.0)


def _det3(A):
    return (
        A[..., 0, 0] * (A[..., 1, 1] * A[..., 2, 2] - A[..., 1, 2] * A[..., 2, 1])
        - A[..., 0, 1] * (A[..., 1, 0] * A[..., 2, 2] - A[..., 1, 2] * A[..., 2, 0])
        + A[..., 0, 2] * (A[..., 1, 0] * A[..., 2, 1] - A[..., 1, 1] * A[..., 2, 0])
    )


def _cof3(A):
    """Cofactor matrix: d det(A) / dA (valid for singular A)."""
    C = np.empty_like(A)
    for i in range(3):
        for j in range(3):
            i1, i2 = (i + 1) % 3, (i + 2) % 3
            j1, j2 = (j + 1) % 3, (j + 2) % 3
            C[..., i, j] = A[..., i1, j1] * A[..., i2, j2] - A[..., i1, j2] * A[..., i2, j1]
    return C


def fefp_scale_tol(hard, mu, a_tr, rtol=NEWTON_RTOL):
    return np.maximum(rtol * stress_scale(hard.sig0, mu), rtol * (SQ32 * mu * a_tr))


def fefp_update(F9, cpinv_n, p_n, E, nu, hard, maxit=NEWTON_MAXIT, rtol=NEWTON_RTOL, tangent=True):
    """F9 (N,9) non-symmetric ordering; cpinv_n (N,6) Mandel; p_n (N,).
    Returns dict(P (N,9), be_bar (N,6), cpinv (N,6), p (N,), Ct (N,9,9), plastic, iters, notconv)."""
    F9 = np.asarray(F9, dtype=np.float64)
    N = F9.shape[0]
    lmbda, mu = lame(E, nu)
    kappa = lmbda + 2 * mu / 3
    I3 = np.eye(3)
    F = nsym_to_tensor(F9)
    G = mandel_to_tensor(cpinv_n)
    p_n = np.asarray(p_n, dtype=np.float64).reshape(-1)

    J = _det3(F)
    Finv = _cof3(F).transpose(0, 2, 1) / J[:, None, None]
    Jm23 = J ** (-2.0 / 3.0)
    GFt = G @ F.transpose(0, 2, 1)
    B = F @ GFt
    btr = Jm23[:, None, None] * B
    Itr = (btr[:, 0, 0] + btr[:, 1, 1] + btr[:, 2, 2]) / 3.0
    d = btr - Itr[:, None, None] * I3
    a_tr = np.sqrt(np.einsum("nij,nij->n", d, d))
    f_trial = SQ32 * mu * a_tr - hard.R(p_n)
    plastic = f_trial > 0.0

    dp = np.zeros(N)
    Ie = Itr.copy()
    a = a_tr.copy()
    shat = np.zeros((N, 3, 3))
    delta = np.zeros(N)
    iters = np.zeros(N, dtype=np.int32)
    notconv = np.zeros(N, dtype=bool)
    idx = np.nonzero(plastic)[0]
    if idx.size:
        sh = d[idx] / a_tr[idx, None, None]
        de = _det3(sh)
        atr = a_tr[idx]
        pn = p_n[idx]
        x_dp = np.zeros(idx.size)
        x_Ie = Itr[idx].copy()
        tol1 = fefp_scale_tol(hard, mu, atr, rtol)
        it = np.zeros(idx.size, dtype=np.int32)
        done = np.zeros(idx.size, dtype=bool)
        for k in range(maxit + 1):
            aa = np.sqrt(2.0 / 3.0) * hard.R(pn + x_dp) / mu
            r1 = atr - aa - SQ6 * x_dp * x_Ie
            r2 = x_Ie**3 - 0.5 * aa * aa * x_Ie + aa**3 * de - 1.0
            conv = (np.abs(SQ32 * mu * r1) <= tol1) & (np.abs(r2) <= 1e-14)
            done |= conv
            if done.all():
                break
            if k == maxit:
                notconv[idx[~done]] = True
                break
            ap = np.sqrt(2.0 / 3.0) * hard.dR(pn + x_dp) / mu
            j11 = -ap - SQ6 * x_Ie
            j12 = -SQ6 * x_dp
            j21 = (-aa * x_Ie + 3 * aa * aa * de) * ap
            j22 = 3 * x_Ie**2 - 0.5 * aa * aa
            det = j11 * j22 - j12 * j21
            ddp = (-r1 * j22 + r2 * j12) / det
            dIe = (-j11 * r2 + j21 * r1) / det
            act = ~done
            x_dp = np.where(act, x_dp + ddp, x_dp)
            x_Ie = np.where(act, x_Ie + dIe, x_Ie)
            it += act
        dp[idx] = x_dp
        Ie[idx] = x_Ie
        a[idx] = np.sqrt(2.0 / 3.0) * hard.R(pn + x_dp) / mu
        shat[idx] = sh
        delta[idx] = de
        iters[idx] = it

    be = np.where(plastic[:, None, None], Ie[:, None, None] * I3 + a[:, None, None] * shat, btr)
    s = mu * (be - ((be[:, 0, 0] + be[:, 1, 1] + be[:, 2, 2]) / 3.0)[:, None, None] * I3)
    tau = 0.5 * kappa * (J * J - 1.0)[:, None, None] * I3 + s
    FinvT = Finv.transpose(0, 2, 1)
    P = tau @ FinvT
    p = p_n + dp
    cpinv = (J ** (2.0 / 3.0))[:, None, None] * (Finv @ be @ FinvT)

    out = dict(
        P=tensor_to_nsym(P), be_bar=tensor_to_mandel(be), cpinv=tensor_to_mandel(0.5 * (cpinv + cpinv.transpose(0, 2, 1))),
        p=p, plastic=plastic, iters=iters, notconv=notconv, f_trial=f_trial,
    )
    if not tangent:
        return out

    # ---- consistent tangent dP/dF, one column per basis direction dF = e_k (x) e_l -------------
    Ct = np.empty((N, 9, 9))
    with np.errstate(divide="ignore", invalid="ignore"):
        ap = np.sqrt(2.0 / 3.0) * hard.dR(p) / mu
        gI = 3 * Ie**2 - 0.5 * a * a
        dIe_da = (a * Ie - 3 * a * a * delta) / gI
        dIe_dd = -(a**3) / gI
        r_dp = -ap - SQ6 * Ie - SQ6 * dp * dIe_da * ap
        r_dd = -SQ6 * dp * dIe_dd
        cof_s = _cof3(shat)
    for col, (k, l) in enumerate(NSYM_IDX):
        dF = np.zeros((N, 3, 3))
        dF[:, k, l] = 1.0
        trFinvdF = Finv[:, l, k]
        dJ = J * trFinvdF
        M = dF @ GFt
        dB = M + M.transpose(0, 2, 1)
        dbtr = Jm23[:, None, None] * dB - (2.0 / 3.0) * trFinvdF[:, None, None] * btr
        dItr = (dbtr[:, 0, 0] + dbtr[:, 1, 1] + dbtr[:, 2, 2]) / 3.0
        dd = dbtr - dItr[:, None, None] * I3
        ds_el = mu * dd
        with np.errstate(divide="ignore", invalid="ignore"):
            datr = np.einsum("nij,nij->n", shat, dd)
            dsh = (dd - shat * datr[:, None, None]) / a_tr[:, None, None]
            ddel = np.einsum("nij,nij->n", cof_s, dsh)
            ddp = -(datr + r_dd * ddel) / r_dp
            da = ap * ddp
            ds_pl = mu * (da[:, None, None] * shat + a[:, None, None] * dsh)
        ds = np.where(plastic[:, None, None], ds_pl, ds_el)
        dtau = (kappa * J * dJ)[:, None, None] * I3 + ds
        dP = dtau @ FinvT - P @ dF.transpose(0, 2, 1) @ FinvT
        Ct[:, :, col] = tensor_to_nsym(dP)
    out["Ct"] = Ct
    return out


def fefp_initial_state(n):
    """p = 0, be_bar = Cp_bar^-1 = identity (behavior.init_state, jaxmat.py:35;
    finite_strain_elastoplasticity.py:181)."""
    ident = np.zeros((n, 6))
    ident[:, :3] = 1.0
    return dict(p=np.zeros(n), be_bar=ident.copy(), cpinv=ident.copy())


def cpinv_from_be_bar(F9, be_bar):
    """Hidden state from the user-visible pair (F_n, be_bar_n):
    Cp_bar^-1 = J^(2/3) F^-1 be_bar F^-T."""
    F = nsym_to_tensor(F9)
    be = mandel_to_tensor(be_bar)
    J = _det3(F)
    Finv = _cof3(F).transpose(0, 2, 1) / J[:, None, None]
    G = (J ** (2.0 / 3.0))[:, None, None] * (Finv @ be @ Finv.transpose(0, 2, 1))
    return tensor_to_mandel(0.5 * (G + G.transpose(0, 2, 1)))
