"""Tensor <-> vector conventions at the boundary (reference ``dolfinx_materials/utils.py:146-212``,
``docs/intro.md:134-175``) and the host-side state conversion the FeFp law needs."""
from __future__ import annotations

import numpy as np

SQ2 = np.sqrt(2.0)
#: (row, col) of each entry of the non-symmetric 9-vector [11,22,33,12,21,13,31,23,32]
NSYM_IDX = ((0, 0), (1, 1), (2, 2), (0, 1), (1, 0), (0, 2), (2, 0), (1, 2), (2, 1))


def mandel_to_tensor(v):
    v = np.asarray(v, dtype=np.float64)
    T = np.empty(v.shape[:-1] + (3, 3))
    T[..., 0, 0], T[..., 1, 1], T[..., 2, 2] = v[..., 0], v[..., 1], v[..., 2]
    T[..., 0, 1] = T[..., 1, 0] = v[..., 3] / SQ2
    T[..., 0, 2] = T[..., 2, 0] = v[..., 4] / SQ2
    T[..., 1, 2] = T[..., 2, 1] = v[..., 5] / SQ2
    return T


def tensor_to_mandel(T):
    T = np.asarray(T, dtype=np.float64)
    v = np.empty(T.shape[:-2] + (6,))
    v[..., 0], v[..., 1], v[..., 2] = T[..., 0, 0], T[..., 1, 1], T[..., 2, 2]
    v[..., 3] = SQ2 * T[..., 0, 1]
    v[..., 4] = SQ2 * T[..., 0, 2]
    v[..., 5] = SQ2 * T[..., 1, 2]
    return v


def nsym_to_tensor(v):
    v = np.asarray(v, dtype=np.float64)
    T = np.empty(v.shape[:-1] + (3, 3))
    for k, (i, j) in enumerate(NSYM_IDX):
        T[..., i, j] = v[..., k]
    return T


def tensor_to_nsym(T):
    T = np.asarray(T, dtype=np.float64)
    v = np.empty(T.shape[:-2] + (9,))
    for k, (i, j) in enumerate(NSYM_IDX):
        v[..., k] = T[..., i, j]
    return v


def cp_bar_inv_from_be_bar(F9, be_bar6):
    """Hidden FeFp state from the user-visible pair (F_n, be_bar_n):
    ``Cp_bar^-1 = J^(2/3) F^-1 be_bar F^-T`` (Mandel in, Mandel out).

    Rows whose ``be_bar`` is not positive definite (e.g. the all-zero initial value of a dolfinx
    quadrature Function handed over by ``QuadratureMap.initialize_state``,
    ``quadrature_map.py:281-295``) are replaced by the identity, the natural unstressed state the
    reference documents (``finite_strain_elastoplasticity.py:181``).
    """
    F = nsym_to_tensor(F9)
    be = mandel_to_tensor(be_bar6)
    bad = ~(np.linalg.det(be) > 0.0) | ~(be[..., 0, 0] > 0.0)
    if bad.any():
        be = be.copy()
        be[bad] = np.eye(3)
    J = np.linalg.det(F)
    Finv = np.linalg.inv(F)
    G = (J ** (2.0 / 3.0))[..., None, None] * (Finv @ be @ np.swapaxes(Finv, -1, -2))
    return tensor_to_mandel(0.5 * (G + np.swapaxes(G, -1, -2))), tensor_to_mandel(be)


#: (i, j) of the 21 entries of the symmetric-packed 6x6 tangent (upper triangle, row-major)
SYM_IDX = tuple((i, j) for i in range(6) for j in range(i, 6))


def pack_sym_tangent(ct):
    """(N,6,6) symmetric -> (N,21)."""
    ct = np.asarray(ct, dtype=np.float64).reshape(-1, 6, 6)
    ii, jj = np.array([p[0] for p in SYM_IDX]), np.array([p[1] for p in SYM_IDX])
    return ct[:, ii, jj]


def unpack_sym_tangent(ct21):
    """(N,21) -> (N,6,6) symmetric (what ``jacobian_flatten`` of quadrature_map.py:83-105 holds)."""
    ct21 = np.asarray(ct21, dtype=np.float64).reshape(-1, 21)
    out = np.empty((ct21.shape[0], 6, 6))
    for t, (i, j) in enumerate(SYM_IDX):
        out[:, i, j] = ct21[:, t]
        out[:, j, i] = ct21[:, t]
    return out


def tangent_from_coefficients(coef):
    """(N,9) ``(c1, c2, c3, n[6])`` -> (N,6,6): ``Ct = c1 1x1 + c2 I + c3 n x n`` (the ``"coef"`` tangent layout)."""
    coef = np.asarray(coef, dtype=np.float64).reshape(-1, 9)
    one = np.array([1.0, 1.0, 1.0, 0.0, 0.0, 0.0])
    n = coef[:, 3:]
    return (coef[:, 0, None, None] * np.outer(one, one)[None] + coef[:, 1, None, None] * np.eye(6)[None]
            + coef[:, 2, None, None] * n[:, :, None] * n[:, None, :])


def tangent_from_pack4(stress, pack):
    """(N,6) stress and (N,4) ``(c1, c2, c3, w)`` of the same update -> (N,6,6) (the ``"pack4"`` tangent layout): the flow
    direction is ``n = dev(stress) w``, formed with the kernel's own three operations (``small_strain.hpp`` step 5), then
    ``Ct = c1 1x1 + c2 I + c3 n x n`` entry by entry as ``fma(c3, n_i n_j, t0)`` would (numpy rounds the product and the sum
    separately: equal to the kernel's block to 1 ulp of the ``c3`` term, not to the bit)."""
    stress = np.asarray(stress, dtype=np.float64).reshape(-1, 6)
    pack = np.asarray(pack, dtype=np.float64).reshape(-1, 4)
    third = (stress[:, 0] + stress[:, 1] + stress[:, 2]) * (1.0 / 3.0)
    n = stress * pack[:, 3:4]
    n[:, :3] = (stress[:, :3] - third[:, None]) * pack[:, 3:4]
    return tangent_from_coefficients(np.concatenate([pack[:, :3], n], axis=1))
