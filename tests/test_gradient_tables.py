"""CPU tests of the host side of the Lagrange-simplex gradient (dolfinx_materials_amd/gradient.py): the tabulated
reference derivatives handed to dxm_mesh_create_simplex and the stand-alone P2 dofmap.  The reference evaluates these
gradients through dolfinx (quadrature_function.py:45-51); the spaces are the P2 ones of its demos
(finite_strain_elastoplasticity.py:115-117, plane_elastoplasticity.py:96-100)."""
import os
import sys

import numpy as np
import pytest

from dolfinx_materials_amd.gradient import BASIX_EDGES, lagrange_simplex_table, p2_dofmap, simplex_quadrature

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from helpers import KUHN, simplex_host_gradient, triangle_grid  # noqa: E402


def _tets(n=2, seed=1):
    from hex_fem import HexMesh

    m = HexMesh(n)
    rng = np.random.default_rng(seed)
    coords = m.coords + 0.2 * m.h * rng.uniform(-1, 1, m.coords.shape)
    return coords, np.concatenate([m.conn[:, list(k)] for k in KUHN], axis=0).astype(np.int32)


def _p2_values(tdim, xi):
    """P2 basis values at reference points, basix dof order (for the finite-difference check of the table)."""
    lam = np.concatenate([1.0 - xi.sum(axis=1, keepdims=True), xi], axis=1)
    cols = [lam[:, v] * (2 * lam[:, v] - 1) for v in range(tdim + 1)] + [4 * lam[:, i] * lam[:, j] for i, j in BASIX_EDGES[tdim]]
    return np.stack(cols, axis=1)


@pytest.mark.parametrize("tdim", [2, 3])
def test_p2_table_is_the_derivative_of_a_nodal_basis(tdim):
    pts = simplex_quadrature(tdim, 2)
    assert pts.shape == ({2: 3, 3: 4}[tdim], tdim)
    tab = lagrange_simplex_table(tdim, 2, pts)
    nd = {2: 6, 3: 10}[tdim]
    assert tab.shape == (len(pts), nd, tdim)
    assert np.abs(tab.sum(axis=1)).max() < 1e-14            # partition of unity
    h = 1e-6
    for d in range(tdim):
        e = np.zeros(tdim)
        e[d] = h
        fd = (_p2_values(tdim, pts + e) - _p2_values(tdim, pts - e)) / (2 * h)
        assert np.abs(fd - tab[:, :, d]).max() < 1e-9
    # nodal: N_m(x_k) = delta_mk at the vertices and the edge midpoints in BASIX_EDGES order
    verts = np.concatenate([np.zeros((1, tdim)), np.eye(tdim)], axis=0)
    nodes = np.concatenate([verts, [0.5 * (verts[i] + verts[j]) for i, j in BASIX_EDGES[tdim]]], axis=0)
    assert np.abs(_p2_values(tdim, nodes) - np.eye(nd)).max() < 1e-15


@pytest.mark.parametrize("tdim,degree", [(2, 1), (2, 2), (3, 1), (3, 2)])
def test_polynomial_fields_are_differentiated_exactly(tdim, degree):
    """u = a polynomial of the element's degree, sampled at the dofs: the tabulated gradient at the Gauss points is the
    analytic one on a distorted (still straight-sided) mesh -- pins table, dof order and the affine map together."""
    coords, cells = triangle_grid(3, seed=2) if tdim == 2 else _tets()
    pts = simplex_quadrature(tdim, 2)
    tab = lagrange_simplex_table(tdim, degree, pts)
    if degree == 1:
        dofmap, n_dofs, xd = cells, len(coords), coords
    else:
        dofmap, n_dofs, edges = p2_dofmap(cells)
        xd = np.concatenate([coords, 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])], axis=0)
        assert n_dofs == len(xd) and len(np.unique(edges, axis=0)) == len(edges)
    rng = np.random.default_rng(0)
    G = rng.standard_normal((tdim, tdim))
    Q = rng.standard_normal((tdim, tdim, tdim)) * (degree == 2)
    Q = 0.5 * (Q + Q.transpose(0, 2, 1))
    u = xd @ G.T + np.einsum("iab,na,nb->ni", Q, xd, xd)
    H = simplex_host_gradient(np.pad(coords, ((0, 0), (0, 3 - tdim))), cells, dofmap, u.ravel(), tab)
    lam = np.concatenate([1.0 - pts.sum(axis=1, keepdims=True), pts], axis=1)
    xq = np.einsum("qv,cva->cqa", lam, coords[cells])
    exact = G[None, None] + 2.0 * np.einsum("iab,cqb->cqia", Q, xq)
    assert np.abs(H[:, :, :tdim, :tdim] - exact).max() < 1e-12
    if tdim == 2:
        assert np.abs(H[:, :, 2]).max() == 0.0 and np.abs(H[:, :, :, 2]).max() == 0.0


def test_p2_dofmap_shares_edge_dofs_between_cells():
    coords, cells = _tets(2)
    dofmap, n_dofs, edges = p2_dofmap(cells)
    assert dofmap.shape == (len(cells), 10) and dofmap.dtype == np.int32
    assert np.array_equal(dofmap[:, :4], cells)
    nv = len(coords)
    for c in (0, 7, len(cells) - 1):
        for k, (i, j) in enumerate(BASIX_EDGES[3]):
            assert sorted(edges[dofmap[c, 4 + k] - nv]) == sorted((cells[c, i], cells[c, j]))
    assert n_dofs == nv + len(edges) and dofmap.max() == n_dofs - 1
    # Euler: a Kuhn-split 2x2x2 cube has 27 vertices, 48 cells, 98 edges
    assert len(edges) == 98
