"""Stand-in host FE loop (dolfinx is not installed on either box): trilinear hexahedra on a
structured unit-cube mesh, 2x2x2 Gauss points (what ``quad_degree=2`` gives on a hexahedron,
SURVEY.md App. A), scipy sparse direct solve.

It plays the role of everything ABOVE the hot path in the reference -- the UFL forms, dolfinx
assembly and the SNES Newton loop of ``NonlinearMaterialProblem`` (``solvers.py:31-96, :182-196``)
-- so that the constitutive engine can be exercised in a real global Newton iteration:
per iteration ``qmap.update()`` (gradients at the Gauss points -> ``material.integrate`` on the
GPU -> flux and tangent quadrature arrays), then host assembly of residual and Jacobian from
those arrays, then a linear solve; ``qmap.advance()`` after convergence.

Not part of the product package and not a re-implementation of dolfinx: host code only.
"""
from __future__ import annotations

import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

SQ2 = np.sqrt(2.0)
#: (row, col) of the entries of the 9-vector [11,22,33,12,21,13,31,23,32] (utils.py:168-190)
NSYM_IDX = ((0, 0), (1, 1), (2, 2), (0, 1), (1, 0), (0, 2), (2, 0), (1, 2), (2, 1))


class HexMesh:
    def __init__(self, n):
        self.n = n
        self.h = 1.0 / n
        g = np.arange(n + 1) * self.h
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
        self.coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        nid = lambda i, j, k: (i * (n + 1) + j) * (n + 1) + k  # noqa: E731
        I, Jj, K = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        I, Jj, K = I.ravel(), Jj.ravel(), K.ravel()
        corners = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
        self.conn = np.stack([nid(I + a, Jj + b, K + c) for a, b, c in corners], axis=1)
        self.num_cells = self.conn.shape[0]
        self.num_nodes = self.coords.shape[0]
        self.ndof = 3 * self.num_nodes
        self.cell_dofs = (3 * self.conn[:, :, None] + np.arange(3)[None, None, :]).reshape(self.num_cells, 24)
        # shape-function gradients at the 8 Gauss points (identical for every cell: uniform cubes)
        gp = np.array([-1.0, 1.0]) / np.sqrt(3.0)
        xi_nodes = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)
        self.nqp = 8
        self.dN = np.zeros((8, 8, 3))  # [gp, node, d/dx_j]
        q = 0
        for a in gp:
            for b in gp:
                for c in gp:
                    xi = np.array([a, b, c])
                    for m in range(8):
                        s = xi_nodes[m]
                        for d in range(3):
                            f = 0.125 * s[d]
                            for o in range(3):
                                if o != d:
                                    f *= 1 + s[o] * xi[o]
                            self.dN[q, m, d] = f * 2.0 / self.h
                    q += 1
        self.wdet = (self.h / 2.0) ** 3  # weight 1 x det J per Gauss point
        # B matrices: strain in Mandel 6-vector form and displacement gradient in 9-vector form
        self.B_eps = np.zeros((8, 6, 24))
        self.B_grad = np.zeros((8, 9, 24))
        for q in range(8):
            for m in range(8):
                dx, dy, dz = self.dN[q, m]
                c = 3 * m
                self.B_eps[q, 0, c + 0] = dx
                self.B_eps[q, 1, c + 1] = dy
                self.B_eps[q, 2, c + 2] = dz
                self.B_eps[q, 3, c + 0] = dy / SQ2
                self.B_eps[q, 3, c + 1] = dx / SQ2
                self.B_eps[q, 4, c + 0] = dz / SQ2
                self.B_eps[q, 4, c + 2] = dx / SQ2
                self.B_eps[q, 5, c + 1] = dz / SQ2
                self.B_eps[q, 5, c + 2] = dy / SQ2
                for t, (i, j) in enumerate(NSYM_IDX):
                    self.B_grad[q, t, c + i] = self.dN[q, m, j]
        rows = np.repeat(self.cell_dofs, 24, axis=1)
        cols = np.tile(self.cell_dofs, (1, 24))
        self._rows, self._cols = rows.ravel(), cols.ravel()

    def nodes_on(self, axis, value):
        return np.nonzero(np.abs(self.coords[:, axis] - value) < 1e-12)[0]

    # gradient "expressions" evaluated at the Gauss points of `cells` (QuadratureExpression.eval)
    def strain(self, u, cells):
        ue = u[self.cell_dofs[cells]]
        return np.einsum("qik,ck->cqi", self.B_eps, ue).reshape(-1, 6)

    def deformation_gradient(self, u, cells):
        ue = u[self.cell_dofs[cells]]
        g = np.einsum("qik,ck->cqi", self.B_grad, ue)
        g[:, :, :3] += 1.0
        return g.reshape(-1, 9)

    def assemble(self, flux, tangent, B):
        """flux (ncell*8, nf), tangent (ncell*8, nf*ng) row-major quadrature arrays ->
        residual vector and sparse Jacobian."""
        nf = B.shape[1]
        f = flux.reshape(self.num_cells, 8, nf)
        Ct = tangent.reshape(self.num_cells, 8, nf, nf)
        re = self.wdet * np.einsum("qik,cqi->ck", B, f)
        Ke = self.wdet * np.einsum("qik,cqij,qjl->ckl", B, Ct, B, optimize=True)
        r = np.zeros(self.ndof)
        np.add.at(r, self.cell_dofs.ravel(), re.ravel())
        K = sp.coo_matrix((Ke.ravel(), (self._rows, self._cols)), shape=(self.ndof, self.ndof)).tocsr()
        return r, K


def newton_solve(mesh, qmap, u, bc_dofs, bc_vals, B, flux_name, atol=1e-8, rtol=1e-10, maxit=25, timers=None):
    """One load increment: Newton iterations with the constitutive update as the residual
    callback (``solvers.py:72``), ``qmap.advance()`` once converged (``solvers.py:194``)."""
    timers = timers if timers is not None else {}
    free = np.setdiff1d(np.arange(mesh.ndof), bc_dofs)
    u[bc_dofs] = bc_vals
    norms = []
    for it in range(maxit):
        t0 = time.perf_counter()
        qmap.update()  # <- the hot path
        t1 = time.perf_counter()
        flux = qmap.fluxes[flux_name].x.array.reshape(-1, qmap.fluxes[flux_name].dim)
        Ct = qmap.jacobian_flatten.x.array.reshape(flux.shape[0], -1)
        r, K = mesh.assemble(flux, Ct, B)
        t2 = time.perf_counter()
        rn = np.linalg.norm(r[free])
        norms.append(rn)
        timers["constitutive"] = timers.get("constitutive", 0.0) + (t1 - t0)
        timers["assembly"] = timers.get("assembly", 0.0) + (t2 - t1)
        if rn < atol or (it > 0 and rn < rtol * norms[0]):
            break
        du = spla.spsolve(K[free][:, free].tocsc(), -r[free])
        timers["solve"] = timers.get("solve", 0.0) + (time.perf_counter() - t2)
        u[free] += du
    else:
        raise RuntimeError(f"Newton did not converge: {norms}")
    qmap.advance()
    return norms
