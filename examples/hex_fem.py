"""Stand-in host FE loop (dolfinx is not installed on either box): trilinear hexahedra on a
structured unit-cube mesh, 2x2x2 Gauss points (what ``quad_degree=2`` gives on a hexahedron,
SURVEY.md App. A).

It plays the role of everything ABOVE the hot path in the reference -- the UFL forms, dolfinx
assembly and the SNES Newton loop of ``NonlinearMaterialProblem`` (``solvers.py:31-96, :182-196``)
-- so that the constitutive engine can be exercised in a real global Newton iteration:
per iteration ``qmap.update()`` (gradients at the Gauss points -> ``material.integrate`` on the
GPU -> flux and tangent quadrature arrays), then host assembly of residual and Jacobian from
those arrays, then a linear solve; ``qmap.advance()`` after convergence.

Host side, sized for BASELINE config 5's 64^3 stand-in (262 144 cells, 2.1e6 Gauss points, 8.2e5 dofs):

* the Jacobian is assembled into a block-CSR matrix (3x3 nodal blocks, 27-point stencil) whose pattern and
  element-to-block map are built once from the grid structure;
* the element matrices can be formed from any of the engine's tangent layouts: the full (N,6,6) / (N,9,9)
  block, its 21-entry upper triangle, or -- the cheapest -- the nine coefficients of
  ``Ct = c1 1x1 + c2 I + c3 n x n`` (``tangent_layout="coef"``): ``B^T Ct B = c1 (B^T 1)(B^T 1)^T + c2 B^T B
  + c3 (B^T n)(B^T n)^T`` needs one 24-vector per Gauss point instead of a 6x6 block;
* the linear systems are solved by conjugate gradients preconditioned with a geometric multigrid V-cycle
  (trilinear prolongation, Galerkin coarse operators, damped-Jacobi smoothing; the role GMRES + GAMG play in
  ``demos/jax/finite_strain_elastoplasticity/finite_strain_elastoplasticity.py:192-200``), a sparse direct
  solve (MUMPS in ``tests/uniaxial_tension.py:74-82``) only on small meshes.

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
CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)])
#: upper triangle (i <= j) of a 6x6 block, row-major: position in the 21-entry packed tangent
SYM_POS = np.zeros((6, 6), dtype=np.int64)
_t = 0
for _i in range(6):
    for _j in range(_i, 6):
        SYM_POS[_i, _j] = SYM_POS[_j, _i] = _t
        _t += 1


class HexMesh:
    def __init__(self, n):
        self.n = n
        self.h = 1.0 / n
        g = np.arange(n + 1) * self.h
        X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
        self.coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
        m = n + 1
        I, Jj, K = (a.ravel() for a in np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"))
        self.conn = np.stack([((I + a) * m + (Jj + b)) * m + (K + c) for a, b, c in CORNERS], axis=1)
        self.num_cells = self.conn.shape[0]
        self.num_nodes = self.coords.shape[0]
        self.ndof = 3 * self.num_nodes
        self.cell_dofs = (3 * self.conn[:, :, None] + np.arange(3)[None, None, :]).reshape(self.num_cells, 24)
        # shape-function gradients at the 8 Gauss points (identical for every cell: uniform cubes)
        gp = np.array([-1.0, 1.0]) / np.sqrt(3.0)
        xi_nodes = 2.0 * CORNERS - 1.0
        self.nqp = 8
        self.dN = np.zeros((8, 8, 3))  # [gp, node, d/dx_j]
        q = 0
        for a in gp:
            for b in gp:
                for c in gp:
                    xi = np.array([a, b, c])
                    for k in range(8):
                        s = xi_nodes[k]
                        for d in range(3):
                            f = 0.125 * s[d]
                            for o in range(3):
                                if o != d:
                                    f *= 1 + s[o] * xi[o]
                            self.dN[q, k, d] = f * 2.0 / self.h
                    q += 1
        self.wdet = (self.h / 2.0) ** 3  # weight 1 x det J per Gauss point
        # B matrices: strain in Mandel 6-vector form and displacement gradient in 9-vector form
        self.B_eps = np.zeros((8, 6, 24))
        self.B_grad = np.zeros((8, 9, 24))
        for q in range(8):
            for k in range(8):
                dx, dy, dz = self.dN[q, k]
                c = 3 * k
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
                    self.B_grad[q, t, c + i] = self.dN[q, k, j]
        self._pattern = None

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

    # ---- sparsity: 3x3 blocks on the 27-point node stencil, built from the grid structure ---------------
    def pattern(self):
        """(indptr, indices, slot): block-CSR structure over nodes and, for every (cell, local node a, local node b),
        the index of the block that element entry adds to."""
        if self._pattern is None:
            m = self.n + 1
            ijk = np.stack(np.unravel_index(np.arange(self.num_nodes), (m, m, m)), axis=1)
            offs = np.array([(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1)])   # ascending node id
            nb = ijk[:, None, :] + offs[None, :, :]
            valid = np.all((nb >= 0) & (nb < m), axis=2)
            indptr = np.concatenate([[0], np.cumsum(valid.sum(axis=1))]).astype(np.int64)
            pos = np.cumsum(valid, axis=1) - 1                      # rank of each valid neighbour within its row
            ids = (nb[..., 0] * m + nb[..., 1]) * m + nb[..., 2]
            indices = ids[valid].astype(np.int32)
            d = CORNERS[None, :, :] - CORNERS[:, None, :]           # offset from local node a to local node b
            o = (d[..., 0] + 1) * 9 + (d[..., 1] + 1) * 3 + (d[..., 2] + 1)
            rows = self.conn[:, :, None]                              # (cells, a, 1)
            slot = indptr[rows] + pos[rows, o[None, :, :]]           # (cells, a, b)
            self._pattern = (indptr, indices, slot.reshape(-1))
        return self._pattern

    def element_matrices(self, tangent, B, layout="full", flux=None):
        """``wdet * sum_q B_q^T Ct_q B_q`` per cell, (cells, 24, 24), from any tangent layout of the engine (``"pack4"``: the four
        coefficients plus the stress of the same update, from which the flow direction is ``n = dev(stress) w``)."""
        nc, w = self.num_cells, self.wdet
        if layout == "pack4":   # -> the nine-coefficient form: n rebuilt with the kernel's own three operations
            pk = np.asarray(tangent).reshape(-1, 4)
            sg = np.asarray(flux).reshape(-1, 6)
            third = (sg[:, 0] + sg[:, 1] + sg[:, 2]) * (1.0 / 3.0)
            nn = sg * pk[:, 3:4]
            nn[:, :3] = (sg[:, :3] - third[:, None]) * pk[:, 3:4]
            tangent, layout = np.concatenate([pk[:, :3], nn], axis=1), "coef"
        if layout == "coef":   # Ct = c1 1x1 + c2 I + c3 n x n: rank structure, no 6x6 block is ever formed
            cf = np.asarray(tangent).reshape(nc, 8, 9)
            one = np.array([1.0, 1.0, 1.0, 0.0, 0.0, 0.0])
            v1 = np.einsum("i,qik->qk", one, B)                                        # B_q^T 1
            Ke = (cf[:, :, 0] @ np.einsum("qk,ql->qkl", v1, v1).reshape(8, 576)
                  + cf[:, :, 1] @ np.einsum("qik,qil->qkl", B, B).reshape(8, 576)).reshape(nc, 24, 24)
            for q in range(8):
                v3 = cf[:, q, 3:] @ B[q]                                               # B_q^T n, (cells, 24)
                Ke += (cf[:, q, 2, None] * v3)[:, :, None] * v3[:, None, :]
            return w * Ke
        nf = B.shape[1]
        if layout == "sym":    # 21 upper-triangle entries per point: indexed in place, never expanded per point
            Ct = np.asarray(tangent).reshape(nc, 8, 21)[:, :, SYM_POS]
        else:
            Ct = np.asarray(tangent).reshape(nc, 8, nf, nf)
        Ke = np.zeros((nc, 24, 24))
        for q in range(8):
            Ke += B[q].T @ (Ct[:, q] @ B[q])
        return w * Ke

    def assemble(self, flux, tangent, B, layout="full"):
        """flux (ncell*8, nf) and the tangent quadrature array -> residual vector and block-CSR Jacobian."""
        nf = B.shape[1]
        f = np.asarray(flux).reshape(self.num_cells, 8, nf)
        re = self.wdet * np.einsum("qik,cqi->ck", B, f)
        r = np.bincount(self.cell_dofs.ravel(), weights=re.ravel(), minlength=self.ndof)
        indptr, indices, slot = self.pattern()
        Ke = self.element_matrices(tangent, B, layout, flux=flux).reshape(self.num_cells, 8, 3, 8, 3)
        data = np.empty((len(indices), 3, 3))
        for i in range(3):
            for j in range(3):
                data[:, i, j] = np.bincount(slot, weights=Ke[:, :, i, :, j].reshape(-1), minlength=len(indices))
        K = sp.bsr_matrix((data, indices, indptr), shape=(self.ndof, self.ndof))
        return r, K


# ---- linear solvers ---------------------------------------------------------------------------------
def _prolongation(nc):
    """Trilinear interpolation from the (nc+1)^3 grid to the (2nc+1)^3 grid, per displacement component."""
    mf, mc = 2 * nc + 1, nc + 1
    rows, cols, vals = [], [], []
    for i in range(mf):
        if i % 2 == 0:
            rows.append(i); cols.append(i // 2); vals.append(1.0)
        else:
            rows += [i, i]; cols += [i // 2, i // 2 + 1]; vals += [0.5, 0.5]
    P1 = sp.csr_matrix((vals, (rows, cols)), shape=(mf, mc))
    return sp.kron(sp.kron(sp.kron(P1, P1), P1), sp.identity(3), format="csr")


class Multigrid:
    """Geometric multigrid V-cycle on the structured grid, used as the CG preconditioner.  Constrained dofs
    carry identity rows in K; their rows of the prolongation are zeroed so that no coarse function reaches them."""

    def __init__(self, K, n, free_mask, coarsest=4, sweeps=2):
        self.sweeps = sweeps
        self.levels = []
        A, mask = K.tocsr(), free_mask
        rng = np.random.default_rng(0)
        while n > coarsest and n % 2 == 0:
            P = sp.diags(mask.astype(float)) @ _prolongation(n // 2)
            Ac = (P.T @ A @ P).tocsr()
            dead = np.asarray(Ac.diagonal() == 0.0)               # coarse dofs that only reach constrained fine dofs
            Ac = Ac + sp.diags(dead.astype(float))
            dinv = 1.0 / A.diagonal()
            # damping from the largest eigenvalue of D^-1 A (a few power iterations): the plastic tangent is strongly
            # anisotropic (stiffness along the flow direction drops to H / (H + 3 mu)) and a fixed factor tuned on
            # elasticity makes the smoother diverge
            v = rng.standard_normal(A.shape[0])
            for _ in range(12):
                v = dinv * (A @ v)
                rho = np.linalg.norm(v)
                v /= rho
            self.levels.append((A, P, (4.0 / (3.0 * 1.1 * rho)) * dinv))
            A, mask, n = Ac, ~dead, n // 2
        self.coarse = spla.splu(A.tocsc())

    def _smooth(self, A, wdinv, b, x):
        for _ in range(self.sweeps):
            x = x + wdinv * (b - A @ x)
        return x

    def vcycle(self, b, lvl=0):
        if lvl == len(self.levels):
            return self.coarse.solve(b)
        A, P, dinv = self.levels[lvl]
        x = self._smooth(A, dinv, b, np.zeros_like(b))
        x = x + P @ self.vcycle(P.T @ (b - A @ x), lvl + 1)
        return self._smooth(A, dinv, b, x)


def solve_linear(mesh, K, rhs, free_mask, method="auto", rtol=1e-10, info=None, symmetric=True, abs_tol=None):
    """Solve K dx = rhs on the free dofs (dx = 0 on the constrained ones): sparse LU on small meshes, otherwise
    multigrid-preconditioned CG (symmetric tangents: the small-strain laws) or GMRES (dP/dF of the FeFp law).
    ``abs_tol``: the linear residual is also driven below this ABSOLUTE value (the Newton loop passes a fraction of its own
    absolute tolerance): with a relative stop alone a large first residual leaves a linear residual above the Newton tolerance, and
    a step that is linear takes several "Newton" iterations that are really restarts of the Krylov solver."""
    D = sp.diags(free_mask.astype(float))
    Kc = (D @ K.tocsr() @ D + sp.diags((~free_mask).astype(float))).tocsr()
    b = rhs * free_mask
    if method == "direct" or (method == "auto" and mesh.ndof <= 20_000):
        return spla.spsolve(Kc.tocsc(), b)
    t0 = time.perf_counter()
    mg = Multigrid(Kc, mesh.n, free_mask)
    t1 = time.perf_counter()
    its = [0]
    M = spla.LinearOperator(Kc.shape, matvec=mg.vcycle)
    count = lambda _x: its.__setitem__(0, its[0] + 1)   # noqa: E731
    if abs_tol is not None:
        bn = float(np.linalg.norm(b))
        if bn > 0.0:
            rtol = max(min(rtol, abs_tol / bn), 1e-14)   # (never below what fp64 residuals can show)
    if symmetric:
        x, flag = spla.cg(Kc, b, rtol=rtol, atol=0.0, maxiter=400, M=M, callback=count)
    else:
        x, flag = spla.gmres(Kc, b, rtol=rtol, atol=0.0, restart=60, maxiter=20, M=M, callback=count, callback_type="pr_norm")
    if flag != 0:
        raise RuntimeError(f"Krylov solver did not converge ({flag}) after {its[0]} iterations")
    if info is not None:
        info.update(cg_iterations=its[0], mg_setup_s=t1 - t0, cg_s=time.perf_counter() - t1)
    return x


def newton_solve(mesh, qmap, u, bc_dofs, bc_vals, B, flux_name, atol=1e-8, rtol=1e-10, maxit=25, timers=None, solver="auto",
                 log=None):
    """One load increment: Newton iterations with the constitutive update as the residual
    callback (``solvers.py:72``), ``qmap.advance()`` once converged (``solvers.py:194``)."""
    timers = timers if timers is not None else {}
    free = np.ones(mesh.ndof, dtype=bool)
    free[bc_dofs] = False
    # predictor: scale the last converged field with the load (imposing the whole increment on the boundary
    # nodes alone starts Newton from a boundary layer of large strains that flips points between the elastic
    # and the plastic branch for many iterations on fine meshes)
    old = u[bc_dofs]
    k = np.abs(old).argmax() if len(old) else 0
    layout = getattr(qmap.material, "tangent_layout", "full")
    if len(old) and abs(old[k]) > 0.0 and np.isfinite(bc_vals[k] / old[k]):
        u *= bc_vals[k] / old[k]
    elif len(old) and np.any(bc_vals != old):
        # nothing to scale (the first increment): a LINEAR predictor with the tangent of the state the map was last updated at --
        # the boundary increment dg is lifted through it, K_ff du_f = -(r + K dg)_f.  Evaluating the law at "u with only the
        # boundary nodes moved" instead puts strains of exx * n into the first layer of cells (16 % at 64^3): that layer returns
        # plastically in the first iterate and an ELASTIC step takes five Newton iterations to shake the layer off
        t0 = time.perf_counter()
        qmap.update()
        r, K = mesh.assemble(qmap.fluxes[flux_name].values, qmap.jacobian_flatten.values, B, layout)
        dg = np.zeros(mesh.ndof)
        dg[bc_dofs] = bc_vals - old
        u += solve_linear(mesh, K, -(r + K @ dg), free, method=solver, symmetric=B.shape[1] == 6, abs_tol=0.1 * atol) + dg
        timers["predictor"] = timers.get("predictor", 0.0) + (time.perf_counter() - t0)
    u[bc_dofs] = bc_vals
    norms = []
    for it in range(maxit):
        t0 = time.perf_counter()
        qmap.update()  # <- the hot path
        t1 = time.perf_counter()
        r, K = mesh.assemble(qmap.fluxes[flux_name].values, qmap.jacobian_flatten.values, B, layout)
        t2 = time.perf_counter()
        rn = np.linalg.norm(r[free])
        norms.append(rn)
        timers["constitutive"] = timers.get("constitutive", 0.0) + (t1 - t0)
        timers["assembly"] = timers.get("assembly", 0.0) + (t2 - t1)
        timers["newton_iterations"] = timers.get("newton_iterations", 0) + 1
        if rn < atol or (it > 0 and rn < rtol * norms[0]):
            break
        info = {}
        du = solve_linear(mesh, K, -r, free, method=solver, info=info, symmetric=B.shape[1] == 6, abs_tol=0.1 * atol)
        timers["solve"] = timers.get("solve", 0.0) + (time.perf_counter() - t2)
        for k, v in info.items():
            timers[k] = timers.get(k, 0) + v
        if log:
            log(f"    newton {it}: |r| = {rn:.3e}  update {t1 - t0:.3f} s  assembly {t2 - t1:.2f} s  solve {time.perf_counter() - t2:.2f} s {info}")
        u += du
    else:
        raise RuntimeError(f"Newton did not converge: {norms}")
    qmap.advance()
    return norms
