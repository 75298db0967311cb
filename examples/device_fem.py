"""Device-resident stand-in FE loop: the consumer of the constitutive engine that keeps displacement, stress and
tangent on the GPU (``examples/hex_fem.py`` is the host-side one).

What the reference does around ``QuadratureMap.update`` -- dolfinx assembly of ``dot(sig, strain(v)) * dx`` and of its
derivative, a PETSc SNES Newton loop with a Krylov solver (``solvers.py:31-96, :182-196``,
``finite_strain_elastoplasticity.py:192-200``) -- is restated matrix-free on the structured hex8 mesh of BASELINE
config 5: per Newton iteration

    displacement (device) --dxm_integrate_displacement_device--> stress (N,6) + tangent COEFFICIENTS (N,9)   [the hot path]
    residual  r = sum_q w detJ B^T sigma                                                [libdxmfem: internal force]
    K du = -r  by preconditioned conjugate gradients, K p = sum_q w detJ B^T Ct B p     [libdxmfem: tangent apply]

so that no (N,6,6) array exists anywhere and nothing but scalars crosses PCIe.  Preconditioner: the diagonal of K
(``Hex8Operators.tangent_diagonal_device``) or a geometric multigrid V-cycle whose coarse operators are the same matrix-free
kernels on coarser meshes with cell-averaged tangents (``Multigrid`` below).  Vectors are torch tensors; torch is
plumbing (axpy, dot); the assembly-side operators are the kernels of ``examples/csrc/dxmfem.hip`` (``fem_operators.py``), built
beside this file -- they are not part of ``libdxmat.so``, whose scope ends with the constitutive update.

Not part of the product package and not a re-implementation of dolfinx / PETSc.
"""
from __future__ import annotations

import time

import numpy as np
import torch

CORNERS = np.array([(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)])


def structured_hex_mesh(n):
    """Unit cube, n^3 trilinear cells: coords ((n+1)^3, 3), conn (n^3, 8) int32 in the corner order of
    ``dxm_mesh_create_hex8``; node (i, j, k) has index (i (n+1) + j)(n+1) + k, cell (i, j, k) index (i n + j) n + k."""
    g = np.arange(n + 1) / n
    coords = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    m = n + 1
    i, j, k = (a.ravel().astype(np.int32) for a in np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"))
    conn = np.stack([((i + a) * m + (j + b)) * m + (k + c) for a, b, c in CORNERS], axis=1).astype(np.int32)
    return coords, conn


def uniaxial_masks(n, device):
    """(free mask (ndof,) of 0/1 doubles, index tensor of the u_x dofs on x = 1): symmetry planes x = 0, y = 0, z = 0
    clamped in their normal direction, u_x imposed on x = 1 (``examples/uniaxial_tension_3d.py``)."""
    m = n + 1
    free = torch.ones((m, m, m, 3), dtype=torch.float64, device=device)
    free[0, :, :, 0] = 0.0
    free[-1, :, :, 0] = 0.0
    free[:, 0, :, 1] = 0.0
    free[:, :, 0, 2] = 0.0
    loaded = torch.zeros((m, m, m, 3), dtype=torch.bool, device=device)
    loaded[-1, :, :, 0] = True
    return free.reshape(-1), loaded.reshape(-1)


class Level:
    """One mesh of the hierarchy: the device mesh, its free-dof mask and the tangent it applies."""

    def __init__(self, n, device):
        from fem_operators import Hex8Operators

        self.n = n
        coords, conn = structured_hex_mesh(n)
        self._coords_conn = (coords, conn)
        self._mesh = None
        self.ops = Hex8Operators(coords, conn, device=device.index or 0)      # examples/libdxmfem.so: the assembly-side kernels
        self.ndof = 3 * len(coords)
        self.free, _ = uniaxial_masks(n, device)
        self.device = device
        self.tangent = None      # device tensor (npoints, 9) or (npoints, 36)
        self.layout = "coef"
        self.wdinv = None        # damped inverse diagonal of the masked operator
        self.applies = 0

    @property
    def mesh(self):
        """The library's mesh handle (gradient evaluation inside the update kernel): only the finest level needs one."""
        if self._mesh is None:
            from dolfinx_materials_amd.gradient import Hex8Mesh

            self._mesh = Hex8Mesh(*self._coords_conn, device=self.device.index or 0)
        return self._mesh

    def apply(self, x, out=None):
        """y = M K M x + (1 - M) x for the level's tangent (x is expected to vanish on the constrained dofs)."""
        y = torch.empty_like(x) if out is None else out
        st = torch.cuda.current_stream().cuda_stream
        self.ops.tangent_apply_device(self.tangent.data_ptr(), x.data_ptr(), y.data_ptr(), layout=self.layout, stream=st)
        self.applies += 1
        y.mul_(self.free)
        return y


def _cell_mean_full(tangent_full, n):
    """(n^3 * 8, 36) point tangents -> (n^3, 36) cell means -> ((n/2)^3, 36) means over the 2x2x2 children."""
    cell = tangent_full.reshape(n, n, n, 8, 36).mean(dim=3)
    h = n // 2
    return cell.reshape(h, 2, h, 2, h, 2, 36).mean(dim=(1, 3, 5)).reshape(-1, 36)


class Multigrid:
    """Geometric multigrid V-cycle on the structured grid as the CG preconditioner, matrix-free on every level.

    Level 0 applies the actual tangent (coefficient layout).  Level l > 0 is the same operator kernel on the mesh of
    n / 2^l cells with, in every coarse cell, the mean of the tangent blocks of its children (full layout) -- a
    rediscretisation with homogenised coefficients instead of the Galerkin products of the host version, which would
    need assembled matrices (for nested trilinear spaces and cell-wise constant coefficients the two coincide).
    Smoother: damped Jacobi with the true diagonal of each level, the damping taken from a power-iteration estimate of
    the largest eigenvalue of D^-1 A (the plastic tangent is strongly anisotropic).  The coarsest level is solved by
    Jacobi-preconditioned CG to 1e-3; that makes the V-cycle a slightly varying map, so the outer CG is the flexible
    (Polak-Ribiere) variant."""

    def __init__(self, n, device, coarsest=8, sweeps=2):
        self.levels = [Level(n, device)]
        while self.levels[-1].n % 2 == 0 and self.levels[-1].n // 2 >= coarsest:
            self.levels.append(Level(self.levels[-1].n // 2, device))
        self.sweeps = sweeps
        self.device = device

    def set_tangent(self, coef):
        """New fine-level coefficients (after a constitutive update): coarse tangents, diagonals, damping factors."""
        from dolfinx_materials_amd import _lib

        lib = _lib.load()
        st = torch.cuda.current_stream().cuda_stream
        fine = self.levels[0]
        fine.tangent, fine.layout = coef, "coef"
        if len(self.levels) > 1:
            full = torch.empty((coef.shape[0], 36), dtype=torch.float64, device=self.device)
            _lib.check(lib.dxm_expand_tangent_device(coef.data_ptr(), coef.shape[0], full.data_ptr(), self.device.index or 0, st or None), lib)
            n = fine.n
            for lvl in self.levels[1:]:
                cell = _cell_mean_full(full, n)                       # (n/2)^3 cells
                full = cell.repeat_interleave(8, dim=0).contiguous()   # the 8 points of a coarse cell share it
                lvl.tangent, lvl.layout = full, "full"
                n //= 2
        gen = torch.Generator(device=self.device).manual_seed(0)
        for lvl in self.levels:
            d = torch.empty(lvl.ndof, dtype=torch.float64, device=self.device)
            if lvl.layout == "coef":
                lvl.ops.tangent_diagonal_device(lvl.tangent.data_ptr(), d.data_ptr(), st)
            else:   # diagonal of a full-layout operator: probe-free, from its action on the coordinate colouring
                d = _diagonal_by_colouring(lvl)
            d = d * lvl.free + (1.0 - lvl.free)
            dinv = 1.0 / d
            v = torch.randn(lvl.ndof, dtype=torch.float64, device=self.device, generator=gen) * lvl.free
            rho = torch.ones((), dtype=torch.float64, device=self.device)
            for _ in range(12):
                v = dinv * lvl.apply(v)
                rho = torch.linalg.vector_norm(v)
                v = v / rho
            lvl.wdinv = (4.0 / (3.0 * 1.1 * rho)) * dinv * lvl.free

    # ---- transfers on the structured grid (per displacement component; trilinear) ---------------------------
    @staticmethod
    def _prolong_axis(c, axis):
        nf = 2 * (c.shape[axis] - 1) + 1
        shape = list(c.shape)
        shape[axis] = nf
        f = torch.empty(shape, dtype=c.dtype, device=c.device)
        sl = [slice(None)] * c.dim()
        sl[axis] = slice(0, None, 2)
        f[tuple(sl)] = c
        lo, hi = [slice(None)] * c.dim(), [slice(None)] * c.dim()
        lo[axis], hi[axis] = slice(0, -1), slice(1, None)
        sl[axis] = slice(1, None, 2)
        f[tuple(sl)] = 0.5 * (c[tuple(lo)] + c[tuple(hi)])
        return f

    @staticmethod
    def _restrict_axis(f, axis):
        sl = [slice(None)] * f.dim()
        sl[axis] = slice(0, None, 2)
        c = f[tuple(sl)].clone()
        odd = [slice(None)] * f.dim()
        odd[axis] = slice(1, None, 2)
        o = f[tuple(odd)]
        lo, hi = [slice(None)] * f.dim(), [slice(None)] * f.dim()
        lo[axis], hi[axis] = slice(0, -1), slice(1, None)
        c[tuple(lo)] += 0.5 * o
        c[tuple(hi)] += 0.5 * o
        return c

    def prolong(self, xc, nc):
        c = xc.reshape(nc + 1, nc + 1, nc + 1, 3)
        for ax in range(3):
            c = self._prolong_axis(c, ax)
        return c.reshape(-1)

    def restrict(self, xf, nf):
        f = xf.reshape(nf + 1, nf + 1, nf + 1, 3)
        for ax in range(3):
            f = self._restrict_axis(f, ax)
        return f.reshape(-1)

    def _smooth(self, lvl, b, x):
        for _ in range(self.sweeps):
            x = x + lvl.wdinv * (b - lvl.apply(x))
        return x

    def vcycle(self, b, k=0):
        lvl = self.levels[k]
        if k == len(self.levels) - 1:
            return pcg(lvl.apply, b, lvl.wdinv, rtol=1e-3, maxit=200, check_every=10)[0] if len(self.levels) > 1 else lvl.wdinv * b
        x = lvl.wdinv * b                                   # first sweep from x = 0
        for _ in range(self.sweeps - 1):
            x = x + lvl.wdinv * (b - lvl.apply(x))
        r = (b - lvl.apply(x)) * lvl.free
        nxt = self.levels[k + 1]
        xc = self.vcycle(self.restrict(r, lvl.n) * nxt.free, k + 1)
        x = x + self.prolong(xc, nxt.n) * lvl.free
        return self._smooth(lvl, b, x)


def _diagonal_by_colouring(lvl):
    """Diagonal of a level's operator from 81 applications: nodes whose indices agree modulo 3 in every direction never
    share a cell, so a vector that is 1 on one such colour and one component returns the diagonal entries there."""
    m = lvl.n + 1
    d = torch.zeros((m, m, m, 3), dtype=torch.float64, device=lvl.device)
    x = torch.zeros_like(d)
    y = torch.empty(lvl.ndof, dtype=torch.float64, device=lvl.device)
    st = torch.cuda.current_stream().cuda_stream
    for a in range(3):
        for b in range(3):
            for c in range(3):
                for comp in range(3):
                    x.zero_()
                    x[a::3, b::3, c::3, comp] = 1.0
                    lvl.ops.tangent_apply_device(lvl.tangent.data_ptr(), x.data_ptr(), y.data_ptr(), layout=lvl.layout, stream=st)
                    d[a::3, b::3, c::3, comp] = y.reshape(m, m, m, 3)[a::3, b::3, c::3, comp]
    return d.reshape(-1)


def pcg(apply, b, precond, rtol=1e-8, maxit=5000, check_every=25, flexible=False):
    """Preconditioned conjugate gradients on device tensors; ``precond`` is an inverse-diagonal tensor or a callable.
    The residual norm is looked at (one host synchronisation) every ``check_every`` iterations only.  ``flexible``:
    Polak-Ribiere update, for a preconditioner that is not a fixed linear map (an inner iterative solve)."""
    M = precond if callable(precond) else (lambda r: precond * r)
    x = torch.zeros_like(b)
    r = b.clone()
    z = M(r)
    p = z.clone()
    rz = torch.dot(r, z)
    bnorm = float(torch.linalg.vector_norm(b))
    if bnorm == 0.0:
        return x, 0, 0.0
    its, rel = 0, 1.0
    Ap = torch.empty_like(b)
    while its < maxit:
        Ap = apply(p, out=Ap)
        alpha = rz / torch.dot(p, Ap)
        x.add_(alpha * p)
        r_old = r.clone() if flexible else None
        r.sub_(alpha * Ap)
        its += 1
        if its % check_every == 0 or its == maxit:
            rel = float(torch.linalg.vector_norm(r)) / bnorm
            if rel < rtol:
                break
        z = M(r)
        rz_new = torch.dot(r, z) if not flexible else torch.dot(r - r_old, z)
        beta = rz_new / rz
        rz = torch.dot(r, z) if flexible else rz_new
        p = z + beta * p
    else:
        rel = float(torch.linalg.vector_norm(r)) / bnorm
    return x, its, rel


class DeviceProblem:
    """Uniaxial tension of the unit cube, everything resident on one GPU."""

    def __init__(self, n, material, device, preconditioner="mg", coarsest=8):
        self.n, self.material, self.device = n, material, device
        self.mg = Multigrid(n, device, coarsest=coarsest) if preconditioner == "mg" else None
        self.fine = self.mg.levels[0] if self.mg else Level(n, device)
        self.mesh = self.fine.mesh
        self.npoints = self.mesh.npoints
        material.set_data_manager(self.npoints)
        self.ndof = self.fine.ndof
        self.free, self.loaded = uniaxial_masks(n, device)
        self.u = torch.zeros(self.ndof, dtype=torch.float64, device=device)
        self.flux = torch.empty((self.npoints, 6), dtype=torch.float64, device=device)
        self.coef = torch.empty((self.npoints, 9), dtype=torch.float64, device=device)
        self.r = torch.empty(self.ndof, dtype=torch.float64, device=device)
        self._have_tangent = False
        self.timers = {"constitutive": 0.0, "residual": 0.0, "preconditioner_setup": 0.0, "solve": 0.0, "newton_iterations": 0,
                       "cg_iterations": 0, "operator_applications": 0}

    def _timed(self, key, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        self.timers[key] += time.perf_counter() - t0
        return out

    def update(self):
        st = torch.cuda.current_stream().cuda_stream
        self.material.integrate_displacement_device(self.mesh, self.u.data_ptr(), self.flux.data_ptr(), self.coef.data_ptr(), st)

    def residual(self):
        st = torch.cuda.current_stream().cuda_stream
        self.fine.ops.internal_force_device(self.flux.data_ptr(), self.r.data_ptr(), st)
        self.r.mul_(self.free)
        return float(torch.linalg.vector_norm(self.r))

    def _preconditioner(self):
        if self.mg:
            self._timed("preconditioner_setup", lambda: self.mg.set_tangent(self.coef))
            return self.mg.vcycle

        def setup():
            self.fine.tangent, self.fine.layout = self.coef, "coef"
            d = torch.empty(self.ndof, dtype=torch.float64, device=self.device)
            self.fine.ops.tangent_diagonal_device(self.coef.data_ptr(), d.data_ptr(), torch.cuda.current_stream().cuda_stream)
            return self.free / (d * self.free + (1.0 - self.free))
        return self._timed("preconditioner_setup", setup)

    def _solve(self, rhs, cg_rtol):
        a0 = self.fine.applies
        M = self._preconditioner()
        du, its, rel = self._timed("solve", lambda: pcg(self.fine.apply, rhs, M, rtol=cg_rtol, flexible=self.mg is not None,
                                                         check_every=1 if self.mg else 25))
        self.timers["cg_iterations"] += its
        self.timers["operator_applications"] += self.fine.applies - a0
        return du, its, rel

    def solve_step(self, exx, atol=1e-8, rtol=1e-10, maxit=25, cg_rtol=1e-10, log=None):
        """One load increment: linear predictor with the tangent of the last converged state, Newton iterations with
        the constitutive update as the residual callback (``solvers.py:72``), advance once converged (``solvers.py:194``).

        The predictor solves K_n du = -K_n du_bc for the free dofs, du_bc being the increment of the imposed
        displacements: putting the increment on the boundary nodes alone starts Newton from a one-cell boundary layer of
        strain exx / h (0.5 at 200^3 cells) whose points flip between the elastic and the plastic branch for dozens
        of iterations."""
        if not self._have_tangent:
            self._timed("constitutive", self.update)      # u = 0: the elastic tangent
            self._have_tangent = True
        du_bc = torch.zeros_like(self.u)
        du_bc[self.loaded] = exx - self.u[self.loaded]
        self.fine.tangent, self.fine.layout = self.coef, "coef"
        rhs = -self.fine.apply(du_bc)
        du, its, rel = self._solve(rhs, cg_rtol)
        if log:
            log(f"    predictor: cg {its} its (rel {rel:.1e})")
        self.u.add_(du_bc + du)
        norms = []
        for it in range(maxit):
            self._timed("constitutive", self.update)
            rn = self._timed("residual", self.residual)
            norms.append(rn)
            self.timers["newton_iterations"] += 1
            stats = self.material.stats()[1]
            if stats["n_nan"]:
                raise RuntimeError("NaN in the constitutive update")
            if rn < atol or (it > 0 and rn < rtol * norms[0]):
                break
            du, its, rel = self._solve(-self.r, cg_rtol)
            if log:
                log(f"    newton {it}: |r| = {rn:.3e}  cg {its} its (rel {rel:.1e})  plastic {stats['n_plastic']}")
            self.u.add_(du)
        else:
            raise RuntimeError(f"Newton did not converge: {norms}")
        self.material.data_manager.update()
        return norms
