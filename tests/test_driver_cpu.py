"""CPU tests of the host-side drivers around the hot path (QuadratureFieldMap, the field exchange of
quadrature_map.py:297-360, and the stand-in FE loop), with the oracle-backed material."""
import os
import sys

import numpy as np
import pytest

from dolfinx_materials_amd.field_map import QuadratureFieldMap
from oracle import constitutive_np as onp
from oracle_material import OracleJ2Material

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from helpers import E, NU, SIG0_LIN, H_LIN, j2_history  # noqa: E402


def _mat():
    return OracleJ2Material(E, NU, onp.LinearHardening(SIG0_LIN, H_LIN))


def test_two_maps_on_disjoint_cells_equal_one_map_on_all_cells():
    """Semantics pinned by the reference's tests/mfront/test_multimaterials.py:23-172."""
    ncell, nqp = 10, 4
    eps_all = j2_history(ncell * nqp, seed=5)[2]
    ev = lambda cells: eps_all.reshape(ncell, nqp, 6)[cells].reshape(-1, 6)  # noqa: E731
    full = QuadratureFieldMap(ncell, nqp, _mat())
    full.register_gradient("strain", ev)
    full.update()
    full.advance()
    ca, cb = np.array([0, 3, 4, 9], dtype=np.int32), np.array([1, 2, 5, 6, 7, 8], dtype=np.int32)
    out = {}
    for cells in (ca, cb):
        m = QuadratureFieldMap(ncell, nqp, _mat(), cells=cells)
        m.register_gradient("strain", ev)
        m.update()
        m.advance()
        for name, f in {**m.fluxes, **m.internal_state_variables, "jac": m.jacobian_flatten}.items():
            out[name] = out.get(name, 0) + f.x.array
    for name, f in {**full.fluxes, **full.internal_state_variables, "jac": full.jacobian_flatten}.items():
        assert np.array_equal(out[name], f.x.array), name
    assert full.material.n == ncell * nqp


def test_update_initial_state_layout():
    """tests/mfront/test_initialization.py:61-110: values are tiled per Gauss point, AoS."""
    ncell, nqp = 3, 4
    m = QuadratureFieldMap(ncell, nqp, _mat())
    m.update_initial_state("p", 0.01)
    assert np.array_equal(m.material.s0["p"], np.full((12, 1), 0.01))
    s = np.arange(6.0)
    m.update_initial_state("stress", s)
    assert np.array_equal(m.fluxes["stress"].x.array, np.tile(s, 12))
    assert np.array_equal(m.fluxes["stress"].values, np.tile(s, (12, 1)))
    with pytest.raises(ValueError):
        m.update_initial_state("strain", 0.0)
    with pytest.raises(ValueError):
        m.register_gradient("nonsense", lambda c: None)


def test_uniaxial_tension_3d_fe_loop_reaches_closed_form():
    import dolfinx_materials_amd.field_map as qd
    from hex_fem import HexMesh, newton_solve

    n = 3
    mesh = HexMesh(n)
    u = np.zeros(mesh.ndof)
    mat = _mat()
    qmap = qd.QuadratureFieldMap(mesh.num_cells, mesh.nqp, mat)
    qmap.register_gradient("strain", lambda cells: mesh.strain(u, cells))
    x0, x1, y0, z0 = mesh.nodes_on(0, 0.0), mesh.nodes_on(0, 1.0), mesh.nodes_on(1, 0.0), mesh.nodes_on(2, 0.0)
    bc_dofs = np.concatenate([3 * x0, 3 * x1, 3 * y0 + 1, 3 * z0 + 2])
    for k in range(1, 6):
        exx = 2e-2 * k / 5
        bc_vals = np.concatenate([np.zeros(len(x0)), np.full(len(x1), exx), np.zeros(len(y0)), np.zeros(len(z0))])
        norms = newton_solve(mesh, qmap, u, bc_dofs, bc_vals, mesh.B_eps, "stress")
        assert len(norms) <= 6 and norms[-1] < 1e-6 * norms[0]  # consistent tangent: quadratic convergence
    sig = qmap.fluxes["stress"].values
    expect = (SIG0_LIN + H_LIN * exx) / (1 + H_LIN / E)
    assert np.allclose(sig[:, 0], expect, rtol=1e-9)
    assert np.abs(sig[:, 1:]).max() < 1e-6
    p = qmap.internal_state_variables["p"].x.array
    assert np.allclose(p, exx - expect / E, rtol=1e-9)


def test_element_matrices_from_every_tangent_layout_agree():
    """The host assembly consumes the full (N,6,6) block, its 21-entry upper triangle and the nine coefficients of
    Ct = c1 1x1 + c2 I + c3 n x n (an assembly-side consumer of the packed tangent)."""
    from dolfinx_materials_amd.conventions import pack_sym_tangent, tangent_from_coefficients
    from hex_fem import HexMesh

    mesh = HexMesh(3)
    rng = np.random.default_rng(0)
    coef = rng.standard_normal((mesh.num_cells * 8, 9))
    coef[:, :3] = np.abs(coef[:, :3]) * [50e3, 40e3, -30e3]
    full = tangent_from_coefficients(coef)
    flux = rng.standard_normal((mesh.num_cells * 8, 6))
    r0, K0 = mesh.assemble(flux, full.reshape(-1, 36), mesh.B_eps, "full")
    r1, K1 = mesh.assemble(flux, pack_sym_tangent(full), mesh.B_eps, "sym")
    r2, K2 = mesh.assemble(flux, coef, mesh.B_eps, "coef")
    scale = np.abs(K0.data).max()
    assert np.array_equal(r0, r1) and np.array_equal(r0, r2)
    assert np.abs((K0 - K1).toarray()).max() < 1e-13 * scale and np.abs((K0 - K2).toarray()).max() < 1e-13 * scale
    # against a plain dense assembly
    Kd = np.zeros((mesh.ndof, mesh.ndof))
    for c in range(mesh.num_cells):
        Ke = sum(mesh.wdet * mesh.B_eps[q].T @ full[c * 8 + q] @ mesh.B_eps[q] for q in range(8))
        Kd[np.ix_(mesh.cell_dofs[c], mesh.cell_dofs[c])] += Ke
    assert np.abs(K0.toarray() - Kd).max() < 1e-12 * scale and np.abs(K0.toarray() - K0.toarray().T).max() < 1e-12 * scale


def test_multigrid_preconditioned_cg_solves_the_fe_loop():
    """n = 8 (2187 dofs) through the Krylov path that the 64^3 run uses: same closed-form answer, CG iteration
    counts that do not grow with the Newton iteration (the V-cycle is a mesh-independent preconditioner)."""
    import dolfinx_materials_amd.field_map as qd
    from hex_fem import HexMesh, newton_solve

    mesh = HexMesh(8)
    u = np.zeros(mesh.ndof)
    qmap = qd.QuadratureFieldMap(mesh.num_cells, mesh.nqp, _mat())
    qmap.register_gradient("strain", lambda cells: mesh.strain(u, cells))
    x0, x1, y0, z0 = mesh.nodes_on(0, 0.0), mesh.nodes_on(0, 1.0), mesh.nodes_on(1, 0.0), mesh.nodes_on(2, 0.0)
    bc_dofs = np.concatenate([3 * x0, 3 * x1, 3 * y0 + 1, 3 * z0 + 2])
    timers = {}
    for k in range(1, 4):
        exx = 1.2e-2 * k / 3
        bc_vals = np.concatenate([np.zeros(len(x0)), np.full(len(x1), exx), np.zeros(len(y0)), np.zeros(len(z0))])
        norms = newton_solve(mesh, qmap, u, bc_dofs, bc_vals, mesh.B_eps, "stress", timers=timers, solver="krylov")
        assert len(norms) <= 7
    expect = (SIG0_LIN + H_LIN * exx) / (1 + H_LIN / E)
    assert np.allclose(qmap.fluxes["stress"].values[:, 0], expect, rtol=1e-8)
    assert timers["cg_iterations"] / (timers["newton_iterations"] - 3) < 40


def test_device_loop_grid_transfers_are_trilinear_and_transposes_of_each_other():
    """examples/device_fem.py (the device-resident loop) moves between multigrid levels by slicing on the node grid; the
    pieces that need no GPU are checked here: prolongation reproduces a trilinear function exactly, restriction is its
    transpose, and cell averaging of tangents keeps a constant."""
    import torch
    from device_fem import Multigrid, _cell_mean_full, structured_hex_mesh

    nc = 3
    mg = Multigrid.__new__(Multigrid)
    coords_c, _ = structured_hex_mesh(nc)
    coords_f, conn_f = structured_hex_mesh(2 * nc)
    f = lambda x: 1.0 + x @ np.array([0.3, -0.2, 0.5]) + 0.7 * x[:, 0] * x[:, 1] * x[:, 2]   # noqa: E731  trilinear
    xc = torch.from_numpy(np.stack([f(coords_c), 2 * f(coords_c), -f(coords_c)], axis=1).reshape(-1))
    xf = mg.prolong(xc, nc)
    assert np.abs(xf.numpy().reshape(-1, 3)[:, 0] - f(coords_f)).max() < 1e-14
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.standard_normal(xc.shape[0]))
    b = torch.from_numpy(rng.standard_normal(xf.shape[0]))
    assert abs(float(torch.dot(mg.prolong(a, nc), b) - torch.dot(a, mg.restrict(b, 2 * nc)))) < 1e-12
    assert conn_f.shape == (216, 8) and conn_f.dtype == np.int32 and conn_f.max() == len(coords_f) - 1
    t = torch.ones((216 * 8, 36), dtype=torch.float64) * 3.0
    assert torch.equal(_cell_mean_full(t, 6), torch.full((27, 36), 3.0, dtype=torch.float64))
