"""CPU tests of the host-side drivers around the hot path (QuadratureFieldMap replaying
quadrature_map.py:297-360, and the stand-in FE loop), with the oracle-backed material."""
import os
import sys

import numpy as np
import pytest

from dolfinx_materials_amd.quadrature_driver import QuadratureFieldMap, _get_vals
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
    assert np.array_equal(_get_vals(m.fluxes["stress"]), np.tile(s, (12, 1)))
    with pytest.raises(ValueError):
        m.update_initial_state("strain", 0.0)
    with pytest.raises(ValueError):
        m.register_gradient("nonsense", lambda c: None)


def test_uniaxial_tension_3d_fe_loop_reaches_closed_form():
    import dolfinx_materials_amd.quadrature_driver as qd
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
    sig = _get_vals(qmap.fluxes["stress"])
    expect = (SIG0_LIN + H_LIN * exx) / (1 + H_LIN / E)
    assert np.allclose(sig[:, 0], expect, rtol=1e-9)
    assert np.abs(sig[:, 1:]).max() < 1e-6
    p = qmap.internal_state_variables["p"].x.array
    assert np.allclose(p, exx - expect / E, rtol=1e-9)
