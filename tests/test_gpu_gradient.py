"""GPU: device gradient evaluation (hex8, tet4, Lagrange simplices) against a host evaluation, and the
displacement-driven integrate against the strain-driven one."""
import os
import sys

import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd.gradient import Hex8Mesh, gauss_points_hex
from dolfinx_materials_amd.jaxmat import JAXMaterial

pytestmark = pytest.mark.gpu
from helpers import to_device, to_host  # noqa: E402,F401

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
SQ2 = np.sqrt(2.0)
S = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], float)


def host_gradient(coords, conn, u, qp):
    """Plain numpy isoparametric displacement gradient H (ncells, nqp, 3, 3)."""
    X = coords[conn]                    # (c, 8, 3)
    U = u.reshape(-1, 3)[conn]          # (c, 8, 3)
    out = np.empty((conn.shape[0], len(qp), 3, 3))
    for q, xi in enumerate(qp):
        dN = np.empty((8, 3))
        for m in range(8):
            for d in range(3):
                f = 0.125 * S[m, d]
                for o in range(3):
                    if o != d:
                        f *= 1 + S[m, o] * xi[o]
                dN[m, d] = f
        J = np.einsum("cma,md->cad", X, dN)      # dX_a/dxi_d
        Ji = np.linalg.inv(J)                      # dxi_d/dX_a
        g = np.einsum("md,cda->cma", dN, Ji)      # dN_m/dX_a
        out[:, q] = np.einsum("cmi,cma->cia", U, g)
    return out


def make_mesh(n, distort=0.15, seed=0):
    from hex_fem import HexMesh

    m = HexMesh(n)
    rng = np.random.default_rng(seed)
    coords = m.coords + distort * m.h * rng.uniform(-1, 1, m.coords.shape)
    return m, coords


def test_device_gradient_matches_host_on_distorted_mesh():
    torch = pytest.importorskip("torch")
    m, coords = make_mesh(5)
    rng = np.random.default_rng(1)
    u = 1e-2 * rng.standard_normal(m.ndof)
    qp = gauss_points_hex(2)
    H = host_gradient(coords, m.conn, u, qp).reshape(-1, 3, 3)
    mesh = Hex8Mesh(coords, m.conn)
    dev = torch.device("cuda:0")
    ud = to_device(u)
    eps = torch.empty((mesh.npoints, 6), dtype=torch.float64, device=dev)
    F = torch.empty((mesh.npoints, 9), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    mesh.gradient_device(ud.data_ptr(), 0, eps.data_ptr(), st)
    mesh.gradient_device(ud.data_ptr(), 1, F.data_ptr(), st)
    torch.cuda.synchronize()
    e = 0.5 * (H + H.transpose(0, 2, 1))
    eps_ref = np.stack([e[:, 0, 0], e[:, 1, 1], e[:, 2, 2], SQ2 * e[:, 0, 1], SQ2 * e[:, 0, 2], SQ2 * e[:, 1, 2]], axis=1)
    Fm = np.eye(3) + H
    F_ref = np.stack([Fm[:, 0, 0], Fm[:, 1, 1], Fm[:, 2, 2], Fm[:, 0, 1], Fm[:, 1, 0], Fm[:, 0, 2], Fm[:, 2, 0], Fm[:, 1, 2], Fm[:, 2, 1]], axis=1)
    assert np.abs(to_host(eps) - eps_ref).max() < 1e-13
    assert np.abs(to_host(F) - F_ref).max() < 1e-13


def test_uniform_mesh_matches_the_fe_driver_b_matrices():
    """On uniform cubes the device evaluation equals HexMesh.strain of examples/hex_fem.py
    (same Gauss point order)."""
    torch = pytest.importorskip("torch")
    m, _ = make_mesh(4, distort=0.0)
    u = 1e-3 * np.random.default_rng(2).standard_normal(m.ndof)
    mesh = Hex8Mesh(m.coords, m.conn)
    dev = torch.device("cuda:0")
    eps = torch.empty((mesh.npoints, 6), dtype=torch.float64, device=dev)
    mesh.gradient_device(to_device(u).data_ptr(), 0, eps.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.abs(to_host(eps) - m.strain(u, np.arange(m.num_cells))).max() < 1e-14


@pytest.mark.parametrize("law", ["j2", "fefp"])
def test_integrate_displacement_equals_integrate_of_host_gradient(law):
    m, coords = make_mesh(4)
    rng = np.random.default_rng(3)
    u = 2e-2 * rng.standard_normal(m.ndof) * m.h
    qp = gauss_points_hex(2)
    H = host_gradient(coords, m.conn, u, qp).reshape(-1, 3, 3)
    el = jm.LinearElasticIsotropic(E=70e3, nu=0.3)
    if law == "j2":
        beh = jm.vonMisesIsotropicHardening(el, jm.LinearHardening(250.0, 5e3))
        e = 0.5 * (H + H.transpose(0, 2, 1))
        g = np.stack([e[:, 0, 0], e[:, 1, 1], e[:, 2, 2], SQ2 * e[:, 0, 1], SQ2 * e[:, 0, 2], SQ2 * e[:, 1, 2]], axis=1)
    else:
        beh = jm.FeFpJ2Plasticity(el, jm.VoceHardening(500.0, 750.0, 1000.0))
        Fm = np.eye(3) + H
        g = np.stack([Fm[:, 0, 0], Fm[:, 1, 1], Fm[:, 2, 2], Fm[:, 0, 1], Fm[:, 1, 0], Fm[:, 0, 2], Fm[:, 2, 0], Fm[:, 1, 2], Fm[:, 2, 1]], axis=1)
    mesh = Hex8Mesh(coords, m.conn)
    a, b = JAXMaterial(beh), JAXMaterial(beh)
    a.set_data_manager(mesh.npoints)
    b.set_data_manager(mesh.npoints)
    fa, ia, ca = a.integrate_displacement(mesh, u)
    fb, ib, cb = b.integrate(g)
    assert a.last_stats["n_plastic"] > 0
    scale = np.abs(fb).max()
    assert np.abs(fa - fb).max() < 1e-9 * scale and np.abs(ca - cb).max() < 1e-9 * np.abs(cb).max()
    assert np.abs(ia - ib).max() < 1e-12


def test_tet4_gradient_matches_host_and_drives_the_update():
    """Linear tetrahedra (each cube of the hex mesh split into 6): device gradient == host affine
    gradient, repeated at the cell's 4 Gauss points; integrate_displacement == integrate."""
    torch = pytest.importorskip("torch")
    from dolfinx_materials_amd.gradient import Tet4Mesh

    m, coords = make_mesh(3, distort=0.2, seed=4)
    # Kuhn split of every hexahedron (corner order of HexMesh.conn) into 6 tetrahedra along 0-6
    conn = np.concatenate([m.conn[:, list(k)] for k in KUHN], axis=0).astype(np.int32)
    rng = np.random.default_rng(8)
    u = 2e-2 * m.h * rng.standard_normal(m.ndof)
    X, U = coords[conn], u.reshape(-1, 3)[conn]
    A = (X[:, 1:] - X[:, :1]).transpose(0, 2, 1)           # dX_a/dxi_d
    dU = (U[:, 1:] - U[:, :1]).transpose(0, 2, 1)          # du_i/dxi_d
    H = dU @ np.linalg.inv(A)
    nqp = 4
    Hq = np.repeat(H, nqp, axis=0)
    e = 0.5 * (Hq + Hq.transpose(0, 2, 1))
    eps_ref = np.stack([e[:, 0, 0], e[:, 1, 1], e[:, 2, 2], SQ2 * e[:, 0, 1], SQ2 * e[:, 0, 2], SQ2 * e[:, 1, 2]], axis=1)
    mesh = Tet4Mesh(coords, conn, nqp=nqp)
    dev = torch.device("cuda:0")
    eps = torch.empty((mesh.npoints, 6), dtype=torch.float64, device=dev)
    mesh.gradient_device(to_device(u).data_ptr(), 0, eps.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.abs(to_host(eps) - eps_ref).max() < 1e-13
    beh = jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=70e3, nu=0.3), jm.LinearHardening(250.0, 5e3))
    a, b = JAXMaterial(beh), JAXMaterial(beh)
    a.set_data_manager(mesh.npoints)
    b.set_data_manager(mesh.npoints)
    fa, _, ca = a.integrate_displacement(mesh, u)
    fb, _, cb = b.integrate(eps_ref)
    assert a.last_stats["n_plastic"] > 0
    assert np.abs(fa - fb).max() < 1e-9 * np.abs(fb).max() and np.abs(ca - cb).max() < 1e-9 * np.abs(cb).max()


KUHN = [(0, 1, 2, 6), (0, 2, 3, 6), (0, 3, 7, 6), (0, 7, 4, 6), (0, 4, 5, 6), (0, 5, 1, 6)]   # hex -> 6 tets along 0-6


def _simplex_case(name):
    """(SimplexMesh, dof coordinates (n_dofs, tdim), host arrays for the checker)."""
    from dolfinx_materials_amd.gradient import SimplexMesh, lagrange_simplex_table, p2_dofmap, simplex_quadrature
    from helpers import triangle_grid

    if name.endswith("tri"):
        coords, cells = triangle_grid(7, seed=3)                       # 98 triangles
    else:
        hm, coords = make_mesh(3, distort=0.2, seed=4)
        cells = np.concatenate([hm.conn[:, list(k)] for k in KUHN], axis=0).astype(np.int32)   # 162 tetrahedra
    degree = int(name[1])
    mesh, xd = SimplexMesh.lagrange(coords, cells, degree=degree, quadrature_degree=2)
    tdim = cells.shape[1] - 1
    dofmap = cells if degree == 1 else p2_dofmap(cells)[0]
    dphi = lagrange_simplex_table(tdim, degree, simplex_quadrature(tdim, 2))
    return mesh, xd, (np.pad(coords, ((0, 0), (0, 3 - coords.shape[1]))), cells, dofmap, dphi)


@pytest.mark.parametrize("name", ["p2tet", "p2tri", "p1tet", "p1tri"])
def test_simplex_gradient_matches_host(name):
    """dxm_mesh_create_simplex: tet10 x 4 points and tri6 x 3 points (plane strain) of the reference's demos, and the
    first-order cases, against the numpy evaluation; both gradient kinds."""
    torch = pytest.importorskip("torch")
    from helpers import deformation_gradient9, mandel_strain, simplex_host_gradient

    mesh, xd, host = _simplex_case(name)
    tdim = xd.shape[1]
    assert mesh.nqp == (3 if tdim == 2 else 4) and mesh.displacement_size == xd.size
    rng = np.random.default_rng(11)
    u = (xd * np.array([8e-3, -3e-3, -3e-3][:tdim]) + 1e-3 * rng.standard_normal(xd.shape) + 5e-3 * xd**2).ravel()
    H = simplex_host_gradient(*host[:3], u, host[3]).reshape(-1, 3, 3)
    dev = torch.device("cuda:0")
    ud = to_device(u)
    st = torch.cuda.current_stream().cuda_stream
    for kind, ref in ((0, mandel_strain(H)), (1, deformation_gradient9(H))):
        out = torch.full((mesh.npoints, ref.shape[1]), float("nan"), dtype=torch.float64, device=dev)
        mesh.gradient_device(ud.data_ptr(), kind, out.data_ptr(), st)
        torch.cuda.synchronize()
        assert np.abs(to_host(out) - ref).max() < 1e-13
    if name == "p1tet":   # same numbers as the dedicated tet4 kernel
        from dolfinx_materials_amd.gradient import Tet4Mesh

        t4 = Tet4Mesh(host[0], host[1], nqp=4)
        o2 = torch.empty((mesh.npoints, 6), dtype=torch.float64, device=dev)
        t4.gradient_device(ud.data_ptr(), 0, o2.data_ptr(), st)
        torch.cuda.synchronize()
        assert np.abs(to_host(o2) - mandel_strain(H)).max() < 1e-13


def test_simplex_mesh_rejects_bad_input():
    from dolfinx_materials_amd import _lib
    from dolfinx_materials_amd.gradient import SimplexMesh, lagrange_simplex_table, simplex_quadrature

    coords = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    cells = np.array([[0, 1, 2, 3]], dtype=np.int32)
    tab = lagrange_simplex_table(3, 1, simplex_quadrature(3, 1))
    SimplexMesh(coords, cells, cells, 4, tab).close()
    flat = coords.copy()
    flat[3] = [1.0, 1.0, 0.0]
    with pytest.raises(_lib.DxmError, match="degenerate"):
        SimplexMesh(flat, cells, cells, 4, tab)
    with pytest.raises(_lib.DxmError, match="out of range"):
        SimplexMesh(coords, cells, cells + 1, 4, tab)
    with pytest.raises(ValueError):
        SimplexMesh(coords, cells, cells, 4, tab[:, :3])


@pytest.mark.parametrize("law", ["elastic", "j2_voce", "j2_linear_sym", "fefp"])
@pytest.mark.parametrize("cells", ["hex3", "hex5", "tet3x1", "tet4x4", "hex3x27", "p2tet", "p2tri"])
def test_integrate_displacement_device_equals_gradient_then_update(law, cells):
    """dxm_integrate_displacement_device (hex8 x 8 points, tet4 and Lagrange-simplex meshes: gradient evaluated inside
    the update kernel, no strain / F array; hex8 with 27 points: two kernels through the handle's scratch)
    against the explicit sequence dxm_mesh_gradient_device -> dxm_integrate_device, over two increments
    with an advance between."""
    torch = pytest.importorskip("torch")
    from dolfinx_materials_amd.gradient import Tet4Mesh
    from helpers import E, NU, SIG0_V, SIGU_V, B_V, SIG0_F, SIGU_F, B_F, SIG0_LIN, H_LIN

    dev = torch.device("cuda:0")
    if cells.startswith("p2"):
        mesh, coords, _ = _simplex_case(cells)     # `coords`: the dof positions (n_dofs, tdim)
    else:
        hm, coords = make_mesh(int(cells[3]))
        conn = hm.conn
    if cells.startswith("p2"):
        pass
    elif cells.startswith("tet"):
        tconn = np.concatenate([conn[:, list(k)] for k in KUHN], axis=0).astype(np.int32)
        mesh = Tet4Mesh(coords, tconn, nqp=int(cells.split("x")[1]))   # 162 x 1 and 384 x 4 points
    elif cells.endswith("x27"):
        mesh = Hex8Mesh(coords, conn, qpoints=gauss_points_hex(4))     # 27 points per cell: not fusable
    else:
        mesh = Hex8Mesh(coords, conn)
    n = mesh.npoints            # hex: 27 * 8 = 216 (ragged last tile) and 125 * 8 = 1000
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    kw = {}
    if law == "elastic":
        mk, kind, scale = (lambda: jm.ElasticBehavior(el)), 0, 5e-3
    elif law == "j2_voce":
        mk, kind, scale = (lambda: jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V))), 0, 8e-3
    elif law == "j2_linear_sym":
        mk, kind, scale = (lambda: jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))), 0, 8e-3
        kw = {"tangent_layout": "sym"}
    else:
        mk, kind, scale = (lambda: jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F))), 1, 2e-2
    rng = np.random.default_rng(3)
    st = torch.cuda.current_stream().cuda_stream
    a, b = JAXMaterial(mk(), **kw), JAXMaterial(mk(), **kw)
    a.set_data_manager(n)
    b.set_data_manager(n)
    ng, nf = a._info.n_grad, a._info.n_flux
    nt = 21 if kw else nf * ng
    grad = torch.empty((n, ng), dtype=torch.float64, device=dev)
    fa, fb = torch.empty((n, nf), dtype=torch.float64, device=dev), torch.empty((n, nf), dtype=torch.float64, device=dev)
    ca, cb = torch.empty((n, nt), dtype=torch.float64, device=dev), torch.empty((n, nt), dtype=torch.float64, device=dev)
    for t in (0.6, 1.0):
        u = t * (coords * np.array([scale, -0.4 * scale, -0.4 * scale][:coords.shape[1]]) + rng.standard_normal(coords.shape) * 0.05 * scale)
        ud = to_device(u.ravel().copy())
        mesh.gradient_device(ud.data_ptr(), kind, grad.data_ptr(), st)
        a.integrate_device(grad.data_ptr(), fa.data_ptr(), ca.data_ptr(), st)
        b.integrate_displacement_device(mesh, ud.data_ptr(), fb.data_ptr(), cb.data_ptr(), st)
        torch.cuda.synchronize()
        sa, sb = a.stats()[1], b.stats()[1]
        assert sa == sb and sa["n_nan"] == 0
        if law != "elastic":
            assert sa["n_plastic"] > 0
        # the in-kernel gradient uses the same arithmetic; fused-multiply-add contraction may differ by an ulp
        scale_f = float(fa.abs().max())
        assert float((fa - fb).abs().max()) <= 1e-12 * scale_f
        assert float((ca - cb).abs().max()) <= 1e-12 * float(ca.abs().max())
        fin_a, fin_b = a.get_final_state_dict(), b.get_final_state_dict()
        for k, v in fin_a.items():
            if k in a.gradients or k in a.fluxes:   # device-pointer forms: the host never saw them ("unknown" placeholders)
                assert np.isnan(v).all() and np.isnan(fin_b[k]).all(), k
                continue
            assert np.abs(v - fin_b[k]).max() <= 1e-12 * max(np.abs(v).max(), 1e-300), k
        a.data_manager.update()
        b.data_manager.update()


@pytest.mark.parametrize("law", ["j2", "fefp", "j2_p2tet", "fefp_p2tet"])
def test_chunked_host_displacement_path_uses_the_right_cells(law):
    """Above 524288 points the host-buffer form is cut into chunks issued on two streams; with the gradient
    evaluated inside the update kernel every chunk must start at its own point (MeshSource.point0)."""
    torch = pytest.importorskip("torch")
    p2 = law.endswith("_p2tet")
    law = law.split("_")[0]
    m, coords = make_mesh(28 if p2 else 42)            # hex8: 74 088 cells, 592 704 points -> 2 chunks
    if p2:                                             # tet10: 131 712 cells x 4 points = 526 848 -> 2 chunks
        from dolfinx_materials_amd.gradient import SimplexMesh

        tets = np.concatenate([m.conn[:, list(k)] for k in KUHN], axis=0).astype(np.int32)
        smesh, coords = SimplexMesh.lagrange(coords, tets, degree=2)
    rng = np.random.default_rng(5)
    u = (coords * np.array([8e-3, -3e-3, -3e-3]) + 2e-4 * rng.standard_normal(coords.shape)).ravel()
    el = jm.LinearElasticIsotropic(E=70e3, nu=0.3)
    beh = (jm.vonMisesIsotropicHardening(el, jm.LinearHardening(250.0, 5e3)) if law == "j2"
           else jm.FeFpJ2Plasticity(el, jm.VoceHardening(500.0, 750.0, 1000.0)))
    mesh = smesh if p2 else Hex8Mesh(coords, m.conn)
    n = mesh.npoints
    assert n > 2 * 262144
    a, b = JAXMaterial(beh), JAXMaterial(beh)
    a.set_data_manager(n)
    b.set_data_manager(n)
    fa, ia, ca = a.integrate_displacement(mesh, u)              # host path, chunked, fused
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    ng, nf = b._info.n_grad, b._info.n_flux
    ud = to_device(u.copy())
    grad = torch.empty((n, ng), dtype=torch.float64, device=dev)
    fb = torch.empty((n, nf), dtype=torch.float64, device=dev)
    cb = torch.empty((n, nf * ng), dtype=torch.float64, device=dev)
    mesh.gradient_device(ud.data_ptr(), 0 if ng == 6 else 1, grad.data_ptr(), st)   # gradient kernel -> update kernel
    b.integrate_device(grad.data_ptr(), fb.data_ptr(), cb.data_ptr(), st)
    torch.cuda.synchronize()
    assert a.last_stats["n_plastic"] > 0 and a.last_stats["n_plastic"] == b.stats()[1]["n_plastic"]
    fbh, cbh = to_host(fb), to_host(cb).reshape(ca.shape)
    assert np.abs(fa - fbh).max() <= 1e-12 * np.abs(fbh).max()
    assert np.abs(ca - cbh).max() <= 1e-12 * np.abs(cbh).max()


def test_fused_displacement_path_is_graph_capturable():
    """dxm_integrate_displacement_device on a fusable mesh is one kernel launch without allocation or
    synchronisation: a Newton cadence (new displacement vector -> update) can be captured in a HIP graph and
    replayed after the displacement buffer has been overwritten in place."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    hm, coords = make_mesh(6)
    mesh = Hex8Mesh(coords, hm.conn)
    n = mesh.npoints
    el = jm.LinearElasticIsotropic(E=70e3, nu=0.3)
    mat = JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.LinearHardening(250.0, 5e3)))
    mat.set_data_manager(n)
    rng = np.random.default_rng(11)
    u1 = (coords * np.array([6e-3, -2e-3, -2e-3]) + 1e-4 * rng.standard_normal(coords.shape)).ravel()
    u2 = 1.5 * u1
    ud = to_device(u1.copy())
    f = torch.zeros((n, 6), dtype=torch.float64, device=dev)
    c = torch.zeros((n, 36), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    mat.integrate_displacement_device(mesh, ud.data_ptr(), f.data_ptr(), c.data_ptr(), st)   # eager, u1
    torch.cuda.synchronize()
    f1 = f.clone()
    ud.copy_(to_device(u2))
    mat.integrate_displacement_device(mesh, ud.data_ptr(), f.data_ptr(), c.data_ptr(), st)   # eager, u2
    torch.cuda.synchronize()
    f2, c2 = f.clone(), c.clone()
    assert not torch.equal(f1, f2)
    graph = torch.cuda.CUDAGraph()
    f.zero_()
    with torch.cuda.graph(graph):
        mat.integrate_displacement_device(mesh, ud.data_ptr(), f.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert float(f.abs().max()) == 0.0          # nothing ran during capture
    ud.copy_(to_device(u1))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(f, f1)
    ud.copy_(to_device(u2))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(f, f2) and torch.equal(c, c2)


def test_dolfinx_adapters_run_against_a_stand_in_function_space(monkeypatch):
    """No dolfinx on either box: `SimplexMesh.from_dolfinx` / `Tet4Mesh.from_dolfinx` are exercised with a stand-in that
    exposes the attributes they read (mesh.geometry.x / .dofmap, dofmap.list / index_map, element.basix_element.tabulate,
    basix.make_quadrature) in the shapes dolfinx 0.8+ documents; the real thing is tests/test_dolfinx_integration.py."""
    import sys
    import types

    torch = pytest.importorskip("torch")
    from dolfinx_materials_amd.gradient import (SimplexMesh, Tet4Mesh, lagrange_simplex_table, p2_dofmap, simplex_quadrature)
    from helpers import mandel_strain, simplex_host_gradient

    hm, coords = make_mesh(2, distort=0.2, seed=9)
    cells = np.concatenate([hm.conn[:, list(k)] for k in KUHN], axis=0).astype(np.int32)
    dofmap, n_dofs, edges = p2_dofmap(cells)
    xd = np.concatenate([coords, 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])], axis=0)
    pts = simplex_quadrature(3, 2)

    basix = types.ModuleType("basix")
    basix.CellType = types.SimpleNamespace(triangle="triangle", tetrahedron="tetrahedron", hexahedron="hexahedron")
    basix.make_quadrature = lambda cell, degree: (pts, np.full(len(pts), 1.0 / 24))
    monkeypatch.setitem(sys.modules, "basix", basix)

    def space(degree):
        tab = lagrange_simplex_table(3, degree, pts)                         # (nqp, nd, 3)
        full = np.zeros((4, len(pts), tab.shape[1], 1))
        full[1:, :, :, 0] = tab.transpose(2, 0, 1)                            # basix: (derivative, point, dof, value)
        dm = dofmap if degree == 2 else cells
        nd = n_dofs if degree == 2 else len(coords)
        return types.SimpleNamespace(
            mesh=types.SimpleNamespace(topology=types.SimpleNamespace(dim=3), geometry=types.SimpleNamespace(x=coords, dofmap=cells)),
            dofmap=types.SimpleNamespace(index_map_bs=3, list=dm, index_map=types.SimpleNamespace(size_local=nd, num_ghosts=0),
                                         cell_dofs=lambda c: dm[c]),
            element=types.SimpleNamespace(basix_element=types.SimpleNamespace(tabulate=lambda n, p: full)),
            ufl_element=lambda: types.SimpleNamespace(degree=degree),
            tabulate_dof_coordinates=lambda: xd if degree == 2 else coords)

    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(3)
    mesh = SimplexMesh.from_dolfinx(space(2), 2)
    assert (mesh.nqp, mesh.nd, mesh.n_dofs, mesh.tdim) == (4, 10, n_dofs, 3)
    u = (xd * np.array([8e-3, -3e-3, -3e-3]) + 1e-3 * rng.standard_normal(xd.shape)).ravel()
    out = torch.empty((mesh.npoints, 6), dtype=torch.float64, device=dev)
    mesh.gradient_device(to_device(u).data_ptr(), 0, out.data_ptr(), st)
    torch.cuda.synchronize()
    H = simplex_host_gradient(coords, cells, dofmap, u, lagrange_simplex_table(3, 2, pts)).reshape(-1, 3, 3)
    assert np.abs(to_host(out) - mandel_strain(H)).max() < 1e-13
    t4 = Tet4Mesh.from_dolfinx(space(1), 2)
    assert t4.nqp == 4 and t4.n_cells == len(cells)
