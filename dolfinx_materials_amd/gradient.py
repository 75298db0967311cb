"""Device-side gradient evaluation for first-order hexahedra / tetrahedra and for Lagrange elements of any order on
straight-sided simplices (``dxm_mesh_*`` of ``include/dxmat.h``).

The step before the hot path in the reference is ``QuadratureExpression.eval`` ->
``fem.Expression.eval`` (``quadrature_function.py:45-51``): dolfinx tabulates the UFL gradient at
every Gauss point on the host.  ``Hex8Mesh`` keeps coordinates and connectivity on the GPU so that
the strain / deformation gradient array is produced directly in HBM from the displacement vector.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def gauss_points_hex(degree=2):
    """Tensor Gauss-Legendre points on [-1,1]^3, last coordinate fastest."""
    n = {0: 1, 1: 1, 2: 2, 3: 2, 4: 3, 5: 3}[degree]
    x, _ = np.polynomial.legendre.leggauss(n)
    return np.array([[a, b, c] for a in x for b in x for c in x])


class _DeviceMesh:
    def gradient_device(self, u_ptr, kind, grad_ptr, stream=0):
        """kind 0: Mandel strain (6); 1: F (9).  Device pointers, asynchronous on ``stream``."""
        _lib.check(self._lib.dxm_mesh_gradient_device(self._handle, int(u_ptr), int(kind), int(grad_ptr), int(stream) or None))

    @property
    def npoints(self):
        return self.n_cells * self.nqp

    @property
    def displacement_size(self):
        """Number of doubles of the displacement vector this mesh expects (``u.x.array.size``)."""
        return int(self._lib.dxm_mesh_displacement_size(self._handle))

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.dxm_mesh_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


#: our hex8 corner order (VTK / the order of ``dxm_mesh_create_hex8``) in terms of the tensor-product vertex order
#: of a dolfinx / basix first-order hexahedron ((0,0,0),(1,0,0),(0,1,0),(1,1,0),(0,0,1),(1,0,1),(0,1,1),(1,1,1))
DOLFINX_HEX_TO_VTK = (0, 1, 3, 2, 4, 5, 7, 6)


def _dolfinx_p1_layout(V):
    """Node coordinates and cell -> node table of a first-order vector Lagrange space ``V`` in the numbering of
    ``u.x.array.reshape(-1, 3)`` (owned + ghost dofs; all local cells, ghosts included, which is the cell set
    ``QuadratureMap`` integrates: ``quadrature_map.py:66-70``)."""
    element = V.ufl_element()
    degree = getattr(element, "degree", None)
    degree = degree() if callable(degree) else degree
    if degree not in (1, None) or V.dofmap.index_map_bs != 3:
        raise ValueError("device gradient evaluation needs a first-order vector Lagrange space with block size 3")
    coords = np.ascontiguousarray(V.tabulate_dof_coordinates()[:, :3], dtype=np.float64)
    conn = np.asarray(V.dofmap.list, dtype=np.int32)
    if conn.ndim == 1:   # older dolfinx: flat adjacency list
        conn = conn.reshape(-1, len(V.dofmap.cell_dofs(0)))
    return coords, conn


class Tet4Mesh(_DeviceMesh):
    """Linear tetrahedra: coords ``(n_nodes, 3)``, conn ``(n_cells, 4)``; the (constant) cell
    gradient is repeated at the cell's ``nqp`` Gauss points."""

    def __init__(self, coords, conn, nqp=1, device=0):
        self._lib = _lib.load()
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        self.n_nodes, self.n_cells, self.nqp = coords.shape[0], conn.shape[0], int(nqp)
        self.device = int(device)
        h = self._lib.dxm_mesh_create_tet4(coords.ctypes.data, self.n_nodes, conn.ctypes.data, self.n_cells, self.nqp, self.device)
        if not h:
            raise _lib.DxmError(f"dxm_mesh_create_tet4 failed: {_lib.last_error()}")
        self._handle = h

    @classmethod
    def from_dolfinx(cls, V, quadrature_degree, device=0, cells=None):
        """From a dolfinx P1 vector function space on tetrahedra: ``u.x.array`` is the displacement vector to hand to
        ``integrate_displacement``; ``quadrature_degree`` as given to ``QuadratureMap`` (the gradient of a P1 field is
        constant per cell and is repeated at the cell's points, point = cell * nqp + q: ``quadrature_map.py:255-260``).
        ``cells``: the cells of ONE ``QuadratureMap`` (``qmap.cells``, a multi-material problem) instead of all."""
        import basix

        coords, conn = _dolfinx_p1_layout(V)
        if conn.shape[1] != 4:
            raise ValueError("Tet4Mesh.from_dolfinx needs a tetrahedral mesh")
        pts, _ = basix.make_quadrature(basix.CellType.tetrahedron, int(quadrature_degree))
        return cls(coords, conn if cells is None else conn[np.asarray(cells)], nqp=len(pts), device=device)


class Hex8Mesh(_DeviceMesh):
    """coords ``(n_nodes, 3)``; conn ``(n_cells, 8)`` with the corner order
    ``(-,-,-)(+,-,-)(+,+,-)(-,+,-)(-,-,+)(+,-,+)(+,+,+)(-,+,+)``; Gauss point ``q`` of cell ``c``
    is point ``c * nqp + q``."""

    def __init__(self, coords, conn, qpoints=None, device=0):
        self._lib = _lib.load()
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        qp = np.ascontiguousarray(gauss_points_hex(2) if qpoints is None else qpoints, dtype=np.float64)
        self.n_nodes, self.n_cells, self.nqp = coords.shape[0], conn.shape[0], qp.shape[0]
        self.device = int(device)
        h = self._lib.dxm_mesh_create_hex8(
            coords.ctypes.data, self.n_nodes, conn.ctypes.data, self.n_cells, qp.ctypes.data, self.nqp, self.device
        )
        if not h:
            raise _lib.DxmError(f"dxm_mesh_create_hex8 failed: {_lib.last_error()}")
        self._handle = h

    @classmethod
    def from_dolfinx(cls, V, quadrature_degree, device=0, cells=None):
        """From a dolfinx P1 vector function space on hexahedra (``cells``: those of one ``QuadratureMap`` instead of all).  The Gauss points are basix's for
        ``quadrature_degree`` (what ``QuadratureMap`` uses: ``quadrature_map.py:239-243``, ``utils.py:89-94``), mapped
        from the reference cell [0,1]^3 to [-1,1]^3 in basix's own order, so Gauss point ``q`` of cell ``c`` is row
        ``c * nqp + q`` of the quadrature Functions; the cell's dofs come in the tensor-product vertex order of basix
        and are permuted to the corner order of the kernels (:data:`DOLFINX_HEX_TO_VTK`)."""
        import basix

        coords, conn = _dolfinx_p1_layout(V)
        if conn.shape[1] != 8:
            raise ValueError("Hex8Mesh.from_dolfinx needs a hexahedral mesh")
        pts, _ = basix.make_quadrature(basix.CellType.hexahedron, int(quadrature_degree))
        conn = conn if cells is None else conn[np.asarray(cells)]
        return cls(coords, conn[:, DOLFINX_HEX_TO_VTK], qpoints=2.0 * np.asarray(pts) - 1.0, device=device)


#: local vertex pairs of the edges of the reference triangle / tetrahedron in basix's edge numbering: the second-order
#: Lagrange element has its dofs at the vertices, then at the midpoints of these edges in this order
BASIX_EDGES = {2: ((1, 2), (0, 2), (0, 1)), 3: ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))}


def simplex_quadrature(tdim, degree):
    """Reference points (nqp, tdim) of a quadrature rule of the given degree (0..2) on the reference simplex with
    vertices 0, e_1, ..., e_tdim.  For stand-alone use and tests; with dolfinx take basix's points
    (:meth:`SimplexMesh.from_dolfinx` does), whose order defines the row order of the quadrature Functions."""
    if degree <= 1:
        return np.full((1, tdim), 1.0 / (tdim + 1))
    if degree != 2:
        raise ValueError("simplex_quadrature: degree 0, 1 or 2")
    if tdim == 2:
        return np.array([[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]])
    a, b = (5 + 3 * 5**0.5) / 20, (5 - 5**0.5) / 20
    return np.array([[b, b, b], [a, b, b], [b, a, b], [b, b, a]])


def lagrange_simplex_table(tdim, degree, points):
    """Reference derivatives ``dphi[q, m, d] = d N_m / d xi_d`` of the Lagrange basis of the given degree (1 or 2) on the
    reference simplex at ``points`` (nqp, tdim), dofs in basix's order (vertices, then :data:`BASIX_EDGES` midpoints).
    With barycentric coordinates l_0 = 1 - sum(xi), l_k = xi_k: N_v = l_v (2 l_v - 1), N_(ij) = 4 l_i l_j."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, tdim)
    lam = np.concatenate([1.0 - pts.sum(axis=1, keepdims=True), pts], axis=1)          # (nqp, tdim+1)
    dlam = np.concatenate([-np.ones((1, tdim)), np.eye(tdim)], axis=0)                  # (tdim+1, tdim), constant
    if degree == 1:
        return np.ascontiguousarray(np.broadcast_to(dlam, (len(pts), tdim + 1, tdim)))
    if degree != 2:
        raise ValueError("lagrange_simplex_table: degree 1 or 2 (hand any other table to SimplexMesh directly)")
    cols = [(4.0 * lam[:, v, None] - 1.0) * dlam[v] for v in range(tdim + 1)]
    cols += [4.0 * (lam[:, i, None] * dlam[j] + lam[:, j, None] * dlam[i]) for i, j in BASIX_EDGES[tdim]]
    return np.ascontiguousarray(np.stack(cols, axis=1))


def p2_dofmap(cells):
    """Second-order Lagrange dofmap of a simplex mesh given by its vertex table ``cells`` (n_cells, tdim+1): vertex dofs
    keep the vertex numbers, one more dof per unique edge.  Returns ``(dofmap (n_cells, nd), n_dofs, edge_vertices)``
    with ``edge_vertices`` (n_edges, 2) the end points of the edge dof ``n_vertices + k``."""
    cells = np.asarray(cells, dtype=np.int64)
    tdim = cells.shape[1] - 1
    nv = int(cells.max()) + 1
    pairs = np.stack([np.sort(cells[:, list(e)], axis=1) for e in BASIX_EDGES[tdim]], axis=1)   # (n_cells, n_e, 2)
    uniq, inv = np.unique(pairs.reshape(-1, 2), axis=0, return_inverse=True)
    dofmap = np.concatenate([cells, nv + inv.reshape(len(cells), -1)], axis=1).astype(np.int32)
    return dofmap, nv + len(uniq), uniq


class SimplexMesh(_DeviceMesh):
    """Lagrange displacement of any order on straight-sided simplices (``dxm_mesh_create_simplex``): the P2 spaces of the
    reference's demos (tet10 with 4 Gauss points, ``finite_strain_elastoplasticity.py:115-117``; tri6 with 3, embedded as
    plane strain, ``plane_elastoplasticity.py:96-100``).

    ``coords`` (n_vertices, 2|3) and ``geom_conn`` (n_cells, tdim+1) give the affine geometry; ``dofmap`` (n_cells, nd)
    indexes the ``n_dofs`` blocks of ``tdim`` displacement components; ``dphi`` (nqp, nd, tdim) are the reference
    derivatives of the nd shape functions at the Gauss points (:func:`lagrange_simplex_table`, or basix's tabulation).
    Gauss point ``q`` of cell ``c`` is point ``c * nqp + q``."""

    def __init__(self, coords, geom_conn, dofmap, n_dofs, dphi, device=0):
        self._lib = _lib.load()
        geom_conn = np.ascontiguousarray(geom_conn, dtype=np.int32)
        self.tdim = geom_conn.shape[1] - 1
        coords = np.asarray(coords, dtype=np.float64)
        if coords.shape[1] == 2:
            coords = np.concatenate([coords, np.zeros((len(coords), 1))], axis=1)
        coords = np.ascontiguousarray(coords)
        dofmap = np.ascontiguousarray(dofmap, dtype=np.int32)
        dphi = np.ascontiguousarray(dphi, dtype=np.float64)
        if dphi.ndim != 3 or dphi.shape[1] != dofmap.shape[1] or dphi.shape[2] != self.tdim or len(dofmap) != len(geom_conn):
            raise ValueError("dphi must be (nqp, nd, tdim) with nd = dofmap.shape[1]; one dofmap row per cell")
        self.n_nodes, self.n_cells, self.nqp = coords.shape[0], geom_conn.shape[0], dphi.shape[0]
        self.nd, self.n_dofs, self.device = dofmap.shape[1], int(n_dofs), int(device)
        h = self._lib.dxm_mesh_create_simplex(self.tdim, coords.ctypes.data, self.n_nodes, geom_conn.ctypes.data, self.n_cells,
                                              dofmap.ctypes.data, self.nd, self.n_dofs, dphi.ctypes.data, self.nqp, self.device)
        if not h:
            raise _lib.DxmError(f"dxm_mesh_create_simplex failed: {_lib.last_error()}")
        self._handle = h

    @classmethod
    def lagrange(cls, coords, cells, degree=2, quadrature_degree=None, device=0):
        """Stand-alone constructor (no dolfinx): P1 or P2 displacement on the vertex mesh ``(coords, cells)``; P2 dofs as
        numbered by :func:`p2_dofmap`.  Returns ``(mesh, dof_coords)`` with the (n_dofs, coords.shape[1]) dof positions."""
        coords, cells = np.asarray(coords, dtype=np.float64), np.asarray(cells)
        tdim = cells.shape[1] - 1
        qdeg = 2 * (degree - 1) if quadrature_degree is None else quadrature_degree
        dphi = lagrange_simplex_table(tdim, degree, simplex_quadrature(tdim, qdeg))
        if degree == 1:
            return cls(coords, cells, cells, len(coords), dphi, device=device), coords.copy()
        dofmap, n_dofs, edges = p2_dofmap(cells)
        dof_coords = np.concatenate([coords, 0.5 * (coords[edges[:, 0]] + coords[edges[:, 1]])], axis=0)
        return cls(coords, cells, dofmap, n_dofs, dphi, device=device), dof_coords

    @classmethod
    def from_dolfinx(cls, V, quadrature_degree, device=0, cells=None):
        """From a dolfinx vector Lagrange space (any order, block size = topological dimension; ``cells``: those of one
        ``QuadratureMap`` instead of all) on a first-order
        (straight-sided) triangle or tetrahedron mesh: geometry from ``mesh.geometry``, the dofmap of ``V`` as it is,
        Gauss points and tabulated derivatives from basix -- so point ``c * nqp + q`` is row ``c * nqp + q`` of the
        quadrature Functions (``quadrature_map.py:239-260``) and ``u.x.array`` is the vector to hand to
        ``integrate_displacement``."""
        import basix

        mesh = V.mesh
        tdim = mesh.topology.dim
        if tdim not in (2, 3) or V.dofmap.index_map_bs != tdim:
            raise ValueError("SimplexMesh.from_dolfinx needs a vector Lagrange space with block size = topological dimension")
        geom_conn = np.asarray(mesh.geometry.dofmap, dtype=np.int32)
        if geom_conn.ndim == 1:
            geom_conn = geom_conn.reshape(-1, tdim + 1)
        if geom_conn.shape[1] != tdim + 1:
            raise ValueError("SimplexMesh.from_dolfinx needs a first-order (affine) simplex geometry")
        dofmap = np.asarray(V.dofmap.list, dtype=np.int32)
        if dofmap.ndim == 1:
            dofmap = dofmap.reshape(len(geom_conn), -1)
        n_dofs = V.dofmap.index_map.size_local + V.dofmap.index_map.num_ghosts
        cell = basix.CellType.triangle if tdim == 2 else basix.CellType.tetrahedron
        pts, _ = basix.make_quadrature(cell, int(quadrature_degree))
        tab = np.asarray(V.element.basix_element.tabulate(1, np.asarray(pts)))
        if tab.ndim == 4:
            tab = tab[..., 0]
        if tab.shape[2] != dofmap.shape[1]:   # blocked element tabulated with its block: scalar basis is every bs-th
            tab = tab[:, :, ::tdim]
        dphi = np.ascontiguousarray(tab[1:1 + tdim].transpose(1, 2, 0))
        if cells is not None:
            geom_conn, dofmap = geom_conn[np.asarray(cells)], dofmap[np.asarray(cells)]
        return cls(np.asarray(mesh.geometry.x), geom_conn, dofmap, n_dofs, dphi, device=device)
