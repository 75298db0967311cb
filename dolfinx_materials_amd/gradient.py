"""Device-side gradient evaluation for first-order hexahedra (``dxm_mesh_*`` of ``include/dxmat.h``).

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
    def from_dolfinx(cls, V, quadrature_degree, device=0):
        """From a dolfinx P1 vector function space on tetrahedra: ``u.x.array`` is the displacement vector to hand to
        ``integrate_displacement``; ``quadrature_degree`` as given to ``QuadratureMap`` (the gradient of a P1 field is
        constant per cell and is repeated at the cell's points, point = cell * nqp + q: ``quadrature_map.py:255-260``)."""
        import basix

        coords, conn = _dolfinx_p1_layout(V)
        if conn.shape[1] != 4:
            raise ValueError("Tet4Mesh.from_dolfinx needs a tetrahedral mesh")
        pts, _ = basix.make_quadrature(basix.CellType.tetrahedron, int(quadrature_degree))
        return cls(coords, conn, nqp=len(pts), device=device)


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
    def from_dolfinx(cls, V, quadrature_degree, device=0):
        """From a dolfinx P1 vector function space on hexahedra.  The Gauss points are basix's for
        ``quadrature_degree`` (what ``QuadratureMap`` uses: ``quadrature_map.py:239-243``, ``utils.py:89-94``), mapped
        from the reference cell [0,1]^3 to [-1,1]^3 in basix's own order, so Gauss point ``q`` of cell ``c`` is row
        ``c * nqp + q`` of the quadrature Functions; the cell's dofs come in the tensor-product vertex order of basix
        and are permuted to the corner order of the kernels (:data:`DOLFINX_HEX_TO_VTK`)."""
        import basix

        coords, conn = _dolfinx_p1_layout(V)
        if conn.shape[1] != 8:
            raise ValueError("Hex8Mesh.from_dolfinx needs a hexahedral mesh")
        pts, _ = basix.make_quadrature(basix.CellType.hexahedron, int(quadrature_degree))
        return cls(coords, conn[:, DOLFINX_HEX_TO_VTK], qpoints=2.0 * np.asarray(pts) - 1.0, device=device)
