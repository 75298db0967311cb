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
