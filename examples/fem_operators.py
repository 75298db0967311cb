"""Matrix-free assembly-side operators on a hex8 mesh, on the device -- ``examples/libdxmfem.so`` (``examples/csrc/dxmfem.hip``).

What dolfinx assembly does with the quadrature Functions ``QuadratureMap.update`` filled -- ``dot(sig, strain(v)) * dx`` and its
derivative with the tangent blocks (reference ``tests/uniaxial_tension.py:62-67``, ``quadrature_map.py:132-158``) -- for a caller
that keeps displacement, stress and tangent on the GPU (``examples/device_fem.py``):

    internal force   f = sum_q w detJ B_q^T sigma_q               stress (npoints, 6) Mandel -> f (n_nodes * 3)
    tangent apply    y = sum_q w detJ B_q^T Ct_q B_q x            tangent "full" (npoints, 36) or "coef" (npoints, 9)
    tangent diagonal d = diag(sum_q w detJ B_q^T Ct_q B_q)        coefficient layout only

NOT product code: FEM assembly stays on the host in the scope of ``dolfinx_materials_amd``, and
``libdxmat.so`` exports the constitutive update only.  The library here is stateless; this class owns every array (torch tensors).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdxmfem.so")
_lib = None


def load():
    """``examples/libdxmfem.so`` (built by ``make -C examples``; cross-compiles without a GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.run(["make", "-C", HERE], check=True)
        lib = C.CDLL(LIB_PATH)
        lib.dxf_last_error.restype = C.c_char_p
        lib.dxf_hex8_operator.restype = C.c_int
        lib.dxf_hex8_operator.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


def _to_device(a, dev):
    """numpy -> device through a page-locked staging tensor (pageable memory is never handed to the runtime: DESIGN.md section 1)."""
    t = torch.from_numpy(np.ascontiguousarray(a))
    pin = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pin.copy_(t)
    out = pin.to(dev)
    torch.cuda.synchronize(dev)
    return out


class Hex8Operators:
    """coords ``(n_nodes, 3)``, conn ``(n_cells, 8)`` in the corner order of ``dxm_mesh_create_hex8``; 8 Gauss points per cell
    (``qpoints`` in [-1, 1]^3, default the 2x2x2 Gauss-Legendre points of ``gradient.gauss_points_hex(2)``, unit weights)."""

    OPS = {"force": 0, "apply": 1, "diagonal": 2}

    def __init__(self, coords, conn, qpoints=None, weights=None, device=0):
        from dolfinx_materials_amd.gradient import gauss_points_hex

        self._lib = load()
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        if coords.ndim != 2 or coords.shape[1] != 3 or conn.ndim != 2 or conn.shape[1] != 8:
            raise ValueError("coords (n_nodes, 3) and conn (n_cells, 8) expected")
        self.n_nodes, self.n_cells = coords.shape[0], conn.shape[0]
        if conn.min() < 0 or conn.max() >= self.n_nodes:
            raise ValueError("connectivity entry out of range")
        self.device = int(device)
        dev = torch.device("cuda", self.device)
        self.xi = np.ascontiguousarray(gauss_points_hex(2) if qpoints is None else qpoints, dtype=np.float64)
        if self.xi.shape != (8, 3):
            raise ValueError("the device operators need 8 Gauss points per cell")
        self.w = np.ones(8)
        if weights is not None:
            self.set_weights(weights)
        # node -> (cell, corner) table: element values are stored corner-major, entry = corner * n_cells + cell, ascending per node
        # (a stable counting sort of the transposed connectivity: a fixed summation order, reproducible bit for bit)
        self._coords, self._conn = _to_device(coords, dev), _to_device(conn, dev)
        flat = self._conn.t().reshape(-1)
        self._adj = torch.argsort(flat, stable=True).to(torch.int32)
        self._ptr = torch.zeros(self.n_nodes + 1, dtype=torch.int64, device=dev)
        self._ptr[1:] = torch.cumsum(torch.bincount(flat, minlength=self.n_nodes), 0)
        del flat
        self._fe = torch.empty(8 * self.n_cells * 3, dtype=torch.float64, device=dev)

    def set_weights(self, weights):
        w = np.ascontiguousarray(weights, dtype=np.float64)
        if w.size != 8:
            raise ValueError("8 quadrature weights expected")
        self.w = w.reshape(8).copy()

    def _run(self, op, field_ptr, layout, x_ptr, y_ptr, stream):
        rc = self._lib.dxf_hex8_operator(self.OPS[op], self.device, self._coords.data_ptr(), self._conn.data_ptr(), self.n_cells, self.n_nodes,
                                         self.xi.ctypes.data, self.w.ctypes.data, int(field_ptr) or None, layout, int(x_ptr) or None,
                                         self._fe.data_ptr(), self._ptr.data_ptr(), self._adj.data_ptr(), int(y_ptr) or None, int(stream) or None)
        if rc != 0:
            raise RuntimeError(f"dxf_hex8_operator({op}) failed ({rc}): {self._lib.dxf_last_error().decode()}")

    def internal_force_device(self, flux_ptr, f_ptr, stream=0):
        """``f = sum_q w detJ B_q^T sigma_q``: stress ``(npoints, 6)`` -> nodal vector ``(n_nodes * 3)``; device pointers."""
        self._run("force", flux_ptr, 0, 0, f_ptr, stream)

    def tangent_apply_device(self, ct_ptr, x_ptr, y_ptr, layout="coef", stream=0):
        """``y = sum_q w detJ B_q^T Ct_q B_q x`` with the tangent in the ``"coef"`` (N,9) or ``"full"`` (N,36) layout."""
        self._run("apply", ct_ptr, {"full": 0, "coef": 2}[layout], x_ptr, y_ptr, stream)

    def tangent_diagonal_device(self, coef_ptr, d_ptr, stream=0):
        """Diagonal of the operator of :meth:`tangent_apply_device` (coefficient layout)."""
        self._run("diagonal", coef_ptr, 2, 0, d_ptr, stream)
