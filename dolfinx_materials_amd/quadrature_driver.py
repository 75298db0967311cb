"""dolfinx-free replay of the ``QuadratureMap`` data flow (reference ``quadrature_map.py:51-360``).

dolfinx is not installed on the build container nor on the GPU box, so the real ``QuadratureMap``
cannot run there.  This module reproduces, on plain numpy arrays, exactly the part of it that
surrounds the hot path -- the call order and the ``(N, dim)`` gather / scatter contract:

* quadrature "Functions" are flat arrays of ``num_cells * nqp * dim`` doubles whose ``(-1, dim)``
  view is indexed by ``dof = cell * nqp + q`` (``utils.py:98-104``, ``quadrature_map.py:255-260``);
* ``update()``: evaluate the registered gradients on ``cells`` -> ``material.integrate`` -> scatter
  flux / ISV / flattened tangent (``quadrature_map.py:297-348``, ``utils.py:136-143``);
* ``advance()``: ``data_manager.update()`` then write the final state back (``:350-360``);
* ``initialize_state`` / ``update_initial_state`` (``:262-295``).

It is host logic only (no arithmetic of the constitutive laws) and is what the tests and the
stand-in FE loop of ``examples/`` drive the engine with.
"""
from __future__ import annotations

import numpy as np


class _X:
    def __init__(self, n):
        self.array = np.zeros(n)


class ArrayFunction:
    """Minimal stand-in of a quadrature ``fem.Function``: ``.x.array`` flat storage + ``dim``."""

    def __init__(self, name, dim, num_points):
        self.name = name
        self.dim = max(1, int(dim))
        self.x = _X(num_points * self.dim)


def _get_vals(fun: ArrayFunction):
    """``utils.py:98-104``."""
    return fun.x.array.reshape((-1, fun.dim))


def _update_vals(fun: ArrayFunction, array, cells=None):
    """``utils.py:136-143`` (block scatter by cell).  A map over all cells in order is the identity
    permutation; it is written with one contiguous copy instead of the reference's fancy-index
    scatter (same result, and the only host-side cost that scales with the batch here)."""
    arr = np.asarray(array).ravel()
    if cells is None or (len(arr) == fun.x.array.size and len(cells) > 0 and cells[0] == 0
                         and cells[-1] == len(cells) - 1 and np.all(np.diff(cells) == 1)):
        fun.x.array[:] = arr
    else:
        bs = len(arr) // len(cells)
        dofs = np.add.outer(np.asarray(cells) * bs, np.arange(bs)).ravel()
        fun.x.array[dofs] = arr


class QuadratureFieldMap:
    """Same members and call order as ``QuadratureMap`` for the constitutive-update path."""

    def __init__(self, num_cells, nqp, material, cells=None):
        self.num_cells_total = int(num_cells)
        self.nqp = int(nqp)
        self.material = material
        self.cells = np.arange(num_cells, dtype=np.int32) if cells is None else np.asarray(cells, dtype=np.int32)
        npts = self.num_cells_total * self.nqp
        buff = sum(int(np.prod(shape)) for shape in material.tangent_blocks.values())
        self.jacobian_flatten = ArrayFunction("jacobian", buff, npts)  # quadrature_map.py:83-87
        self.fluxes = {n: ArrayFunction(n, d, npts) for n, d in material.fluxes.items()}
        self.internal_state_variables = {
            n: ArrayFunction(n, d, npts) for n, d in material.internal_state_variables.items()
        }
        self.gradients = {}
        self._grad_eval = {}
        self._device_gradient = None
        self.set_data_manager(self.cells)
        self._initialized = False

    # ---- quadrature_map.py:231-260 ----------------------------------------------------------
    def set_data_manager(self, cells):
        self.dofs = self._cell_to_dofs(cells)
        self.material.set_data_manager(len(self.dofs))

    def _cell_to_dofs(self, cells):
        q = self.nqp
        return (np.repeat(q * cells[:, np.newaxis], q, axis=1) + np.repeat(np.arange(q)[np.newaxis, :], len(cells), axis=0)).ravel()

    @property
    def variables(self):
        return {**self.gradients, **self.fluxes, **self.internal_state_variables}

    # ---- quadrature_map.py:197-220 ----------------------------------------------------------
    def register_gradient(self, name, evaluator):
        """``evaluator(cells) -> (len(cells) * nqp, dim)``: stands for the compiled
        ``fem.Expression`` of ``QuadratureExpression.eval`` (``quadrature_function.py:45-51``)."""
        if name not in self.material.gradients:
            raise ValueError(f"Gradient '{name}' is not available from the material law.")
        self.gradients[name] = ArrayFunction(name, self.material.gradients[name], self.num_cells_total * self.nqp)
        self._grad_eval[name] = evaluator

    def register_device_gradient(self, mesh, displacement):
        """Evaluate the gradient on the GPU (``gradient.Hex8Mesh``) from ``displacement()`` instead
        of on the host: only the displacement vector is uploaded per update.  Needs a map over all
        cells in mesh order (point = cell * nqp + q)."""
        if len(self.cells) != self.num_cells_total or mesh.npoints != len(self.dofs):
            raise ValueError("device gradient evaluation needs a map over all cells of the mesh")
        self._device_gradient = (mesh, displacement)

    def get_gradient_vals(self, name, cells):
        fun = self.gradients[name]
        vals = np.asarray(self._grad_eval[name](cells), dtype=np.float64).reshape(len(cells) * self.nqp, fun.dim)
        _update_vals(fun, vals, cells)
        return _get_vals(fun)[self.dofs, :]

    # ---- quadrature_map.py:262-295 ----------------------------------------------------------
    def update_initial_state(self, field_name, value=None):
        if field_name in self.fluxes:
            field = self.fluxes[field_name]
        elif field_name in self.internal_state_variables:
            field = self.internal_state_variables[field_name]
        else:
            raise ValueError("Can only initialize a flux or internal state variables.")
        values = _get_vals(field)[self.dofs]
        if value is not None:
            values = np.broadcast_to(np.asarray(value, dtype=np.float64), values.shape).copy()
            _update_vals(field, values, self.cells)
        self.material.set_initial_state_dict({field_name: values})

    def initialize_state(self):
        state_flux = {k: _get_vals(f)[self.dofs] for k, f in self.fluxes.items()}
        state_isv = {k: _get_vals(f)[self.dofs] for k, f in self.internal_state_variables.items()}
        state_grad = {k: self.get_gradient_vals(k, self.cells) for k in self.gradients}
        self.material.set_initial_state_dict({**state_grad, **state_flux, **state_isv})
        self._initialized = True

    # ---- quadrature_map.py:297-348 ----------------------------------------------------------
    def update(self):
        if not self._initialized:
            self.initialize_state()
        if self._device_gradient is not None:
            mesh, displacement = self._device_gradient
            flux_vals, isv_vals, Ct_vals = self.material.integrate_displacement(mesh, displacement())
        else:
            grad_vals = np.concatenate(
                [self.get_gradient_vals(name, self.cells) for name in self.material.gradients.keys()], axis=1
            )
            flux_vals, isv_vals, Ct_vals = self.material.integrate(grad_vals)
        # the reference makes three full np.isnan passes here (quadrature_map.py:322-324); the
        # engine reports the same condition from the device
        stats = getattr(self.material, "last_stats", None)
        if stats is not None:
            assert stats["n_nan"] == 0
        else:
            assert not np.any(np.isnan(flux_vals)) and not np.any(np.isnan(Ct_vals))
        self.update_fluxes(flux_vals)
        self.update_internal_state_variables(isv_vals)
        _update_vals(self.jacobian_flatten, Ct_vals, self.cells)

    def update_fluxes(self, flux_vals):
        buff = 0
        for name, dim in self.material.fluxes.items():
            _update_vals(self.fluxes[name], flux_vals[:, buff : buff + dim], self.cells)
            buff += dim

    def update_internal_state_variables(self, isv_vals):
        buff = 0
        for name, dim in self.material.internal_state_variables.items():
            _update_vals(self.internal_state_variables[name], isv_vals[:, buff : buff + dim], self.cells)
            buff += dim

    # ---- quadrature_map.py:350-360 ----------------------------------------------------------
    def advance(self):
        self.material.data_manager.update()
        final_state = self.material.get_final_state_dict()
        for key in self.variables.keys():
            if key not in self.gradients:
                _update_vals(self.variables[key], final_state[key], self.cells)
