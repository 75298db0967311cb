"""Host-side exchange between quadrature fields and the engine.

What ``QuadratureMap`` does around the hot call (reference ``quadrature_map.py:231-360``, ``utils.py:98-143``) --
hand the gradients of a set of cells to ``material.integrate``, put flux / tangent / internal state variables
back into per-Gauss-point fields, roll the state at the end of an increment -- organised around what this
engine can do, for callers that do not have dolfinx (the FE loop of ``examples/``, the tests) and as the model
of how a dolfinx ``QuadratureMap`` is wired to it (``INTEGRATION.md``):

* a field is ONE ``(points, dim)`` array (``.values``; ``.x.array`` is its flat view, the memory layout of a
  dolfinx quadrature Function: point = cell * nqp + q, component fastest);
* a map over all cells binds the flux and tangent fields as the material's output arrays
  (``HIPMaterial.bind_outputs`` page-locks them in place): ``integrate`` delivers into them, nothing is
  scattered; a map over a subset of cells scatters rows through a point index built once;
* internal state variables cross PCIe when an increment is accepted (``advance``), not in every Newton
  iteration; NaNs are reported by the kernel's own status record instead of three full-array passes
  (``quadrature_map.py:322-324``);
* the gradient comes from a host evaluator (``register_gradient``) or from the device
  (``register_device_gradient``: only the displacement vector is uploaded).

The method names ``update / advance / initialize_state / update_initial_state / register_gradient`` and the
attributes ``fluxes / internal_state_variables / gradients / jacobian_flatten`` are the reference's public
surface for this path (``quadrature_map.py:107-117, :197, :262, :281, :297, :350``).
"""
from __future__ import annotations

import numpy as np


class _Flat:
    """``field.x.array``: the flat view a dolfinx ``Function`` offers."""

    def __init__(self, owner):
        self._owner = owner

    @property
    def array(self):
        return self._owner.values.reshape(-1)


class Field:
    """Per-Gauss-point field of one quantity: ``values`` is ``(points, dim)``, C-contiguous, fp64."""

    def __init__(self, name, dim, points):
        self.name, self.dim = name, max(1, int(dim))
        self.values = np.zeros((points, self.dim))
        self.x = _Flat(self)


class QuadratureFieldMap:
    """Fields of one material region: ``cells`` (default: all ``num_cells``) with ``nqp`` points each."""

    def __init__(self, num_cells, nqp, material, cells=None):
        self.material = material
        self.nqp, self.num_cells_total = int(nqp), int(num_cells)
        self.cells = np.arange(num_cells, dtype=np.int32) if cells is None else np.asarray(cells, dtype=np.int32)
        total = self.num_cells_total * self.nqp
        # rows of the fields this map owns, in the order the material sees them
        self.points = (self.cells.astype(np.int64)[:, None] * self.nqp + np.arange(self.nqp)).reshape(-1)
        self.covers_everything = len(self.points) == total and np.array_equal(self.points, np.arange(total))
        material.set_data_manager(len(self.points))
        direct = self.covers_everything and hasattr(material, "bind_outputs")
        # quadrature_map.py:83-87; narrower for the engine's packed tangent layouts
        width = getattr(material, "tangent_size", None) or sum(int(np.prod(shape)) for shape in material.tangent_blocks.values())
        self.jacobian_flatten = Field("jacobian", width, total)
        self.fluxes = {name: Field(name, dim, total) for name, dim in material.fluxes.items()}
        self.internal_state_variables = {name: Field(name, dim, total) for name, dim in material.internal_state_variables.items()}
        self.gradients, self._evaluators, self._on_device = {}, {}, None
        self._bound = False
        if direct and len(self.fluxes) == 1:
            (flux_field,) = self.fluxes.values()
            material.bind_outputs(flux=flux_field.x.array, tangent=self.jacobian_flatten.x.array)
            self._bound = True
        self._initialized = False

    # kept for callers written against the reference (quadrature_map.py:231-233)
    @property
    def dofs(self):
        return self.points

    @property
    def variables(self):
        return {**self.gradients, **self.fluxes, **self.internal_state_variables}

    # ---- rows in / rows out ----------------------------------------------------------------------
    def _rows(self, field):
        return field.values if self.covers_everything else field.values[self.points]

    def _store(self, field, rows):
        rows = np.asarray(rows).reshape(len(self.points), field.dim)
        if self.covers_everything:
            if not np.shares_memory(rows, field.values):   # bound outputs are already in place
                field.values[...] = rows
        else:
            field.values[self.points] = rows

    def _store_columns(self, fields, block):
        col = 0
        for name, f in fields.items():
            self._store(f, block[:, col:col + f.dim])
            col += f.dim

    # ---- gradients ---------------------------------------------------------------------------------
    def register_gradient(self, name, evaluator):
        """``evaluator(cells) -> (len(cells) * nqp, dim)`` stands for the compiled expression of
        ``QuadratureExpression.eval`` (``quadrature_function.py:45-51``)."""
        if name not in self.material.gradients:
            raise ValueError(f"Gradient '{name}' is not available from the material law.")
        self.gradients[name] = Field(name, self.material.gradients[name], self.num_cells_total * self.nqp)
        self._evaluators[name] = evaluator

    def register_device_gradient(self, mesh, displacement):
        """Evaluate the gradient on the GPU (``gradient.Hex8Mesh`` / ``Tet4Mesh``) from ``displacement()``: only the
        nodal vector is uploaded per update.  Needs a map over all cells in mesh order."""
        if not self.covers_everything or mesh.npoints != len(self.points):
            raise ValueError("device gradient evaluation needs a map over all cells of the mesh")
        self._on_device = (mesh, displacement)

    def _evaluate_gradients(self):
        blocks = []
        for name in self.material.gradients:
            f = self.gradients[name]
            rows = np.asarray(self._evaluators[name](self.cells), dtype=np.float64).reshape(len(self.points), f.dim)
            self._store(f, rows)
            blocks.append(rows)
        return blocks[0] if len(blocks) == 1 else np.concatenate(blocks, axis=1)

    # ---- state life cycle ----------------------------------------------------------------------------
    def update_initial_state(self, field_name, value=None):
        """Initial value of a flux or an internal state variable: a number / one row broadcast over the points,
        or the field's current content (``quadrature_map.py:262-279``)."""
        field = self.fluxes.get(field_name) or self.internal_state_variables.get(field_name)
        if field is None:
            raise ValueError("Can only initialize a flux or internal state variables.")
        if value is not None:
            self._store(field, np.broadcast_to(np.asarray(value, dtype=np.float64), (len(self.points), field.dim)))
        self.material.set_initial_state_dict({field_name: np.array(self._rows(field))})

    def initialize_state(self):
        """s0 from the current content of every field and the gradients at the current configuration."""
        state = {name: np.array(self._rows(f)) for name, f in {**self.fluxes, **self.internal_state_variables}.items()}
        if self._evaluators:
            self._evaluate_gradients()
            state.update({name: np.array(self._rows(self.gradients[name])) for name in self.gradients})
        self.material.set_initial_state_dict(state)
        self._initialized = True

    def update(self):
        """One constitutive update of the region (called once per global Newton iteration, ``solvers.py:173-176``)."""
        if not self._initialized:
            self.initialize_state()
        if self._on_device is not None:
            mesh, displacement = self._on_device
            flux, isv, tangent = self.material.integrate_displacement(mesh, displacement())
        else:
            flux, isv, tangent = self.material.integrate(self._evaluate_gradients())
        status = getattr(self.material, "last_stats", None)
        if status is not None:
            assert status["n_nan"] == 0, "non-finite constitutive update"
        else:
            assert not (np.isnan(flux).any() or np.isnan(np.asarray(tangent)).any())
        self._store_columns(self.fluxes, np.asarray(flux))
        self._store(self.jacobian_flatten, tangent)
        # the internal state variables are written back by advance(); callers that want them per iteration read
        # `isv` (it downloads on first access) or call refresh_internal_state_variables()
        self._last_isv = isv

    def refresh_internal_state_variables(self):
        self._store_columns(self.internal_state_variables, np.asarray(self._last_isv))

    def advance(self):
        """Accept the increment: s0 <- s1 on the device, final state into the fields (``quadrature_map.py:350-360``)."""
        self.material.data_manager.update()
        final = self.material.get_final_state_dict()
        for name, f in {**self.fluxes, **self.internal_state_variables}.items():
            self._store(f, final[name])
