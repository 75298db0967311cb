"""``QuadratureMap`` without dolfinx: the accelerated update of ``quadrature_map.AcceleratedUpdate`` over a stand-in base.

``quadrature_map.py`` overrides ``update / advance / initialize_state`` of the reference's ``QuadratureMap`` in terms of
that class's attribute surface.  dolfinx exists neither in the build container nor on the GPU box, so the same mixin is
put over :class:`FieldMapBase`, a dolfinx-free stand-in that offers exactly that surface -- ``material, mesh, cells, dofs,
gradients, fluxes, internal_state_variables, external_state_variables, jacobian_flatten, get_gradient_vals,
set_data_manager, update_initial_state, _initialized`` (reference ``quadrature_map.py:51-130, :197-279``) -- with plain
numpy arrays where dolfinx has quadrature ``Function`` objects:

* :class:`Field` stands for a quadrature ``fem.Function``: ``.x.array`` is the flat memory (point = cell * nqp + q,
  component fastest, ``utils.py:98-104``), ``.values`` its ``(points, dim)`` view;
* :class:`ExpressionField` stands for ``QuadratureExpression`` (``quadrature_function.py:23-51``): ``.expression.eval(mesh,
  cells, values=None)`` like ``dolfinx.fem.Expression.eval``, ``.eval(cells)`` scatters into ``.function``.

``QuadratureFieldMap(num_cells, nqp, material, cells=None)`` is what the FE loop of ``examples/``, the bench's
``host_path`` figures and the tests use; everything it does per update is the mixin's code, i.e. the code a dolfinx user
gets from ``dolfinx_materials_amd.quadrature_map.QuadratureMap``.
"""
from __future__ import annotations

import numpy as np

from .quadrature_map import AcceleratedUpdate


class _Flat:
    """``field.x``: what carries ``.array`` on a dolfinx ``Function``."""

    def __init__(self, array):
        self.array = array


class Field:
    """Per-Gauss-point field of one quantity over all cells: ``x.array`` flat, ``values`` its ``(points, dim)`` view."""

    def __init__(self, name, dim, points):
        self.name, self.dim = name, max(1, int(dim))
        self.x = _Flat(np.zeros(points * self.dim))

    @property
    def values(self):
        return self.x.array.reshape(-1, self.dim)


class _Evaluator:
    """Stands for the compiled ``dolfinx.fem.Expression``: ``eval(mesh, cells, values=None)`` returns / fills a
    ``(len(cells), nqp * dim)`` array, cell-major."""

    def __init__(self, func):
        self._func = func

    def eval(self, mesh, cells, values=None):
        rows = np.asarray(self._func(cells), dtype=np.float64).reshape(len(cells), -1)
        if values is None:
            return rows
        values[...] = rows
        return values


class ExpressionField:
    """Stands for ``QuadratureExpression``: an expression plus the Function its values are kept in."""

    def __init__(self, name, dim, evaluator, num_cells, nqp):
        self.name = name
        self.expression = _Evaluator(evaluator)
        self.function = Field(name, dim, num_cells * nqp)
        self._num_cells = num_cells

    def eval(self, cells):
        self.function.x.array.reshape(self._num_cells, -1)[cells] = self.expression.eval(None, cells)


class FieldMapBase:
    """The attribute surface of the reference's ``QuadratureMap`` for ``num_cells`` cells of ``nqp`` points each;
    ``cells`` (default: all) are the cells this map's material acts on."""

    def __init__(self, num_cells, nqp, material, cells=None):
        self.mesh = None
        self.material = material
        self.nqp, self.num_cells_total = int(nqp), int(num_cells)
        self.cells = np.arange(num_cells, dtype=np.int32) if cells is None else np.asarray(cells, dtype=np.int32)
        self.num_cells = len(self.cells)
        total = self.num_cells_total * self.nqp
        # quadrature_map.py:83-87; narrower for the engine's packed tangent layouts ("sym" / "coef")
        self.jacobian_width = getattr(material, "tangent_size", None) or sum(int(np.prod(s)) for s in material.tangent_blocks.values())
        self.jacobian_flatten = Field("jacobian", self.jacobian_width, total)
        self.fluxes = {name: Field(name, dim, total) for name, dim in material.fluxes.items()}
        self.internal_state_variables = {name: Field(name, dim, total) for name, dim in material.internal_state_variables.items()}
        self.gradients, self.external_state_variables = {}, {}
        self.rotation_func = None
        self.set_data_manager(self.cells)
        self._initialized = False

    def set_data_manager(self, cells):
        # rows of the fields this map owns, in the order the material sees them (quadrature_map.py:231-233, :259-260)
        self.dofs = (np.asarray(cells, dtype=np.int64)[:, None] * self.nqp + np.arange(self.nqp)).reshape(-1)
        self.material.set_data_manager(len(self.dofs))

    @property
    def variables(self):
        return {**self.gradients, **self.fluxes, **self.internal_state_variables}

    def register_gradient(self, name, evaluator):
        """``evaluator(cells) -> (len(cells) * nqp, dim)`` stands for the compiled UFL expression."""
        if name not in self.material.gradients:
            raise ValueError(f"Gradient '{name}' is not available from the material law.")
        self.gradients[name] = ExpressionField(name, self.material.gradients[name], evaluator, self.num_cells_total, self.nqp)

    def get_gradient_vals(self, gradient, cells):
        gradient.eval(cells)
        return gradient.function.values[self.dofs, :]

    def update_initial_state(self, field_name, value=None):
        """Initial value of a flux or an internal state variable: a number / one row broadcast over the map's points,
        or the field's current content (``quadrature_map.py:262-279``)."""
        field = self.fluxes.get(field_name) or self.internal_state_variables.get(field_name)
        if field is None:
            raise ValueError("Can only initialize a flux or internal state variables.")
        if value is not None:
            field.values[self.dofs] = np.broadcast_to(np.asarray(value, dtype=np.float64), (len(self.dofs), field.dim))
        self.material.set_initial_state_dict({field_name: field.values[self.dofs]})


class QuadratureFieldMap(AcceleratedUpdate, FieldMapBase):
    """``AcceleratedUpdate`` (the ``update / advance / initialize_state`` a dolfinx user gets) over the stand-in base."""

    def _jacobian_width(self):
        return self.jacobian_width

    # names used by the examples and tests
    @property
    def points(self):
        return self.dofs

    @property
    def covers_everything(self):
        return self._accel_plan().identity

    @property
    def _bound(self):
        return self._accel_plan().bound
