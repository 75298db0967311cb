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

from .quadrature_map import AcceleratedUpdate, tangent_entries


class Column:
    """Stands for a scalar UFL expression over the quadrature points: component ``i`` of a :class:`Field` and what ``+ - * /``
    make of such components and numbers.  ``evaluate()`` returns the ``(points,)`` values from the fields' CURRENT content --
    like a UFL expression, it is built once (at map construction) and read at assembly time."""

    ufl_shape = ()

    def __init__(self, fn):
        self._fn = fn

    def evaluate(self):
        return self._fn()

    @staticmethod
    def _value(x):
        return x.evaluate() if isinstance(x, Column) else x

    def __add__(self, other):
        return Column(lambda: self.evaluate() + Column._value(other))

    def __radd__(self, other):
        return Column(lambda: Column._value(other) + self.evaluate())

    def __sub__(self, other):
        return Column(lambda: self.evaluate() - Column._value(other))

    def __rsub__(self, other):
        return Column(lambda: Column._value(other) - self.evaluate())

    def __mul__(self, other):
        return Column(lambda: self.evaluate() * Column._value(other))

    def __rmul__(self, other):
        return Column(lambda: Column._value(other) * self.evaluate())

    def __truediv__(self, other):
        return Column(lambda: self.evaluate() / Column._value(other))

    def __neg__(self):
        return Column(lambda: -self.evaluate())


def evaluate_block(entries, rows=None):
    """``(points, n, n)`` values of a nested list of :class:`Column` expressions (``QuadratureFieldMap.jacobians[block]``): what
    a form compiler does with the UFL matrix at every quadrature point.  ``rows``: only these points."""
    n, m = len(entries), len(entries[0])
    first = np.asarray(Column._value(entries[0][0]))
    out = np.empty(((len(first) if rows is None else len(rows)), n, m))
    for i in range(n):
        for j in range(m):
            v = np.asarray(Column._value(entries[i][j]))
            out[:, i, j] = v if rows is None else v[rows]
    return out


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

    def __getitem__(self, i):
        """Component ``i`` as an expression (a UFL ``Indexed`` of the Function this stands for)."""
        if not 0 <= int(i) < self.dim:
            raise IndexError(i)
        return Column(lambda: self.values[:, int(i)])


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
        # quadrature_map.py:88-105: the block derivative() contracts, written in the entries of jacobian_flatten (and of the
        # stress for the "pack4" layout) -- the same expression the dolfinx class gets (AcceleratedUpdate._accel_packed_jacobians)
        layout = getattr(material, "tangent_layout", "full")
        self.jacobians, col = {}, 0
        for block, shape in material.tangent_blocks.items():
            if layout == "full":
                if self.jacobian_width != sum(int(np.prod(sh)) for sh in material.tangent_blocks.values()):
                    break   # a material that sizes its tangent its own way without naming a layout: no block expression
                jf = self.jacobian_flatten
                self.jacobians[block] = [[jf[col + shape[1] * i + j] for j in range(shape[1])] for i in range(shape[0])]
                col += int(np.prod(shape))
            else:
                self.jacobians[block] = tangent_entries(layout, self.jacobian_flatten, self.fluxes[block[0]], shape[0])
        self._accel_jacobian_layout = layout
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

    def tangent_block_values(self, block=None, rows=None):
        """``jacobians[block]`` evaluated at the quadrature points, ``(points, nf, ng)``: the full block whatever the layout of
        ``jacobian_flatten`` (what an assembly over this map consumes)."""
        if block is None:
            (block,) = self.jacobians
        return evaluate_block(self.jacobians[block], rows)

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
