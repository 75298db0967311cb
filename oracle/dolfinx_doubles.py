"""numpy-backed doubles of the few ``dolfinx`` / ``ufl`` / ``basix`` / ``mpi4py`` / ``petsc4py`` / ``jax`` names that the
reference's ``quadrature_map.py``, ``quadrature_function.py`` and ``utils.py`` touch while a ``QuadratureMap`` is built and
while ``update()`` / ``advance()`` / ``update_initial_state()`` run.

TEST INFRASTRUCTURE ONLY (build container only: it exists to import the reference from ``/root/reference``).  None of the
real libraries is installed here, so the reference's ``QuadratureMap`` -- the caller of the hot path, SURVEY.md section 8 rows
a1-a4 -- could never execute.  With these doubles its OWN code runs unmodified: ``QuadratureMap.__init__``,
``register_gradient``, ``QuadratureExpression.eval``, ``get_gradient_vals``, ``initialize_state``, ``update``,
``update_fluxes``, ``update_internal_state_variables``, ``advance``, ``_get_vals``, ``_update_vals``,
``_build_cell_to_dofs_map``, ``create_quadrature_functionspace`` are the reference's; only what they call INTO dolfinx is
replaced, by the data model those calls rely on:

* a quadrature ``fem.Function`` is ``x.array`` of ``num_cells * nqp * prod(shape)`` doubles, point = cell * nqp + q,
  component fastest (what ``_cell_to_dofs`` and ``_update_vals`` assume, ``quadrature_map.py:255-260``, ``utils.py:136-143``);
* ``fem.Expression(expr, points).eval(mesh, cells[, values])`` returns / fills ``(len(cells), nqp * dim)``; the "UFL
  expression" handed to ``register_gradient`` is an :class:`PointwiseExpression` around a callable ``cells -> rows``;
* UFL algebra is a scalar expression tree that can be EVALUATED per quadrature point: indexing a Function gives a
  :class:`Scalar` that reads the Function's current ``x.array`` column, ``+ - * /`` combine them, ``ufl.as_matrix`` /
  ``ufl.as_vector`` keep their entries (:class:`_Tensor`, ``.evaluate() -> (points, ...)``).  That is what lets a test read
  ``QuadratureMap.jacobians[block]`` (``quadrature_map.py:88-105``) numerically, the way a compiled form would at assembly.
  Forms themselves (``ufl.derivative``, measures) stay out: they are not on the update path.

Used by ``tests/test_reference_quadrature_map.py`` (the accelerated mixin over the real class against the real class) and by
``tests/golden/make_quadrature_map_golden.py`` (fields the reference's ``update()`` / ``advance()`` leave, as fixtures).
"""
from __future__ import annotations

import sys
import types

import numpy as np

_NAMES = ("dolfinx", "dolfinx.fem", "dolfinx.common", "ufl", "ufl.log", "basix", "basix.ufl", "mpi4py", "mpi4py.MPI", "petsc4py",
          "petsc4py.PETSc", "jax")

_NQP = {("hexahedron", 0): 1, ("hexahedron", 1): 1, ("hexahedron", 2): 8, ("quadrilateral", 2): 4, ("triangle", 2): 3,
        ("tetrahedron", 2): 4, ("tetrahedron", 1): 1, ("triangle", 1): 1}


class _IndexMap:
    def __init__(self, n):
        self.size_local, self.num_ghosts = n, 0


class _CellType:
    def __init__(self, name):
        self.name = name


class _Topology:
    def __init__(self, num_cells, cell, dim):
        self._n, self.dim, self.cell_type, self._cell = num_cells, dim, _CellType(cell), cell

    def index_map(self, dim):
        return _IndexMap(self._n)

    def create_connectivity(self, a, b):
        pass

    def cell_name(self):
        return self._cell


class _Geometry:
    def __init__(self, dim):
        self.dim = dim


class Mesh:
    """``num_cells`` cells of type ``cell``: all a quadrature space needs to know."""

    def __init__(self, num_cells, cell="hexahedron", dim=3):
        self.topology, self.geometry = _Topology(int(num_cells), cell, dim), _Geometry(dim)


class PointwiseExpression:
    """Stands for the UFL expression of a gradient: ``rows(cells) -> (len(cells) * nqp, dim)`` values."""

    def __init__(self, rows, dim):
        self.rows, self.ufl_shape = rows, (int(dim),)


class _Element:
    def __init__(self, cell, value_shape, degree):
        self.cell, self.value_shape, self.degree = cell, tuple(value_shape), degree


class _Dofmap:
    def __init__(self, num_cells, nqp, bs):
        self.list = np.arange(num_cells * nqp, dtype=np.int32).reshape(num_cells, nqp)
        self.bs = bs


class _Space:
    def __init__(self, mesh, element):
        self.mesh, self.element = mesh, element
        n = mesh.topology.index_map(0).size_local
        self.nqp = _NQP[(element.cell, element.degree)]
        self.value_size = int(np.prod(element.value_shape)) if element.value_shape else 1
        self.dofmap = _Dofmap(n, self.nqp, self.value_size)
        self.size = n * self.nqp * self.value_size


class _X:
    def __init__(self, n):
        self.array = np.zeros(n)


class Scalar:
    """A scalar UFL expression over the quadrature points, evaluated lazily from the Functions' current content."""

    ufl_shape = ()

    def __init__(self, fn):
        self._fn = fn

    def evaluate(self):
        return self._fn()

    @staticmethod
    def value(x):
        return x.evaluate() if isinstance(x, Scalar) else x

    def __add__(self, o):
        return Scalar(lambda: self.evaluate() + Scalar.value(o))

    def __radd__(self, o):
        return Scalar(lambda: Scalar.value(o) + self.evaluate())

    def __sub__(self, o):
        return Scalar(lambda: self.evaluate() - Scalar.value(o))

    def __rsub__(self, o):
        return Scalar(lambda: Scalar.value(o) - self.evaluate())

    def __mul__(self, o):
        return Scalar(lambda: self.evaluate() * Scalar.value(o))

    def __rmul__(self, o):
        return Scalar(lambda: Scalar.value(o) * self.evaluate())

    def __truediv__(self, o):
        return Scalar(lambda: self.evaluate() / Scalar.value(o))

    def __neg__(self):
        return Scalar(lambda: -self.evaluate())


class Function:
    def __init__(self, V, name=None):
        self.function_space, self.name = V, name
        self.ufl_shape = V.element.value_shape
        self.x = _X(V.size)

    def __len__(self):
        if not self.ufl_shape:
            raise TypeError("scalar Function")
        return self.ufl_shape[0]

    def __getitem__(self, i):
        dim = int(np.prod(self.ufl_shape)) if self.ufl_shape else 1
        if not 0 <= int(i) < dim:
            raise IndexError(i)
        return Scalar(lambda: self.x.array.reshape(-1, dim)[:, int(i)])


class Expression:
    def __init__(self, ufl_expression, points):
        self.ufl_expression, self._nqp = ufl_expression, len(points)

    def eval(self, mesh, cells, values=None):
        rows = np.asarray(self.ufl_expression.rows(cells), dtype=np.float64).reshape(len(cells), -1)
        if values is None:
            return np.array(rows)
        values[...] = rows
        return values


class _Tensor:
    """``ufl.as_matrix`` / ``ufl.as_vector`` of scalar expressions: keeps the entries."""

    def __init__(self, shape, entries=None):
        self.ufl_shape, self.entries = shape, entries

    def __getitem__(self, idx):
        e = self.entries
        for k in (idx if isinstance(idx, tuple) else (idx,)):
            e = e[k]
        return e

    def evaluate(self):
        """``(points,) + ufl_shape`` values."""
        flat = [e for row in self.entries for e in row] if len(self.ufl_shape) == 2 else list(self.entries)
        cols = [np.asarray(Scalar.value(e), dtype=np.float64) for e in flat]
        npts = max((c.shape[0] for c in cols if c.ndim), default=1)
        out = np.stack([np.broadcast_to(c, (npts,)) for c in cols], axis=1)
        return out.reshape((npts,) + tuple(self.ufl_shape))


class Timer:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _modules():
    m = {name: types.ModuleType(name) for name in _NAMES}
    dolfinx, fem, common = m["dolfinx"], m["dolfinx.fem"], m["dolfinx.common"]
    fem.Function, fem.Expression = Function, Expression
    fem.functionspace = lambda mesh, element: _Space(mesh, element)
    fem.Constant = lambda mesh, value: PointwiseExpression(lambda cells: None, 1)
    fem.petsc = types.SimpleNamespace()
    common.Timer = Timer
    dolfinx.fem, dolfinx.common = fem, common
    ufl = m["ufl"]
    ufl.Measure = lambda *a, **k: types.SimpleNamespace(args=a, kwargs=k)
    ufl.dx = types.SimpleNamespace()
    ufl.shape = lambda f: f.ufl_shape

    def as_matrix(rows):
        return _Tensor((len(rows), len(rows[0])), [list(r) for r in rows])

    ufl.as_matrix = as_matrix
    ufl.as_vector = lambda comps: _Tensor((len(comps),), list(comps))
    ufl.log = m["ufl.log"]
    ufl.log.UFLValueError = ValueError
    basix = m["basix"]
    basix.CellType = types.SimpleNamespace(hexahedron="hexahedron", quadrilateral="quadrilateral", triangle="triangle", tetrahedron="tetrahedron")
    basix.make_quadrature = lambda celltype, degree: (np.zeros((_NQP[(celltype, degree)], 3)), np.ones(_NQP[(celltype, degree)]))
    basix.ufl = m["basix.ufl"]
    basix.ufl.quadrature_element = lambda cell, value_shape=(), scheme="default", degree=2: _Element(cell, value_shape, degree)
    m["mpi4py"].MPI = m["mpi4py.MPI"]
    m["mpi4py.MPI"].COMM_WORLD = types.SimpleNamespace(rank=0, size=1)
    m["petsc4py"].PETSc = m["petsc4py.PETSc"]
    m["jax"].Array = type("Array", (), {})
    return m


class installed:
    """Context manager: the doubles and ``/root/reference`` importable inside, everything restored outside."""

    def __init__(self, reference_root):
        self.root = reference_root

    def __enter__(self):
        self._saved = {k: sys.modules.get(k) for k in list(sys.modules) if k in _NAMES or k == "dolfinx_materials" or k.startswith("dolfinx_materials.")}
        for k in self._saved:
            del sys.modules[k]
        sys.modules.update(_modules())
        sys.path.insert(0, self.root)
        try:
            import dolfinx_materials.quadrature_map as qm
        except BaseException:
            self.__exit__(None, None, None)
            raise
        return qm

    def __exit__(self, *exc):
        for k in [k for k in sys.modules if k in _NAMES or k == "dolfinx_materials" or k.startswith("dolfinx_materials.")]:
            del sys.modules[k]
        sys.modules.update({k: v for k, v in self._saved.items() if v is not None})
        if self.root in sys.path:
            sys.path.remove(self.root)
        return False
