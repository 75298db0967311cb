"""``QuadratureMap`` with the exchange around the hot call re-organised for this engine.

The reference's ``QuadratureMap.update()`` (``dolfinx_materials/quadrature_map.py:297-334``) spends, per global
Newton iteration and around ``material.integrate``: a scatter of the evaluated gradient into its quadrature Function
and a fancy-index gather back out of it (``quadrature_function.py:45-51``, ``quadrature_map.py:255-257``), a
``np.concatenate`` copy (``:313``), three full-array ``np.isnan`` passes (``:322-324``) and three scatters through an
index that ``_update_vals`` rebuilds on every call (``utils.py:136-143``: ``np.add.outer(cells * bs, arange(bs))`` --
for the 36-wide tangent of 1e7 points a 3.6e8-entry int64 array, 2.9 GB).  None of it is arithmetic of the law.

:class:`AcceleratedUpdate` is a mixin that overrides ``update() / advance() / initialize_state()`` in terms of the
reference class's public attribute surface only -- ``material, mesh, cells, dofs, gradients, fluxes,
internal_state_variables, external_state_variables, jacobian_flatten, rotation_func, _initialized`` -- and does the
same job with what the engine offers:

* a map over all cells (``dofs`` is the identity; the default, ``quadrature_map.py:66-70``) **binds** the ``x.array``
  of the flux and ``jacobian_flatten`` Functions as the material's output arrays (``HIPMaterial.bind_outputs``
  page-locks them in place): ``integrate`` delivers into them and nothing is scattered.  The gradient expression is
  evaluated straight into its own Function's memory (``Expression.eval(mesh, cells, values=...)``), which is page-locked
  too, so it is uploaded by DMA without a staging copy -- no gather, no concatenate;
* a map over a subset of cells keeps a persistent page-locked gradient buffer and delivers through ``self.dofs``, the
  point index the constructor already built (``quadrature_map.py:231-233, :259-260``) -- nothing is rebuilt per call: the
  engine stores every point's stress and tangent block straight into its row of the Functions
  (``HIPMaterial.integrate_rows``); rows of other arrays are moved on several threads (``scatter_rows``);
* NaNs are taken from the kernel's status record (``material.last_stats["n_nan"]``: stress, state AND tangent, like the
  three asserts of ``quadrature_map.py:322-324``) instead of three host passes.  One ordering difference: the reference asserts
  BEFORE it writes the Functions (``:331-334``), while a bound map's ``integrate`` has delivered flux, tangent and (default mode)
  the internal state variables into the Functions' memory by the time the assertion fires -- after a failed ``update()`` the
  Functions hold the non-finite result, not the previous one.  The material's state is the same on both sides (s1 is the failed
  update, s0 untouched), and the next successful ``update()`` rewrites every Function;
* internal state variables: ``isv_every_update = True`` (default) writes their Functions in every ``update()`` exactly like the
  reference (``:332``; +56 B/point over PCIe per update for J2).  Callers that read them between ``update()`` and ``advance()``
  only through the map opt in to ``"lazy"``: the Functions are refreshed on the first access to
  ``qmap.internal_state_variables`` / ``qmap.variables`` / ``qmap.project_on`` after an ``update()`` and at ``advance()`` -- what that
  mode cannot see is a Function object taken out of the dict EARLIER (or captured in a compiled form) and read directly in
  between, which is why it is not the default; ``False`` writes them at ``advance()`` only;
* the four phases carry the reference's timer names (``"dx_mat: ..."``, ``quadrature_map.py:302-331``), so ``list_timings`` keeps
  its rows;
* optionally the gradient is evaluated on the GPU from the displacement vector (``register_device_gradient``);
* a material created with a packed ``tangent_layout`` (``"sym"`` 21, ``"coef"`` 9, ``"pack4"`` 4 doubles per point instead of
  36) gets a ``jacobian_flatten`` of exactly that width and ``jacobians[block]`` -- the UFL matrix ``derivative()`` contracts
  (``quadrature_map.py:83-105, :132-158``) -- written in terms of it: an index-only change for ``"sym"``;
  ``c1 1x1 + c2 I + c3 n x n`` for ``"coef"``; the same with ``n = dev(stress) w`` taken from the flux Function the map already
  owns for ``"pack4"`` (the assembly-side consumer).  The host then receives 48 + 32 B/point of
  stress + tangent and rebuilds nothing; ``derivative()`` itself is inherited unchanged.

With a material that offers none of this (any duck-typed ``Material``: the oracle-backed one of the tests, a
``generic.Material`` subclass) the same methods fall back to plain row copies and host-side NaN checks; results are
identical either way (``tests/test_quadrature_map.py``, ``tests/test_gpu_quadrature_map.py``).

``QuadratureMap`` below is ``accelerate(dolfinx_materials.quadrature_map.QuadratureMap)`` -- ``type("QuadratureMap",
(AcceleratedUpdate, reference), {})`` -- when the reference package imports (the third import swap of INTEGRATION.md section 1);
``field_map.QuadratureFieldMap`` is the same mixin over a dolfinx-free stand-in base, which is how this code path is
tested where dolfinx is absent.
"""
from __future__ import annotations

import numpy as np

try:  # the reference wraps the four phases of update() in dolfinx Timers (quadrature_map.py:302-331)
    from dolfinx.common import Timer as _Timer
except Exception:  # dolfinx is optional for the engine itself
    import contextlib

    def _Timer(name):
        return contextlib.nullcontext()


def rows_of(fun, dim):
    """``(points, dim)`` view of the memory of a quadrature Function (``utils.py:98-104``): point = cell * nqp + q,
    component fastest."""
    return fun.x.array.reshape(-1, max(1, int(dim)))


def _slow_path_warning(what, exc):
    import warnings

    from . import PerformanceWarning

    warnings.warn(f"could not page-lock the memory of {what} in place ({exc}): results are delivered through staging copies", PerformanceWarning)


def _same_memory(a, b):
    return a.size == b.size and a.ctypes.data == b.ctypes.data


def sym_position(i, j, n=6):
    """Where entry ``(i, j)`` of a symmetric ``n x n`` block sits among its ``n (n + 1) / 2`` upper-triangle entries stored row by
    row (``conventions.SYM_IDX``; the kernels' ``TL_SYM`` order)."""
    i, j = (i, j) if i <= j else (j, i)
    return i * n - i * (i - 1) // 2 + (j - i)


def tangent_entries(layout, jf, flux=None, n=6):
    """The ``n x n`` tangent block of one point as a nested list of scalar expressions in the entries of ``jf`` (the tangent
    quadrature Function of the layout's width) and, for ``"pack4"``, of ``flux`` (the stress Function of the same update).
    Only indexing and ``+ - * /`` are used, so ``jf`` / ``flux`` may be UFL Functions (then ``to_mat`` of the result is what
    ``quadrature_map.py:92-104`` builds by indexing a 36-wide Function), :class:`field_map.Field` objects (numpy columns: the
    stand-in assembly evaluates the same expression) or component-major arrays (``tangent.T`` of an ``(N, width)`` array, so
    that ``jf[k]`` is component ``k`` of every point).

    * ``"full"``: ``jf[n i + j]`` (``quadrature_map.py:97``);
    * ``"sym"``: ``jf[sym_position(i, j)]`` -- 21 entries for a 6 x 6 block;
    * ``"coef"``: ``jf = (c1, c2, c3, n[0..5])``, ``Ct = c1 1x1 + c2 I + c3 n x n`` in the Mandel basis, ``1 = (1,1,1,0,0,0)``
      (``tests/mfront/IsotropicLinearHardeningPlasticity.mfront:66-69`` with ``M`` expanded, ``include/dxmat.h`` DXM_TANGENT_COEF);
    * ``"pack4"``: ``jf = (c1, c2, c3, w)`` and ``n = dev(flux) w`` -- the kernels build their own block from ``n`` in exactly
      this form (``csrc/small_strain.hpp`` step 5), so nothing is lost; elastic points carry ``c3 = 0, w = 0``."""
    if layout == "full":
        return [[jf[n * i + j] for j in range(n)] for i in range(n)]
    if layout == "sym":
        return [[jf[sym_position(i, j, n)] for j in range(n)] for i in range(n)]
    if layout not in ("coef", "pack4") or n != 6:
        raise ValueError(f"no tangent expression for layout {layout!r} and a {n} x {n} block")
    c1, c2, c3 = jf[0], jf[1], jf[2]
    if layout == "coef":
        nn = [jf[3 + k] for k in range(6)]
    else:
        if flux is None:
            raise ValueError("the 'pack4' layout rebuilds the flow direction from the stress: pass the flux Function")
        w = jf[3]
        third = (flux[0] + flux[1] + flux[2]) / 3
        nn = [(flux[k] - third) * w for k in range(3)] + [flux[k] * w for k in range(3, 6)]
    rows = []
    for i in range(6):
        row = []
        for j in range(6):
            # the kernels' association (small_strain.hpp tangent_pair): (c1 + c2) + c3 (n_i n_j), symmetric in (i, j) as written
            entry = c3 * (nn[i] * nn[j])
            if i < 3 and j < 3:
                entry = ((c1 + c2) if i == j else c1) + entry
            elif i == j:
                entry = c2 + entry
            row.append(entry)
        rows.append(row)
    return rows


class _LazyFields(dict):
    """``qmap.internal_state_variables``: name -> quadrature Function, refreshed from the device the first time anybody looks
    after an ``update()`` (``isv_every_update = "lazy"``).  Every read access goes through :meth:`_sync`; ``{**d}`` /
    ``dict(d)`` take the slow path (``keys`` + ``__getitem__``) because ``__iter__`` is overridden too."""

    _owner = None

    def _sync(self):
        owner = self._owner
        if owner is not None and owner.__dict__.get("_accel_isv_stale") and owner.isv_every_update == "lazy":
            owner.refresh_internal_state_variables()

    def __getitem__(self, key):
        self._sync()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._sync()
        return dict.get(self, key, default)

    def __iter__(self):
        self._sync()
        return dict.__iter__(self)

    def keys(self):
        self._sync()
        return dict.keys(self)

    def values(self):
        self._sync()
        return dict.values(self)

    def items(self):
        self._sync()
        return dict.items(self)

    def raw(self):
        """The Functions without a refresh (the mixin's own writes)."""
        return dict(dict.items(self))


class _Plan:
    """What is decided once per map (first ``update`` / ``advance``), not once per call."""

    identity = False          # dofs == arange(all points of the Functions): rows are the Functions' memory itself
    bound = False             # flux + jacobian_flatten are the material's output arrays
    npoints = 0
    grad_buffers = None       # subset maps: persistent (page-locked where the material can) gradient rows per name
    rows = None               # subset maps: self.dofs as a C-contiguous int64 index, made once
    row_outputs = False       # subset maps: the material delivers flux and tangent into the Functions' rows itself
    state_buffers = None      # subset maps: persistent (page-locked) landing rows of the final state per field (advance)
    device_gradient = None    # (mesh, displacement callable)
    bound_keys = ()           # what THIS map bound on the material (close() gives back exactly these)


class AcceleratedUpdate:
    """Mixin: ``update / advance / initialize_state`` of ``QuadratureMap`` around a batched engine."""

    #: when the internal state variables reach their Functions: ``True`` (default) in every ``update()`` like the reference
    #: (``quadrature_map.py:332``); opt-in ``"lazy"``: on the first access through the map after an ``update()`` and at ``advance()``;
    #: ``False``: at ``advance()`` only
    isv_every_update = True

    # ---- construction: the tangent Function and the matrix derivative() contracts -------------------------------
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        layout = getattr(self.material, "tangent_layout", "full")
        if layout != "full" and self.__dict__.get("_accel_jacobian_layout", "full") != layout:
            self._accel_packed_jacobians(layout)

    def _accel_packed_jacobians(self, layout):
        """``WJ`` / ``jacobian_flatten`` / ``jacobians`` of ``quadrature_map.py:83-105`` for a material that returns its tangent
        packed (``material.tangent_layout``: ``"sym"`` / ``"coef"`` / ``"pack4"``, ``material.tangent_size`` doubles per point):
        the Function gets the packed width and ``jacobians[block]`` is :func:`tangent_entries` of it -- what ``derivative()``
        (``:147, :156``) contracts with the variation of the gradient.  The reference constructor has built the 36-wide
        Function by then; it is dropped here (a transient allocation at construction, never one per update).

        ``"pack4"`` reads the stress Function of the map, which every ``update()`` writes together with the four coefficients,
        so the pair is consistent whenever a form is assembled; the tangent of a point is a function of its own stress, which
        ``derivative()`` treats as data (it differentiates F, not the block)."""
        m = self.material
        if len(m.tangent_blocks) != 1:
            raise ValueError("packed tangent layouts are defined for materials with one (flux, gradient) block")
        ((block, shape),) = m.tangent_blocks.items()
        if getattr(m, "rotation_matrix", None) is not None:
            raise ValueError("a packed tangent cannot be rotated block by block (quadrature_map.py:326-330): use tangent_layout='full'")
        if shape[0] != shape[1] or (layout in ("coef", "pack4") and shape[0] != 6):
            raise ValueError(f"tangent layout {layout!r} is not defined for a {shape} block")
        width = int(m.tangent_size)
        ns = self._accel_reference_namespace()
        self.WJ = ns["create_quadrature_functionspace"](self.mesh, self.degree, width)
        self.jacobian_flatten = ns["fem"].Function(self.WJ)
        flux = self.fluxes[block[0]] if layout == "pack4" else None
        self.jacobians = {block: ns["to_mat"](tangent_entries(layout, self.jacobian_flatten, flux, shape[0]))}
        self.__dict__["_accel_jacobian_layout"] = layout
        self.__dict__["_accel_jacobian_width"] = width

    def _accel_reference_namespace(self):
        """The module namespace of the reference class under this mixin (``fem``, ``create_quadrature_functionspace``,
        ``to_mat``: ``quadrature_map.py:4-11``) -- the names its own constructor built ``jacobian_flatten`` with."""
        for klass in type(self).__mro__:
            init = klass.__dict__.get("__init__")
            ns = getattr(init, "__globals__", None)
            if ns is not None and "create_quadrature_functionspace" in ns and "to_mat" in ns and "fem" in ns:
                return ns
        raise TypeError("a packed tangent layout needs the reference's QuadratureMap (or field_map.FieldMapBase) under AcceleratedUpdate")

    # ---- set-up, once ------------------------------------------------------------------------------------------
    def _accel_plan(self):
        plan = self.__dict__.get("_accel")
        if plan is not None:
            return plan
        plan = _Plan()
        m = self.material
        isv = self.__dict__.get("internal_state_variables")
        if isinstance(isv, dict) and not isinstance(isv, _LazyFields):   # same Functions, access-aware container
            lazy = _LazyFields(isv)
            lazy._owner = self
            self.internal_state_variables = lazy
        dofs = np.asarray(self.dofs)
        plan.npoints = len(dofs)
        widths = {name: max(1, int(dim)) for name, dim in {**m.fluxes, **m.internal_state_variables}.items()}
        total = {len(rows_of(f, widths[name])) for name, f in {**self.fluxes, **self._isv_functions()}.items()}
        total.add(len(self.jacobian_flatten.x.array) // self._jacobian_width())
        plan.identity = len(total) == 1 and total.pop() == plan.npoints and bool(np.array_equal(dofs, np.arange(plan.npoints)))
        if not plan.identity:
            plan.rows = np.ascontiguousarray(dofs, dtype=np.int64)
            plan.row_outputs = bool(len(self.fluxes) == 1 and plan.npoints > 0 and getattr(m, "supports_row_outputs", False))
        plan.grad_buffers, plan.state_buffers = {}, {}
        width = getattr(m, "tangent_size", None)
        if width is not None and int(width) != self._jacobian_width():
            raise ValueError(f"the material returns {width} tangent entries per point but jacobian_flatten holds {self._jacobian_width()} "
                             "(quadrature_map.py:83-105): the map was not constructed through AcceleratedUpdate.__init__ with this material")
        # results straight into the Functions: one flux, full-width tangent, and a material that can take caller arrays
        if plan.identity and plan.npoints > 0 and len(self.fluxes) == 1 and hasattr(m, "bind_outputs"):
            (flux_fun,) = self.fluxes.values()
            # page-locking caller memory can be refused (RLIMIT_MEMLOCK, a range that is registered already): the map
            # then works through the material's own arrays and row copies, slower but identical
            try:
                m.bind_outputs(flux=flux_fun.x.array, tangent=self.jacobian_flatten.x.array)
                plan.bound = True
                plan.bound_keys += ("flux", "tangent")
            except Exception as exc:
                _slow_path_warning("the flux / jacobian_flatten Functions", exc)
                if hasattr(m, "_unbind"):
                    for key in ("flux", "tangent"):
                        m._unbind(key)
            pin_state = getattr(m, "bind_state_outputs", None)
            fields = self._isv_functions()
            if plan.bound and pin_state is not None and fields:   # advance() downloads straight into these
                try:
                    pin_state({name: f.x.array for name, f in fields.items()})
                    plan.bound_keys += tuple("isv:" + name for name in fields)
                except Exception as exc:
                    _slow_path_warning("the internal-state Functions", exc)
        self.__dict__["_accel"] = plan
        return plan

    def _accel_deliveries(self, plan, want):
        """``isv_every_update = True`` with the ISV Functions bound: let the engine write them inside every ``integrate`` (its
        transfer pipeline carries the 56 B/point along: ``HIPMaterial.bind_state_outputs(deliver=True)``) instead of a second
        pass over the state after it.  Switched when the mode changes; returns whether the Functions are written by the call."""
        m = self.material
        names = tuple(k[4:] for k in plan.bound_keys if k.startswith("isv:"))
        if not names or not hasattr(m, "delivers_state_outputs") or set(names) != set(m.internal_state_variables):
            return False
        on = set(names) <= set(m.delivers_state_outputs)
        if bool(want) != on:
            fields = self._isv_functions()
            try:
                m.bind_state_outputs({name: fields[name].x.array for name in names}, deliver=bool(want))
            except Exception as exc:   # page-locking refused on the re-bind: the second pass keeps working through staged copies
                _slow_path_warning("the internal-state Functions", exc)
                plan.bound_keys = tuple(k for k in plan.bound_keys if not k.startswith("isv:"))
                return False
            on = bool(want)
        return on

    def _accel_row_deliveries(self, plan, want):
        """The same for a map over a SUBSET of the cells whose results go into rows of the Functions (``integrate_rows``): the ISV
        Functions over all cells are bound as row destinations (``bind_state_outputs(deliver=True, rows=True)``) and the threads that
        put stress and tangent block of point i into row ``dofs[i]`` put its state fields there too.  Returns whether the call
        writes the Functions."""
        m = self.material
        names = tuple(getattr(m, "internal_state_variables", {}))
        if not names or not getattr(m, "supports_row_state_outputs", False):
            return False
        on = set(names) <= set(m.delivers_state_outputs)
        if bool(want) == on:
            return on
        if want:
            fields = self._isv_functions()
            try:
                m.bind_state_outputs({name: fields[name].x.array for name in names}, deliver=True, rows=True)
                plan.bound_keys += tuple("isv:" + name for name in names if "isv:" + name not in plan.bound_keys)
            except Exception as exc:   # page-locking refused: the second pass (refresh_internal_state_variables) keeps working
                _slow_path_warning("the internal-state Functions", exc)
                for name in names:
                    m._unbind("isv:" + name)
                return False
            return True
        for name in names:
            m._unbind("isv:" + name)
        plan.bound_keys = tuple(k for k in plan.bound_keys if not k.startswith("isv:"))
        return False

    def _isv_functions(self):
        """name -> Function of the internal state variables, without triggering a lazy refresh."""
        d = self.internal_state_variables
        return d.raw() if isinstance(d, _LazyFields) else d

    def _jacobian_width(self):
        """Doubles per point of ``jacobian_flatten``: the packed width when the constructor built it for a packed layout."""
        packed = self.__dict__.get("_accel_jacobian_width")
        return int(packed) if packed else (int(sum(int(np.prod(shape)) for shape in self.material.tangent_blocks.values())) or 1)

    def register_device_gradient(self, mesh, displacement):
        """Evaluate the gradient on the GPU from the nodal vector ``displacement()`` returns (``gradient.Hex8Mesh`` /
        ``Tet4Mesh`` / ``SimplexMesh``; ``*.from_dolfinx(V, degree)`` builds them from a function space, and
        ``lambda: u.x.array`` is the callable): only that vector is uploaded per update, the step before the path
        (``quadrature_function.py:45-51``) runs inside the update kernel.  ``mesh`` holds the cells of this map in its
        order: all cells of the mesh for a map over everything, the connectivity restricted to ``self.cells`` for a map
        over a subset (``Hex8Mesh(coords, conn[cells])``)."""
        plan = self._accel_plan()
        if mesh.npoints != plan.npoints or not (plan.identity or plan.row_outputs):
            raise ValueError("device gradient evaluation needs a mesh object with exactly the cells of this map, in its order")
        if not hasattr(self.material, "integrate_displacement"):
            raise ValueError("this material cannot evaluate gradients on the device")
        plan.device_gradient = (mesh, displacement)

    def close(self):
        """Give the Functions' memory back: un-page-lock the arrays bound to the material.  Call before the Functions
        are destroyed when the map and the material do not die together."""
        plan = self.__dict__.pop("_accel", None)
        if plan is not None and hasattr(self.material, "_unbind"):
            for key in plan.bound_keys:   # what this map page-locked; bindings the user or another map made stay
                self.material._unbind(key)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- rows in / rows out ------------------------------------------------------------------------------------
    def _take(self, fun, dim):
        """Rows of ``fun`` that belong to this map, in the order the material sees them (a view for a map over all
        cells, a gathered copy otherwise: ``_get_vals(field)[self.dofs]``, ``quadrature_map.py:271, :283-289``)."""
        r = rows_of(fun, dim)
        plan = self._accel_plan()
        if plan.identity:
            return r
        gather = getattr(self.material, "gather_rows", None)   # on several threads where the material offers it
        return gather(r, plan.rows) if gather is not None else r[self.dofs]

    def _put(self, fun, dim, values):
        dst = rows_of(fun, dim)
        plan = self._accel_plan()
        src = np.asarray(values, dtype=np.float64).reshape(plan.npoints, dst.shape[1])
        if plan.identity:
            if not _same_memory(src, dst):          # bound outputs are already in place
                dst[...] = src
            return
        scatter = getattr(self.material, "scatter_rows", None)
        if scatter is not None:
            scatter(dst, plan.rows, src)
        else:
            dst[self.dofs] = src

    def _put_columns(self, funs, sizes, block):
        block = np.asarray(block)
        if len(sizes) == 1:
            ((name, dim),) = sizes.items()
            self._put(funs[name], dim, block)
            return
        col = 0
        for name, dim in sizes.items():
            w = max(1, int(dim))
            self._put(funs[name], dim, block[:, col:col + w])
            col += w

    # ---- gradients -----------------------------------------------------------------------------------------------
    def _evaluate_into(self, grad, out_rows):
        """``Expression.eval`` with the destination given (dolfinx >= 0.8 takes ``values=``): the compiled expression
        writes the rows of this map's cells, in this map's order, straight into ``out_rows``.  False when the
        expression object cannot do that."""
        expr = getattr(grad, "expression", None)
        if expr is None or not self.__dict__.get("_accel_eval_into", True):
            return False
        try:
            expr.eval(self.mesh, self.cells, values=out_rows.reshape(len(self.cells), -1))
            return True
        except TypeError:
            self.__dict__["_accel_eval_into"] = False   # an older Expression.eval(mesh, cells): ask once only
            return False

    def _gradient_rows(self, name, dim):
        plan = self._accel_plan()
        grad = self.gradients[name]
        if plan.identity:
            rows = rows_of(grad.function, dim)
            if name not in plan.grad_buffers:   # page-lock the Function's memory once: uploaded by DMA, no staging copy
                pin = getattr(self.material, "bind_inputs", None)
                if pin is not None and rows.size:
                    try:
                        pin(gradient=grad.function.x.array)
                        plan.bound_keys += ("gradient",)
                    except Exception as exc:   # uploaded through the library's staging ring instead
                        _slow_path_warning(f"the '{name}' gradient Function", exc)
                plan.grad_buffers[name] = rows
            if not self._evaluate_into(grad, rows):
                grad.eval(self.cells)           # the reference's route: scatter through the expression's dof table
            return rows
        buf = plan.grad_buffers.get(name)
        if buf is None:
            alloc = getattr(self.material, "pinned_array", None)
            buf = plan.grad_buffers[name] = alloc((plan.npoints, max(1, int(dim)))) if alloc else np.empty((plan.npoints, max(1, int(dim))))
        if not self._evaluate_into(grad, buf):
            buf[...] = self.get_gradient_vals(grad, self.cells)
        return buf

    def _gradient_block(self):
        sizes = self.material.gradients
        blocks = [self._gradient_rows(name, dim) for name, dim in sizes.items()]
        return blocks[0] if len(blocks) == 1 else np.concatenate(blocks, axis=1)

    # ---- the path ------------------------------------------------------------------------------------------------
    def initialize_state(self):
        """Initial state of the material from the current content of the flux / internal-state Functions and the
        gradients at the current configuration (``quadrature_map.py:281-295``)."""
        m = self.material
        state = {}
        plan = self._accel_plan()
        if plan.device_gradient is None:
            for name, dim in m.gradients.items():
                if name in self.gradients:
                    state[name] = np.array(self._gradient_rows(name, dim))
        for funs, sizes in ((self.fluxes, m.fluxes), (self._isv_functions(), m.internal_state_variables)):
            for name, dim in sizes.items():
                state[name] = np.array(self._take(funs[name], dim))
        m.set_initial_state_dict(state)
        self._initialized = True

    def update(self):
        """One constitutive update of the map's points (once per global Newton iteration, ``solvers.py:173-176``), in the
        reference's four timed phases (``quadrature_map.py:302-331``)."""
        if not self._initialized:
            self.initialize_state()
        m = self.material
        plan = self._accel_plan()
        with _Timer("dx_mat: External state variable update"):
            if getattr(self, "external_state_variables", None):
                self.update_external_state_variables()
        rotate = getattr(m, "rotation_matrix", None) is not None
        rows_mode = plan.row_outputs and not rotate
        grad = None
        with _Timer("dx_mat: Gradients evaluation"):
            if plan.device_gradient is None:   # (else: evaluated inside the update kernel from the displacement vector)
                grad = self._gradient_block()
        if rotate and grad is not None:   # in place, on the rows (quadrature_map.py:315-318); a bound gradient Function is re-evaluated next call
            m.rotate_gradients(grad.ravel(), self.rotation_func.x.array)
        flux = tangent = None
        if rows_mode:
            delivered = self._accel_row_deliveries(plan, self.isv_every_update is True)
        else:
            delivered = plan.bound and self._accel_deliveries(plan, self.isv_every_update is True)
        with _Timer("dx_mat: Material integration"):
            if rows_mode:
                # a map over a subset of the cells: the engine stores each point's stress and tangent block in its row of the
                # Functions (the index the constructor built, quadrature_map.py:231-233) -- no scatter afterwards
                (flux_fun,), (flux_dim,) = self.fluxes.values(), m.fluxes.values()
                out = (plan.rows, rows_of(flux_fun, flux_dim), rows_of(self.jacobian_flatten, self._jacobian_width()))
                if plan.device_gradient is not None:
                    mesh, displacement = plan.device_gradient
                    isv = m.integrate_displacement_rows(mesh, displacement(), *out)
                else:
                    isv = m.integrate_rows(grad, *out)
            elif plan.device_gradient is not None:
                mesh, displacement = plan.device_gradient
                flux, isv, tangent = m.integrate_displacement(mesh, displacement())
            else:
                flux, isv, tangent = m.integrate(grad)
            stats = getattr(m, "last_stats", None)
            if stats is not None and "n_nan" in stats:   # counted by the kernel: flux, state and tangent
                assert stats["n_nan"] == 0, "non-finite constitutive update"
            else:   # quadrature_map.py:322-324
                assert not np.any(np.isnan(flux))
                assert not np.any(np.isnan(isv))
                assert not np.any(np.isnan(tangent))
        if rotate:
            m.rotate_fluxes(np.asarray(flux).ravel(), self.rotation_func.x.array)
            m.rotate_tangent_operator(np.asarray(tangent).ravel(), self.rotation_func.x.array)
        with _Timer("dx_mat: Update values and tangent operators"):
            self.__dict__["_accel_rows_current"] = rows_mode     # rows mode: the flux Function holds the final flux already (advance)
            if not rows_mode:
                self._put_columns(self.fluxes, m.fluxes, flux)
                self._put(self.jacobian_flatten, self._jacobian_width(), tangent)
            self._last_isv = isv
            self.__dict__["_accel_isv_stale"] = bool(m.internal_state_variables) and not delivered
            if self.isv_every_update is True and not delivered:
                self.refresh_internal_state_variables()

    def refresh_internal_state_variables(self):
        """Internal state variables of the last ``update()`` into their Functions (what the reference does in every
        update, ``quadrature_map.py:332, :343-348``); with the engine this is where they are downloaded."""
        sizes = self.material.internal_state_variables
        self.__dict__["_accel_isv_stale"] = False
        if not sizes or getattr(self, "_last_isv", None) is None:
            return
        plan = self._accel_plan()
        reader = getattr(self.material, "read_final_state", None)
        fields = self._isv_functions()
        if plan.identity and reader is not None:   # device -> the Function's (page-locked) memory, one transfer per field
            for name, dim in sizes.items():
                reader(name, rows_of(fields[name], dim))
            return
        if reader is not None and hasattr(self.material, "pinned_array"):
            # a map over a subset of the cells: every field comes off the device as its own contiguous (points, dim) block into a
            # persistent page-locked landing area (one DMA transfer) and goes to the map's rows on the library's threads -- slicing
            # the interleaved (points, 7) array the material returns costs numpy two single-threaded strided copies per field
            # (66 of the 77 ms of such an update at 5e6 points, profiles/r06_packed_update.md)
            for name, dim in sizes.items():
                buf = plan.state_buffers.get(name)
                if buf is None:
                    buf = plan.state_buffers[name] = self.material.pinned_array((plan.npoints, max(1, int(dim))))
                reader(name, buf)
                self._put(fields[name], dim, buf)
            return
        self._put_columns(fields, sizes, np.asarray(self._last_isv))

    @property
    def variables(self):
        """``quadrature_map.py``'s merged dict of gradients, fluxes and state Functions -- with the state refreshed first."""
        if self.__dict__.get("_accel_isv_stale") and self.isv_every_update == "lazy":
            self.refresh_internal_state_variables()
        return super().variables

    def project_on(self, *args, **kwargs):
        """``quadrature_map.py:362-405`` on current fields (the lazy mode downloads the state here if an ``update()`` came since)."""
        if self.__dict__.get("_accel_isv_stale") and self.isv_every_update == "lazy":
            self.refresh_internal_state_variables()
        return super().project_on(*args, **kwargs)

    def advance(self):
        """Accept the increment: initial state <- final state in the material, final flux and internal state
        variables into the Functions (``quadrature_map.py:350-360``)."""
        m = self.material
        m.data_manager.update()
        plan = self._accel_plan()
        reader = getattr(m, "read_final_state", None)
        final = None
        self.__dict__["_accel_isv_stale"] = False   # written below
        # (the fields a delivering update() wrote are read again here, 56 B/point per INCREMENT: between that update and this call
        # the Functions may have been written by anybody -- the reference's own update_initial_state does, quadrature_map.py:262-279 --
        # and the reference's advance() overwrites whatever they hold with the accepted state, :356-360)
        for funs, sizes in ((self.fluxes, m.fluxes), (self._isv_functions(), m.internal_state_variables)):
            for name, dim in sizes.items():
                if plan.identity and reader is not None:
                    reader(name, rows_of(funs[name], dim))     # device -> the Function's memory, no intermediate array
                    continue
                if funs is self.fluxes and self.__dict__.get("_accel_rows_current"):
                    continue                                   # delivered into its rows by the last update
                if reader is not None and hasattr(m, "pinned_array"):
                    buf = plan.state_buffers.get(name)         # device -> page-locked rows (one DMA transfer) -> the map's rows
                    if buf is None:
                        buf = plan.state_buffers[name] = m.pinned_array((plan.npoints, max(1, int(dim))))
                    reader(name, buf)
                    self._put(funs[name], dim, buf)
                    continue
                if final is None:
                    final = m.get_final_state_dict()
                self._put(funs[name], dim, final[name])


def accelerate(reference_class):
    """``reference_class`` (``dolfinx_materials.quadrature_map.QuadratureMap``) with ``update() / advance() /
    initialize_state()`` of :class:`AcceleratedUpdate`: same constructor, same attributes, same forms."""
    return type("QuadratureMap", (AcceleratedUpdate, reference_class), {
        "__doc__": "``dolfinx_materials.quadrature_map.QuadratureMap`` with ``update() / advance() / initialize_state()`` of "
                   ":class:`dolfinx_materials_amd.quadrature_map.AcceleratedUpdate`.",
        "__module__": __name__,
    })


def _reference_class():
    try:
        from dolfinx_materials.quadrature_map import QuadratureMap as reference
    except Exception:   # dolfinx / ufl / basix / mpi4py / the reference package are not installed
        return None
    return reference


_reference = _reference_class()
if _reference is not None:
    QuadratureMap = accelerate(_reference)
else:
    class QuadratureMap:   # pragma: no cover - only reached where dolfinx is absent
        """Placeholder where the reference package does not import: constructing it says what is missing."""

        def __init__(self, *args, **kwargs):
            raise ImportError(
                "dolfinx_materials_amd.quadrature_map.QuadratureMap extends dolfinx_materials.quadrature_map.QuadratureMap: "
                "install dolfinx and dolfinx_materials (dolfinx-free callers use dolfinx_materials_amd.field_map.QuadratureFieldMap)")
