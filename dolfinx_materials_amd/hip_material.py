"""The reference's duck-typed ``Material`` protocol on top of ``libdxmat.so``.

``HIPMaterial(behavior)`` is used exactly like ``JAXMaterial(behavior)``
(reference ``dolfinx_materials/jaxmat.py:141-234``) or a ``generic.Material`` subclass
(``dolfinx_materials/generic.py:103-201``): ``QuadratureMap`` (``quadrature_map.py:51-360``)
consumes ``gradients / fluxes / internal_state_variables / tangent_blocks``,
``set_data_manager``, ``set_initial_state_dict``, ``integrate`` and ``data_manager.update()``.

Differences that are deliberate:
  * the state is carried from s0 to s1 properly (the reference discards the converted state,
    ``jaxmat.py:135-138``);
  * the persistent state lives on the GPU in SoA form; the dicts returned here are host copies.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np

from . import _lib
from ._lib import S0, S1, DxmError, Stats


try:  # the reference wraps the hot call in dolfinx Timers read back by the demos
    from dolfinx.common import Timer as _Timer  # (jaxmat.py:209-223, plane_elastoplasticity.py:240-249)
except Exception:  # dolfinx is optional for the engine itself
    import contextlib

    def _Timer(name):
        return contextlib.nullcontext()


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _as_c(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


from .lazy_rows import LazyFinalRows, LazyInitialRows, LazyISV, _reaper  # noqa: E402,F401  (re-exported: tests and callers import them from here)


class DataManager:
    """``update`` / ``revert`` of ``generic.py:204-216`` and ``jaxmat.py:30-43``."""

    def __init__(self, material: "HIPMaterial", ngauss: int):
        self._m = material
        self.ngauss = ngauss
        num_gradients = sum(material.gradients.values())
        num_fluxes = sum(material.fluxes.values())
        self.K = np.zeros((num_fluxes, num_gradients))  # generic.py:208, jaxmat.py:34

    def update(self):
        """s0 <- s1 (end of a converged increment; ``quadrature_map.py:355``)."""
        self._m._advance()

    def revert(self):
        """s1 <- s0."""
        self._m._revert()

    @property
    def s0(self):
        return self._m.get_initial_state_dict()

    @property
    def s1(self):
        return self._m.get_final_state_dict()


class HIPMaterial:
    """A constitutive behaviour integrated on an MI355X through ``libdxmat.so``."""

    def __init__(self, behavior, jit=True, device: int = 0, gradient_name=None, flux_name=None, tangent_layout="full",
                 lazy_isv=True, devices=None):
        """``JAXMaterial(behavior, jit=True)`` (``jaxmat.py:144``): ``jit`` is accepted for signature
        compatibility and has no effect -- the kernels are compiled ahead of time (or, for a traced /
        custom hardening law, by hipcc on construction).

        ``tangent_layout="sym"`` (small-strain laws only) makes ``integrate`` return the 21
        upper-triangle entries per point, ``(N, 21)``, instead of the full ``(N, 6, 6)`` block the
        reference's ``jacobian_flatten`` expects (``conventions.unpack_sym_tangent`` expands it);
        ``"coef"`` (J2 laws) the nine coefficients ``(c1, c2, c3, n[6])`` of ``Ct = c1 1x1 + c2 I + c3 n x n``,
        ``(N, 9)`` (``conventions.tangent_from_coefficients``; an assembly can use the rank structure directly);
        ``"pack4"`` only ``(c1, c2, c3, w)``, ``(N, 4)``: the flow direction is ``n = dev(stress) w`` by construction of the
        kernels, so a consumer that holds the stress of the same update rebuilds the block bit for bit
        (``conventions.tangent_from_pack4``, ``dxm_expand_tangent_pack4_device``; 80 B/point of stress + tangent).

        ``lazy_isv=True``: the ``isv`` array ``integrate`` returns is a :class:`LazyISV`, downloaded when it
        is first looked at; ``False`` downloads it in every call like the reference.

        ``devices=[0, 1, ..., G-1]``: ONE process, G GPUs.  The points are cut into G contiguous blocks
        (``sharding.ShardPlan``), each block has its own ``dxm_material`` handle with its state resident on its GPU,
        and a host-buffer ``integrate`` runs the G chunk pipelines side by side, every GPU's DMA delivering straight
        into its rows of the one host array -- G PCIe links for the PCIe-bound form, no collective, no gather
        (north_star: "reassemble ... into the dolfinx quadrature Function", i.e. into host memory of one process,
        ``quadrature_map.py:66-70``).  The device-pointer forms belong to one GPU and raise for such a material."""
        if tangent_layout not in ("full", "sym", "coef", "pack4"):
            raise ValueError("tangent_layout must be 'full', 'sym', 'coef' or 'pack4'")
        if not isinstance(jit, (bool, type(None))):
            raise TypeError("the second argument of JAXMaterial / HIPMaterial is `jit` (jaxmat.py:144); pass the GPU index as device=")
        self.jit = bool(jit)
        self.lazy_isv = bool(lazy_isv)
        self._serial = 0
        self._serial0 = 0     # counts the changes of s0 (advance, set_initial_state_dict)
        self._bound = {}
        self._delivered = set()    # ISV names that host-buffer calls write into bound arrays (bind_state_outputs(deliver=True))
        self.tangent_layout = tangent_layout
        self.behavior = behavior
        self.devices = [int(d) for d in devices] if devices is not None else [int(device)]
        if not self.devices:
            raise ValueError("devices must name at least one GPU")
        self.device = self.devices[0]
        custom = getattr(behavior, "custom_hardening", None)
        # a user-supplied hardening law lives in its own JIT-compiled copy of the library
        self._lib = _lib.load_custom(custom.expr_R, custom.expr_dR) if custom is not None else _lib.load()
        self._info = _lib.law_info(behavior.law, self._lib)
        self._gname = gradient_name or behavior.gradient_name
        self._fname = flux_name or behavior.flux_name
        self.material_properties = dict(behavior.flat_properties())  # jaxmat.py:146
        self._parts = []      # (handle, first point, one past the last point, device): one per GPU
        self._pool = None
        self._n = 0
        self.data_manager = None
        self.last_stats = None
        self.last_upload = None
        self.dt = 0.0
        self._warm = False

    def _chk(self, rc):
        return _lib.check(rc, self._lib)

    # ---- protocol: names and sizes ---------------------------------------------------------
    @property
    def name(self):
        return self.behavior.__class__.__name__

    @property
    def rotation_matrix(self):
        return None  # generic.py:129-131

    @property
    def gradients(self):
        return {self._gname: int(self._info.n_grad)}

    @property
    def fluxes(self):
        return {self._fname: int(self._info.n_flux)}

    @property
    def tangent_blocks(self):
        # generic.py:141-146
        return {
            (kf, kg): (vf, vg)
            for (kf, vf), (kg, vg) in zip(self.fluxes.items(), self.gradients.items())
        }

    @property
    def internal_state_variables(self):
        return {
            self._info.isv_name[f].decode(): int(self._info.isv_dim[f])
            for f in range(self._info.n_isv_fields)
        }

    @property
    def variables(self):
        return {**self.gradients, **self.fluxes, **self.internal_state_variables}

    @property
    def gradient_names(self):
        return list(self.gradients.keys())

    @property
    def flux_names(self):
        return list(self.fluxes.keys())

    @property
    def internal_state_variable_names(self):
        return list(self.internal_state_variables.keys())

    @property
    def tangent_size(self):
        """Doubles per point of the tangent array ``integrate`` returns (``dxm_tangent_size``): 36 / 81 for the
        full block, 21 / 9 for the ``"sym"`` / ``"coef"`` layouts."""
        nf, ng = int(self._info.n_flux), int(self._info.n_grad)
        return {"full": nf * ng, "sym": nf * (nf + 1) // 2, "coef": 9, "pack4": 4}[self.tangent_layout]

    @property
    def algorithmic_bytes_per_point(self):
        return int(self._info.algorithmic_bytes_per_point)

    @property
    def kernel_name(self):
        return self._lib.dxm_kernel_name(self._parts[0][0]).decode() if self._parts else ""

    # ---- protocol: parameters ----------------------------------------------------------------
    def update_material_property(self, key, value):
        """``generic.py:119-120``; here the change reaches the kernel parameters."""
        obj = self.behavior
        parts = key.split(".")
        for p in parts[:-1]:
            obj = getattr(obj, p)
        if not hasattr(obj, parts[-1]):
            raise ValueError(f"Unknown material property {key!r}")
        arr = np.asarray(value, dtype=np.float64).reshape(-1)
        if arr.size == 0:
            raise ValueError(f"empty value for material property {key!r}")
        # QuadratureMap.update_material_properties hands 0-d arrays for numbers and one value per Gauss point
        # for UFL-valued properties (quadrature_map.py:160-172); a uniform field is a number
        if not np.all(arr == arr[0]):
            raise NotImplementedError(
                f"material property {key!r} varies from point to point: the fused kernels take uniform parameters. "
                "Split the domain into one QuadratureMap per material (QuadratureMap(mesh, deg, material, cells=...), as the "
                "reference's multi-material demo does) -- the reference's own JAX back-end ignores per-point values altogether")
        value = float(arr[0])
        setattr(obj, parts[-1], value)
        self.material_properties[key] = value
        if self._parts:
            prm = np.asarray(self.behavior.params(), dtype=np.float64)
            for h, *_ in self._parts:
                self._chk(self._lib.dxm_set_params(h, prm.ctypes.data_as(C.POINTER(C.c_double)), prm.size))

    def default_properties(self):
        """``generic.py:122-123``: the base class's (empty) defaults -- the properties of a behaviour live in ``material_properties``
        (``jaxmat.py:146``: the flattened behaviour)."""
        return {}

    def set_newton(self, maxit=25, rtol=1e-14):
        for h in self._handles():
            self._chk(self._lib.dxm_set_newton(h, int(maxit), float(rtol)))
        self._newton = (int(maxit), float(rtol))

    # ---- protocol: life cycle ------------------------------------------------------------------
    def set_data_manager(self, ngauss):
        """Allocate device state for ``ngauss`` points (``generic.py:172-174``, ``jaxmat.py:195-197``)."""
        self.close()
        prm = np.asarray(self.behavior.params(), dtype=np.float64)
        self._n = int(ngauss)
        # contiguous blocks, one per GPU (the first n % G blocks hold one point more: sharding.ShardPlan.bounds)
        G = len(self.devices)
        q, r = divmod(self._n, G)
        lo = 0
        try:
            for k, dev in enumerate(self.devices):
                hi = lo + q + (1 if k < r else 0)
                h = self._lib.dxm_create(self.behavior.law, prm.ctypes.data_as(C.POINTER(C.c_double)), prm.size, hi - lo, dev)
                if not h:
                    raise DxmError(f"dxm_create failed: {_lib.last_error(self._lib)}")
                self._parts.append((h, lo, hi, dev))
                if self.tangent_layout != "full":
                    self._chk(self._lib.dxm_set_tangent_layout(h, {"sym": 1, "coef": 2, "pack4": 3}[self.tangent_layout]))
                lo = hi
        except Exception:
            self.close()
            raise
        if G > 1:
            from concurrent.futures import ThreadPoolExecutor

            self._pool = ThreadPoolExecutor(max_workers=G, thread_name_prefix="dxm-device")
        ng, nf = self._info.n_grad, self._info.n_flux
        # host mirrors of the fields that are not device state (gradient and flux of s0 / s1)
        self._grad = [self._initial_gradient(), self._initial_gradient()]
        self._flux = [np.zeros((self._n, nf)), np.zeros((self._n, nf))]
        # output arrays owned by the material: page-locked so that D2H runs at full PCIe rate, allocated when the first
        # host-buffer call needs them (a bound array or a device-pointer caller never does)
        self._ct_shape = {"full": (self._n, nf, ng), "sym": (self._n, nf * (nf + 1) // 2), "coef": (self._n, 9), "pack4": (self._n, 4)}[self.tangent_layout]
        self._pinned = {}
        self._out_isv = self._out_ct = None
        self._flux_buf = []
        self._flux_next = 0
        self.data_manager = DataManager(self, self._n)

    def _own(self, key, shape):
        if key not in self._pinned:
            self._pinned[key] = _lib.PinnedArray(shape)
            self._pinned[key].array[...] = 0.0
        return self._pinned[key].array

    def _ensure_outputs(self, isv=False):
        nf = self._info.n_flux
        if self._out_ct is None:
            self._out_ct = self._own("tangent", self._ct_shape)
        if not self._flux_buf:
            self._flux_buf = [self._own("flux0", (self._n, nf)), self._own("flux1", (self._n, nf))]
        if isv and self._out_isv is None:
            self._out_isv = self._own("isv", (self._n, self._info.n_isv_total))

    def _initial_gradient(self):
        g = np.zeros((self._n, self._info.n_grad))
        if self._info.n_grad == 9:
            g[:, :3] = 1.0  # F = I
        return g

    def close(self):
        """Release the device state.  Arrays already returned by ``integrate`` / the state dicts stay
        valid: their page-locked memory is owned by the arrays themselves (``_lib.PinnedArray``)."""
        for h, *_ in getattr(self, "_parts", []):
            self._lib.dxm_destroy(h)
        self._parts = []
        sc = self.__dict__.pop("_scratch_material", None)
        if sc is not None:
            sc.close()
        if getattr(self, "_pool", None) is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self._unbind()
        for p in getattr(self, "_pinned", {}).values():
            p.release()
        self._pinned = {}
        self._out_isv = self._out_ct = None
        self._flux_buf = []
        self._rows_checked = None

    # ---- protocol: external state variables (quadrature_map.py:195, :225) --------------------------
    def initialize_external_state_variable(self, name, values):
        raise NotImplementedError(
            f"external state variable {name!r}: the HIP laws (elastic, J2, FeFp J2) take none "
            "(neither do the jaxmat behaviours they replace: JAXMaterial has no such method either)")

    def update_external_state_variable(self, name, values):
        self.initialize_external_state_variable(name, values)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _handles(self):
        """The handle of every block (one per GPU), in point order."""
        if not self._parts:
            raise DxmError("set_data_manager(ngauss) must be called first")
        return [p[0] for p in self._parts]

    def _require(self):
        """THE handle: for the calls that belong to one GPU (device pointers, streams, graphs)."""
        hs = self._handles()
        if len(hs) > 1:
            raise DxmError("this call addresses one GPU: not available for a material spread over several devices (devices=[...])")
        return hs[0]

    @property
    def _handle(self):
        return self._parts[0][0] if len(self._parts) == 1 else None

    def _blocks(self, *arrays):
        """(handle, first point, end, [address of the block's rows in each C-contiguous (N, ...) array | None]) per GPU."""
        return [(h, lo, hi, [None if a is None else a.ctypes.data + lo * a.strides[0] for a in arrays]) for h, lo, hi, _ in self._parts]

    def _run(self, calls):
        """Run one callable per block: side by side on the per-GPU threads when there are several."""
        if len(calls) <= 1 or self._pool is None:
            return [c() for c in calls]
        return [f.result() for f in [self._pool.submit(c) for c in calls]]

    # ---- protocol: state dictionaries ------------------------------------------------------------
    def _isv_dict(self, which):
        self._handles()
        out = {}
        for f, (name, dim) in enumerate(self.internal_state_variables.items()):
            a = np.empty((self._n, dim))
            self._run([lambda h=h, ptr=ptrs[0], f=f: self._chk(self._lib.dxm_get_state(h, which, f, ptr))
                       for h, lo, hi, ptrs in self._blocks(a) if hi > lo])
            out[name] = a
        return out

    def _hand_out(self, mirror):
        """What a state dictionary shows for a gradient / flux mirror.  The reference's dictionaries hold copies
        (``generic.py:265-277``: fancy-indexed rows of the state manager), so what goes out must not change under the caller:
        a lazy mirror goes out as a view it can count (:class:`LazyInitialRows`, :class:`LazyFinalRows`: downloaded only if
        still alive when the state it stands for is replaced); an array that a later update writes again (the material's two alternating page-locked flux buffers, a bound
        flux or gradient array, i.e. a Function's memory) goes out as a copy; everything else -- the array the caller passed to
        ``integrate`` (kept by reference, theirs to change), snapshots, placeholders -- as it is."""
        if type(mirror) in (LazyInitialRows, LazyFinalRows):
            return mirror._view()
        if isinstance(mirror, np.ndarray) and mirror.size and mirror.strides[0] != 0:
            bound = self._bound.get("gradient")
            if any(mirror is b for b in self._flux_buf) or (bound is not None and np.may_share_memory(mirror, bound)):
                return self._snapshot(mirror)
        return mirror

    def get_initial_state_dict(self):
        self._handles()
        return {self._gname: self._hand_out(self._grad[0]), self._fname: self._hand_out(self._flux[0]), **self._isv_dict(S0)}

    def get_final_state_dict(self):
        self._handles()   # (after revert() the s1 mirrors ARE the s0 mirrors)
        return {self._gname: self._hand_out(self._grad[1]), self._fname: self._hand_out(self._flux[1]), **self._isv_dict(S1)}

    def set_initial_state_dict(self, state):
        """``generic.py:200-201`` / ``quadrature_map.py:279,294``: any subset of the fields."""
        self._handles()
        isv_names = self.internal_state_variable_names
        unknown = [k for k in state if k not in self.variables]
        assert len(unknown) == 0, "Material state contains unknown field to update with."
        for key, value in state.items():
            dim = self.variables[key]
            a = _as_c(value, (self._n, max(1, dim)))
            if key == self._gname:
                self._retire_initial_views((0,))
                self._grad[0] = a.copy()
                self._serial0 += 1
            elif key == self._fname:
                self._retire_initial_views((1,))
                self._flux[0] = a.copy()
                self._serial0 += 1
            elif key == "be_bar" and self._info.n_grad == 9:
                continue  # handled below together with F
            else:
                self._retire_final_views(materializing=True)   # dxm_set_state gives s1 its own storage back: its device copies go
                self._set_state(isv_names.index(key), a)
        if self._info.n_grad == 9 and ("be_bar" in state or self._gname in state):
            # the kernel's state is the isochoric Cp^-1 (hidden field 2), rebuilt from (F_n, be_bar_n)
            from .conventions import cp_bar_inv_from_be_bar

            be = _as_c(state["be_bar"], (self._n, 6)) if "be_bar" in state else self._isv_dict(S0)["be_bar"]
            cpi, be = cp_bar_inv_from_be_bar(self._grad[0], be)
            self._retire_final_views(materializing=True)
            self._set_state(1, _as_c(be))
            self._set_state(2, _as_c(cpi))

    def _set_state(self, field, a):
        for h, lo, hi, ptrs in self._blocks(a):
            if hi > lo:
                self._chk(self._lib.dxm_set_state(h, S0, field, ptrs[0]))

    def _retire_initial_views(self, replaced):
        """s0 is about to be replaced in the fields named by `replaced` (0 gradient, 1 flux): if a view of the OLD s0 that a state
        dictionary handed out is still alive, the mirror takes its rows off the device now and its views keep showing them."""
        for kind in replaced:
            v = (self._grad, self._flux)[kind][0]
            if type(v) is LazyInitialRows and not v._frozen and v._handed_out():
                v._freeze()

    def _retire_final_views(self, materializing=False):
        """s1 is about to be replaced (an update of any form, ``revert``): a lazy s1 flux mirror takes its rows off the device
        first if a view of it is still alive -- the launch invalidates the device copy.  ``materializing``
        (``set_initial_state_dict`` of an internal state variable): s1 stays, but an s1 that is served from s0 gets its own
        storage back and the device drops the copies it showed; lazy s1 mirrors then settle whether or not a view is alive."""
        drop_all = materializing and self.__dict__.get("_s1_from_s0", False)
        for v in (self._grad[1], self._flux[1]) if getattr(self, "_flux", None) is not None else ():
            if type(v) in (LazyFinalRows, LazyInitialRows) and not v._frozen and (drop_all or (type(v) is LazyFinalRows and v._handed_out())):
                v._freeze()
        self._s1_from_s0 = False   # written in full by the caller's launch, or materialised by dxm_set_state

    def _advance(self):
        self._handles()
        # every s0 mirror that is about to be replaced settles with the views handed out of it while the device still holds its
        # rows (after revert() the s1 mirror IS the s0 mirror and stays: nothing to settle)
        self._retire_initial_views([kind for kind, pair in enumerate((self._grad, self._flux)) if pair[1] is not pair[0]])
        # An s1 mirror that IS the s0 mirror stays -- but the device may be about to drop the rows it stands for: after
        # revert() + set_initial_state_dict(...) s1 has its own storage again and brings no copies along, so dxm_advance leaves s0
        # without any.  The mirror takes them off the device first (rare; one download).
        for kind, pair in enumerate((self._grad, self._flux)):
            v = pair[0]
            if pair[1] is v and type(v) is LazyInitialRows and not v._frozen:
                if not all(max(0, int(self._lib.dxm_io_held(h, S1))) & (1 << kind) for h, lo, hi, _dev in self._parts if hi > lo):
                    v._freeze()
        held = 3
        for h, lo, hi, _dev in self._parts:
            self._chk(self._lib.dxm_advance(h))
            if hi > lo:
                held &= max(0, int(self._lib.dxm_io_held(h, S0)))
        # A bound gradient / flux array is overwritten by the next update, so the s0 mirror cannot be that array.  When
        # every handle kept its device copy (bind_* set option keep_initial_io; bit 0 gradient, bit 1 flux) the mirror is
        # a lazy view of it, else a snapshot of the array.  An s1 mirror that is the s0 mirror stays (advance after revert: s0
        # did not change).
        old = (self._grad[0], self._flux[0])
        new = []
        for kind, (cur, key) in enumerate(((self._grad[1], "gradient"), (self._flux[1], "flux"))):
            if isinstance(cur, LazyFinalRows):     # results of integrate_rows: the device copy became that of s0 ...
                if cur._frozen or not held & (1 << kind):   # ... unless the device dropped it (settled on the host beforehand)
                    new.append(np.asarray(cur))
                else:
                    new.append(LazyInitialRows(self, cur.shape, kind))
            elif isinstance(cur, LazyInitialRows):
                new.append(cur)
            elif isinstance(cur, np.ndarray) and cur.size and cur.strides[0] == 0:   # the "unknown" placeholder of a device-pointer call
                new.append(cur)
            elif key not in self._bound:
                new.append(cur)
            elif held & (1 << kind):
                new.append(LazyInitialRows(self, cur.shape, kind))
            else:
                new.append(self._snapshot(cur))
        self._grad[0], self._flux[0] = new
        self._serial0 += 1
        self._s1_from_s0 = True
        for a in old:   # the mirrors of the increment before: freed off this thread
            if isinstance(a, np.ndarray) and a is not self._grad[0] and a is not self._flux[0] and not any(a is b for b in self._flux_buf):
                _reaper.drop(a)
        del old, a, new

    def _fetch_io_rows(self, which, kind):
        """Download the gradient (0) / flux (1) of s0 / s1 from the device copies of the last host-buffer call
        (:class:`LazyInitialRows`, :class:`LazyFinalRows`)."""
        out = np.empty((self._n, int(self._info.n_flux if kind else self._info.n_grad)))
        self._run([lambda h=h, ptr=ptrs[0]: self._chk(self._lib.dxm_get_io(h, which, kind, ptr))
                   for h, lo, hi, ptrs in self._blocks(out) if hi > lo])
        return out

    def _snapshot(self, a):
        """Copy of a (large, C-contiguous) array on several threads (``dxm_host_copy``; numpy copies on one)."""
        if not (a.flags.c_contiguous and a.nbytes >= (4 << 20)):
            return a.copy()
        out = np.empty_like(a)
        self._chk(self._lib.dxm_host_copy(_ptr(out), _ptr(a), a.nbytes, 8))
        return out

    def _revert(self):
        self._handles()
        self._retire_final_views()
        for h in self._handles():
            self._chk(self._lib.dxm_revert(h))
        self._grad[1] = self._grad[0]
        self._flux[1] = self._flux[0]
        self._serial += 1   # s1 changed: lazy ISV views refetch
        self._s1_from_s0 = True

    # ---- protocol: the hot path -----------------------------------------------------------------
    def integrate(self, gradients, dt=0):
        """``(N, ng)`` host gradients -> ``(flux (N,nf), isv (N,sum isv), Ct (N,nf,ng))``.

        Same contract as ``JAXMaterial.integrate`` (``jaxmat.py:208-234``).  The returned arrays
        are page-locked buffers owned by the material and reused from call to call (the flux
        alternates between two so that the s0 mirror survives): they are valid until the next
        ``integrate``, which is how ``QuadratureMap.update`` consumes them -- it copies into the
        quadrature Functions right away (``utils.py:140-143``); the reference returns views of its
        state manager too (``generic.py:185-189``).  The gradient array is kept by reference as
        the s1 gradient mirror.
        """
        self._handles()
        self._refuse_with_row_deliveries("integrate")
        ng, nf = self._info.n_grad, self._info.n_flux
        # the reference's four timer names (jaxmat.py:209, :215, :218, :223), so that scripts reading
        # timing("jaxmat: ...") keep working when dolfinx is present
        with _Timer("jaxmat: dolfinx to jaxmat conversion"):
            g = _as_c(gradients)
            if g.shape != (self._n, ng):
                raise ValueError(f"gradients must have shape {(self._n, ng)}, got {g.shape}")
            eager = not self.lazy_isv
            self._ensure_outputs(isv=eager)
            self._retire_final_views()
            flux = self._next_flux_buffer()
            old = self._grad[1]
            self._grad[1] = g
        timer_name = "jaxmat: Constitutive update" if self._warm else "jaxmat: First pass (includes jit compilation)"
        self._warm = True
        with _Timer(timer_name):
            rc = self._integrate_blocks(self._lib.dxm_integrate, None, g, float(dt), flux, self._out_isv if eager else None)
        with _Timer("jaxmat: jaxmat to dolfinx conversion"):
            if rc > 0:
                warnings.warn(
                    f"local Newton did not converge at {rc} quadrature points", RuntimeWarning
                )
            self._flux[1] = flux
            self._serial += 1
            isv = self._out_isv if eager else LazyISV(self, (self._n, self._info.n_isv_total))
            # the gradient array of the previous call: released off this thread, after the transfers of this call
            if isinstance(old, np.ndarray) and old is not g and old is not self._grad[0]:
                _reaper.drop(old)
            del old
        return flux, isv, self._out_ct

    # ---- protocol: the explicit-state callables (jaxmat.py:147-164, generic.py:115-117, docs/jax.md:46-50) ---------------
    def natural_state(self, n=1):
        """The state the law starts from, as a state dictionary of ``n`` points (``behavior.init_state(n)`` of ``jaxmat.py:35``):
        zero strain / ``F = I``, zero flux, zero internal state variables, ``be_bar = I`` for the FeFp laws
        (``finite_strain_elastoplasticity.py:181``)."""
        n = int(n)
        g = np.zeros((n, int(self._info.n_grad)))
        if self._info.n_grad == 9:
            g[:, :3] = 1.0
        out = {self._gname: g, self._fname: np.zeros((n, int(self._info.n_flux)))}
        for name, dim in self.internal_state_variables.items():
            a = np.zeros((n, max(1, dim)))
            if name == "be_bar":
                a[:, :3] = 1.0
            out[name] = a
        return out

    def _scratch(self, n):
        """A second material of the same behaviour for ``n`` points (same device, same library, full tangent blocks, same Newton
        controls): the explicit-state callables integrate on it, so the state of THIS material (s0 / s1, its mirrors, its bound
        arrays) is never touched.  Kept for the next call of the same size."""
        sc = self.__dict__.get("_scratch_material")
        if sc is None or sc._n != n or not sc._parts:
            if sc is not None:
                sc.close()
            sc = HIPMaterial(self.behavior, device=self.device, gradient_name=self._gname, flux_name=self._fname, lazy_isv=False)
            sc.set_data_manager(n)
            self._scratch_material = sc
        prm = np.asarray(self.behavior.params(), dtype=np.float64)   # (update_material_property may have changed them since)
        for h in sc._handles():
            sc._chk(sc._lib.dxm_set_params(h, prm.ctypes.data_as(C.POINTER(C.c_double)), prm.size))
        if self.__dict__.get("_newton") and sc.__dict__.get("_newton") != self._newton:
            sc.set_newton(*self._newton)
        return sc

    def batched_constitutive_update(self, gradients, state, dt=0):
        """``Ct, new_state = material.batched_constitutive_update(gradients, state, dt)`` with the state passed in and handed
        back EXPLICITLY -- the attribute ``JAXMaterial.__init__`` builds as ``jit(vmap(jacfwd(constitutive_update, argnums=0,
        has_aux=True), in_axes=(0, 0, None)))`` (``jaxmat.py:147-155``; ``generic.py:115-117`` for the Python materials) and
        ``integrate`` calls (``jaxmat.py:219-221``).

        ``gradients``: ``(N, ng)``.  ``state``: dict name -> ``(N, dim)`` as the state dictionaries of this class
        (``get_initial_state_dict()``, :meth:`natural_state`): the internal state variables the update starts from; for the FeFp
        laws also the previous ``F`` (the kernel's state is rebuilt from ``(F_n, be_bar_n)``); other keys (the previous strain and
        stress of the small-strain laws) are not read by these laws, absent keys take the natural state.  Returns the consistent
        tangent ``(N, nf, ng)`` -- ``d flux / d gradient`` of the algorithm, what ``jacfwd`` differentiates -- and the new state
        (gradient, flux and every internal state variable, ``(N, dim)`` each, owned by the caller).  The material's own
        ``s0`` / ``s1`` are not touched: the update runs on a scratch material of the same behaviour (one ``dxm_integrate``)."""
        ng, nf = int(self._info.n_grad), int(self._info.n_flux)
        g = _as_c(gradients)
        if g.ndim != 2 or g.shape[1] != ng:
            raise ValueError(f"gradients must have shape (N, {ng}), got {g.shape}")
        n = g.shape[0]
        unknown = [k for k in state if k not in self.variables]
        assert len(unknown) == 0, "Material state contains unknown field to update with."
        start = self.natural_state(n)
        for key, value in state.items():
            start[key] = _as_c(value, start[key].shape)
        sc = self._scratch(n)
        wanted = list(self.internal_state_variables) + ([self._gname] if ng == 9 else [])
        sc.set_initial_state_dict({k: start[k] for k in wanted})
        flux, isv, ct = sc.integrate(g, float(dt))
        new_state = {self._gname: g.copy(), self._fname: np.array(flux)}
        col = 0
        isv = np.asarray(isv)
        for name, dim in self.internal_state_variables.items():
            w = max(1, dim)
            new_state[name] = np.array(isv[:, col:col + w])
            col += w
        return np.array(ct).reshape(n, nf, ng), new_state

    def constitutive_update(self, gradients, state, dt=0):
        """``sig, new_state = material.constitutive_update(eps, state, dt)`` at ONE material point (``jaxmat.py:158-164``,
        ``docs/jax.md:46-50``): ``gradients`` ``(ng,)``, ``state`` dict name -> ``(dim,)`` (absent keys: the natural state); returns
        the flux ``(nf,)`` and the new state with ``(dim,)`` entries.  (The reference vmaps / differentiates this function; here
        the batched form is the primitive and this is its one-point case.)"""
        g = np.asarray(gradients, dtype=np.float64).reshape(1, -1)
        _, new_state = self.batched_constitutive_update(g, {k: np.asarray(v, dtype=np.float64).reshape(1, -1) for k, v in state.items()}, dt)
        return new_state[self._fname][0], {k: v[0] for k, v in new_state.items()}

    def _refuse_with_row_deliveries(self, what):
        if self.__dict__.get("_delivers_rows"):
            raise DxmError(f"{what}: internal state variables are bound for delivery into ROWS of larger arrays (bind_state_outputs(rows=True)): "
                           "call integrate_rows / integrate_displacement_rows, or unbind first")

    def _fetch_isv(self):
        """Download the ISVs of the current s1 (for :class:`LazyISV`)."""
        self._ensure_outputs(isv=True)
        self._run([lambda h=h, ptr=ptrs[0]: self._chk(self._lib.dxm_isv_host(h, S1, ptr))
                   for h, lo, hi, ptrs in self._blocks(self._out_isv) if hi > lo])
        return self._out_isv

    def _integrate_blocks(self, entry, mesh_handle, src, dt, flux, isv):
        """``dxm_integrate`` (``mesh_handle`` None: every block reads its rows of the gradient array ``src``) or
        ``dxm_integrate_displacement`` (one block: the whole displacement vector ``src``) per block, side by side over
        the GPUs, each delivering into its rows of ``flux`` / ``isv`` / the tangent array.  Sets ``last_stats`` (sums over
        the blocks) and returns the number of points whose local Newton did not converge."""
        recs = [Stats() for _ in self._parts]
        calls = []
        for (h, lo, hi, p), st in zip(self._blocks(src if mesh_handle is None else None, flux, isv, self._out_ct), recs):
            if mesh_handle is None:
                calls.append(lambda h=h, p=p, st=st: entry(h, p[0], dt, p[1], p[2], p[3], C.byref(st)))
            else:
                calls.append(lambda h=h, p=p, st=st: entry(h, mesh_handle, _ptr(src), dt, p[1], p[2], p[3], C.byref(st)))
        return self._finish_blocks(self._run_over_one_array(calls, src if mesh_handle is None else None), recs)

    def _run_over_one_array(self, calls, src):
        """Several GPUs read their rows of ONE pageable gradient array: page-locked once for the call here -- the blocks'
        sub-ranges share pages at their boundaries, so the per-handle registration of ``dxm_integrate`` would be refused for
        every second block (and staged instead).  A refusal here leaves that per-handle behaviour."""
        locked = False
        if len(calls) > 1 and src is not None and src.nbytes >= (1 << 20):
            locked = self._lib.dxm_host_register(_ptr(src), src.nbytes) == 0
        try:
            return self._run(calls)
        finally:
            if locked:
                self._lib.dxm_host_unregister(_ptr(src))

    def _finish_blocks(self, rcs, recs):
        for rc in rcs:
            self._chk(rc)
        self.last_upload = recs[0].upload_mode   # how the gradient array reached the GPU (first block)
        tot = {"n_points": 0, "n_plastic": 0, "n_not_converged": 0, "n_nan": 0, "max_local_iters": 0}
        for st in recs:
            d = st.as_dict()
            for k in ("n_points", "n_plastic", "n_not_converged", "n_nan"):
                tot[k] += d[k]
            tot["max_local_iters"] = max(tot["max_local_iters"], d["max_local_iters"])
        self.last_stats = tot
        return sum(rc for rc in rcs if rc > 0)

    def _next_flux_buffer(self):
        """Two pinned flux buffers alternate so that the s0 mirror (the flux of the last converged
        increment) is never the one being overwritten."""
        if len(self._flux_buf) == 1:   # bound output array
            return self._flux_buf[0]
        cand = self._flux_buf[self._flux_next]
        if cand is self._flux[0]:
            self._flux_next ^= 1
            cand = self._flux_buf[self._flux_next]
        self._flux_next ^= 1
        return cand

    @property
    def supports_row_outputs(self):
        """Whether :meth:`integrate_rows` exists for this material (every law, every tangent layout: a packed layout's rows are
        ``tangent_size`` wide and are moved to their rows as they are, the full blocks are rebuilt there)."""
        return True

    def _check_rows(self, rows, flux, tangent):
        nf, nt = self._info.n_flux, self.tangent_size
        if not (isinstance(rows, np.ndarray) and rows.dtype == np.int64 and rows.flags.c_contiguous and rows.shape == (self._n,)):
            raise ValueError(f"rows must be a C-contiguous int64 array of {self._n} entries")
        for name, arr, w in (("flux", flux, nf), ("tangent", tangent, nt)):
            if not (isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.flags.c_contiguous and arr.size % w == 0):
                raise ValueError(f"{name} must be a C-contiguous float64 array of whole rows of {w}")
        total = min(flux.size // nf, tangent.size // nt)
        # every call: dxm_integrate_rows does not range-check the index, and neither the address of the index array nor its
        # length says that its CONTENT is still the one checked last time (~1 ms per 1e7 entries on the library's threads)
        lo, hi = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.dxm_host_index_range(rows.ctypes.data, self._n, 8, C.byref(lo), C.byref(hi)))
        if self._n and (lo.value < 0 or hi.value >= total):
            raise ValueError(f"rows must lie in [0, {total})")
        if not self.__dict__.get("_rows_checked"):
            self._rows_checked = True
            self.set_option("keep_initial_io", 1)   # the contiguous flux exists on the device only: advance keeps it for s0

    def _after_rows(self, rc):
        if rc > 0:
            warnings.warn(f"local Newton did not converge at {rc} quadrature points", RuntimeWarning)
        self._flux[1] = LazyFinalRows(self, (self._n, self._info.n_flux), 1)
        self._serial += 1
        return LazyISV(self, (self._n, self._info.n_isv_total))

    def integrate_rows(self, gradients, rows, flux, tangent, dt=0):
        """``integrate`` for a map over a SUBSET of the cells (``dxm_integrate_rows``): ``gradients`` are this material's
        ``(N, ng)`` points as usual, but ``flux`` / ``tangent`` are the arrays of the quadrature Functions over ALL cells
        -- ``(M, nf)`` / ``(M, tangent_size)`` (or flat; 36 / 81 for the full layout, 21 / 9 / 4 for the packed ones), ``M >= N`` -- and
        point ``i`` is delivered into their row ``rows[i]``: what
        ``_update_vals(field, values, cells)`` does with one fancy assignment per array per update
        (``utils.py:136-143``), done by the threads that rebuild the tangent blocks.  ``rows``: C-contiguous int64, each row
        once (``QuadratureMap.dofs``).  Returns the internal state variables (lazily, like ``integrate``); the flux of the
        final state in ``get_final_state_dict()`` is a :class:`LazyFinalRows`."""
        self._handles()
        g = _as_c(gradients)
        if g.shape != (self._n, self._info.n_grad):
            raise ValueError(f"gradients must have shape {(self._n, self._info.n_grad)}, got {g.shape}")
        self._check_rows(rows, flux, tangent)
        self._retire_final_views()
        old = self._grad[1]
        self._grad[1] = g
        recs = [Stats() for _ in self._parts]
        calls = [lambda h=h, lo=lo, st=st: self._lib.dxm_integrate_rows(h, g.ctypes.data + lo * g.strides[0], float(dt), _ptr(flux), _ptr(tangent),
                                                                        rows.ctypes.data + lo * 8, C.byref(st))
                 for (h, lo, hi, _dev), st in zip(self._parts, recs)]
        self._warm = True
        isv = self._after_rows(self._finish_blocks(self._run_over_one_array(calls, g), recs))
        if isinstance(old, np.ndarray) and old is not g and old is not self._grad[0]:
            _reaper.drop(old)
        del old
        return isv

    def integrate_displacement_rows(self, mesh, u, rows, flux, tangent, dt=0):
        """:meth:`integrate_rows` with the gradient evaluated on the device from the nodal vector ``u``
        (:meth:`integrate_displacement`): ``mesh`` holds the cells of this material's map only (its connectivity restricted to
        them; coordinates and displacement vector of the whole mesh)."""
        h = self._require()
        u = _as_c(u).reshape(-1)
        if u.size != mesh.displacement_size:
            raise ValueError(f"u must have {mesh.displacement_size} entries, got {u.size}")
        self._check_rows(rows, flux, tangent)
        self._retire_final_views()
        st = Stats()
        rc = self._lib.dxm_integrate_displacement_rows(h, mesh._handle, _ptr(u), float(dt), _ptr(flux), _ptr(tangent), rows.ctypes.data, C.byref(st))
        return self._after_rows(self._finish_blocks([rc], [st]))

    def integrate_displacement(self, mesh, u, dt=0):
        """Same as :meth:`integrate`, with the gradient evaluated on the device from the nodal
        displacement vector ``u`` (``mesh``: :class:`~dolfinx_materials_amd.gradient.Hex8Mesh`,
        :class:`~dolfinx_materials_amd.gradient.Tet4Mesh` or :class:`~dolfinx_materials_amd.gradient.SimplexMesh`):
        only ``u`` crosses PCIe on the way in (the step before the path,
        ``quadrature_function.py:45-51``)."""
        self._require()   # the mesh lives on one GPU
        self._refuse_with_row_deliveries("integrate_displacement")
        nf, ng = self._info.n_flux, self._info.n_grad
        u = _as_c(u).reshape(-1)
        if u.size != mesh.displacement_size:
            raise ValueError(f"u must have {mesh.displacement_size} entries, got {u.size}")
        eager = not self.lazy_isv
        self._ensure_outputs(isv=eager)
        self._retire_final_views()
        flux = self._next_flux_buffer()
        rc = self._integrate_blocks(self._lib.dxm_integrate_displacement, mesh._handle, u, float(dt), flux, self._out_isv if eager else None)
        if rc > 0:
            warnings.warn(f"local Newton did not converge at {rc} quadrature points", RuntimeWarning)
        self._flux[1] = flux
        self._serial += 1
        return flux, (self._out_isv if eager else LazyISV(self, (self._n, self._info.n_isv_total))), self._out_ct

    def _host_mirrors_left_behind(self):
        """The device-pointer forms produce a final state whose gradient and flux the host never sees: the s1 mirrors of the
        state dictionaries become all-NaN placeholders (no memory: a broadcast view) -- "unknown", loudly, instead of the
        arrays of whatever host-buffer call came last.  ``advance`` carries them into s0 like any other mirror; the C side
        holds no device copy for such a state either (``dxm_io_held`` is 0 after ``dxm_advance``)."""
        if getattr(self, "_grad", None) is None:
            return
        self._retire_final_views()
        self._grad[1] = np.broadcast_to(np.nan, (self._n, int(self._info.n_grad)))
        self._flux[1] = np.broadcast_to(np.nan, (self._n, int(self._info.n_flux)))
        self._serial += 1

    def integrate_device(self, grad_ptr, flux_ptr, ct_ptr, stream=0, dt=0.0):
        """Device-pointer form: asynchronous launch on ``stream`` (a ``hipStream_t`` value, e.g.
        ``torch.cuda.current_stream().cuda_stream``); the three arguments are device addresses
        of ``(N,ng)``, ``(N,nf)`` and ``(N,nf*ng)`` fp64 arrays on this material's device."""
        self._host_mirrors_left_behind()
        self._chk(
            self._lib.dxm_integrate_device(
                self._require(), int(grad_ptr), float(dt), int(flux_ptr), int(ct_ptr), int(stream) or None
            )
        )

    def integrate_displacement_device(self, mesh, u_ptr, flux_ptr, ct_ptr, stream=0, dt=0.0):
        """Device-resident form of :meth:`integrate_displacement`: ``u_ptr`` is the device address of
        the displacement vector (``mesh.displacement_size`` doubles), ``flux_ptr`` / ``ct_ptr`` device arrays as for
        :meth:`integrate_device`; asynchronous on ``stream``.  For hex8 meshes with 8 Gauss points per
        cell, tet4 meshes and Lagrange simplex meshes the gradient is evaluated inside the update kernel."""
        self._host_mirrors_left_behind()
        self._chk(self._lib.dxm_integrate_displacement_device(
            self._require(), mesh._handle, int(u_ptr), float(dt), int(flux_ptr), int(ct_ptr), int(stream) or None))

    @property
    def launch_generation(self):
        """``dxm_launch_generation``: a HIP graph that captured ``integrate_device`` /
        ``integrate_displacement_device`` of this material may be replayed only while this value equals the
        one read at capture time (``data_manager.update()`` swaps the state buffers, parameter / option
        changes re-configure the launch)."""
        return int(self._lib.dxm_launch_generation(self._require()))

    def notify_replay(self):
        """Call after replaying a HIP graph that contains a launch of this material (the replay rewrote s1,
        flux, tangent and the stats without the library seeing it): ``dxm_notify_replay``."""
        self._host_mirrors_left_behind()
        self._chk(self._lib.dxm_notify_replay(self._require()))

    def set_option(self, name, value):
        """Per-handle options of ``include/dxmat.h`` (``"pipeline"``, ``"split_streams"``, ``"packed_transfer"``, ``"packed_min_points"``,
        ``"register_input"``, ``"stage_ahead"``, ``"keep_initial_io"``, ``"pageable_dma"``, ``"host_threads"``, ``"max_chunks"``,
        ``"fused_gradient"``, ``"blocks_per_cu"``, ``"verbose"``; process-wide:
        ``"query_foreign_pointers"``)."""
        for h in self._handles():
            self._chk(self._lib.dxm_set_option(h, name.encode(), float(value)))

    def bind_outputs(self, flux=None, tangent=None):
        """Deliver ``integrate`` results straight into caller-owned arrays -- e.g. the ``x.array`` of the flux and
        ``jacobian_flatten`` quadrature Functions, which is what ``QuadratureMap.update`` scatters into
        (``quadrature_map.py:331-334``, ``utils.py:136-143``; an identity scatter when the map covers all cells).
        The arrays are page-locked in place (``dxm_host_register``) and become the arrays ``integrate``
        returns; call after ``set_data_manager``.  ``None`` keeps the material-owned buffer."""
        self._handles()
        nf, ng = self._info.n_flux, self._info.n_grad
        want = {"flux": (flux, self._n * nf), "tangent": (tangent, int(np.prod(self._ct_shape)))}
        for key, (arr, size) in want.items():
            if arr is None:
                continue
            if not (isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.flags.c_contiguous and arr.size == size):
                raise ValueError(f"{key} must be a C-contiguous float64 array with {size} entries")
            self._unbind(key)
            if arr.nbytes:
                self._chk(self._lib.dxm_host_register(_ptr(arr), arr.nbytes))
            self._bound[key] = arr
        if "flux" in self._bound:
            self._flux_buf = [self._bound["flux"].reshape(self._n, nf)]
            self.set_option("keep_initial_io", 1)   # the flux of an accepted state stays on the device (_advance)
        if "tangent" in self._bound:
            self._out_ct = self._bound["tangent"].reshape(self._ct_shape)

    def bind_inputs(self, gradient=None):
        """Page-lock a caller-owned gradient array in place (``dxm_host_register``) -- e.g. the ``x.array`` of the
        gradient's quadrature Function, which ``Expression.eval(..., values=)`` fills per update: ``integrate`` on
        (a view of) that memory then uploads by DMA instead of staging the array through the library's page-locked
        ring chunk by chunk.  The array must stay alive until ``close()`` / ``set_data_manager``."""
        self._handles()
        if gradient is None:
            return
        size = self._n * self._info.n_grad
        if not (isinstance(gradient, np.ndarray) and gradient.dtype == np.float64 and gradient.flags.c_contiguous and gradient.size == size):
            raise ValueError(f"gradient must be a C-contiguous float64 array with {size} entries")
        self._unbind("gradient")
        if gradient.nbytes:
            self._chk(self._lib.dxm_host_register(_ptr(gradient), gradient.nbytes))
        self._bound["gradient"] = gradient
        self.set_option("keep_initial_io", 1)

    #: :meth:`bind_state_outputs` takes ``rows=True`` (the ISV Functions of a map over a subset of the cells)
    supports_row_state_outputs = True

    def bind_state_outputs(self, arrays, deliver=False, rows=False):
        """Page-lock in place the caller-owned arrays that receive internal state variables -- the ``x.array`` of the ISV
        quadrature Functions (``read_final_state(name, out)`` into such memory is one DMA transfer instead of a staged copy).
        ``arrays``: name -> C-contiguous fp64 array of ``N * dim`` entries.  ``deliver=True``: every host-buffer ``integrate``
        writes these fields of the final state into the arrays inside its transfer pipeline (``dxm_bind_isv_output``) -- what
        ``QuadratureMap.update`` does after each ``integrate`` (``quadrature_map.py:332, :343-348``) without a second pass;
        :attr:`delivers_state_outputs` then names the fields.

        ``rows=True`` (with ``deliver=True``): the arrays are the Functions over ALL cells of a map over a subset -- ``M * dim``
        entries, ``M >= N`` -- and :meth:`integrate_rows` / :meth:`integrate_displacement_rows` put the fields of point ``i`` into their
        row ``rows[i]``, like stress and tangent block (``_update_vals(isv, values, cells)``, ``utils.py:136-143``, done by the
        threads that rebuild the blocks).  Such a binding serves the rows forms only: ``integrate`` refuses while it is in place."""
        self._handles()
        if rows and not deliver:
            raise ValueError("rows=True describes where deliveries go: pass deliver=True")
        for name, arr in arrays.items():
            if name not in self.internal_state_variables:
                raise ValueError(f"unknown internal state variable {name!r}")
            dim = max(1, self.internal_state_variables[name])
            size = self._n * dim
            ok = isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.flags.c_contiguous
            if not (ok and (arr.size == size if not rows else (arr.size % dim == 0 and arr.size >= size))):
                raise ValueError(f"{name} must be a C-contiguous float64 array with {size} entries" + (" or more (whole rows)" if rows else ""))
            key = "isv:" + name
            self._unbind(key)
            if arr.nbytes:
                self._chk(self._lib.dxm_host_register(_ptr(arr), arr.nbytes))
            self._bound[key] = arr
            if deliver:
                f = self.internal_state_variable_names.index(name)
                if rows:   # every block gets the BASE: its points find their rows through the index of the call
                    for h, lo, hi, _dev in self._parts:
                        self._chk(self._lib.dxm_bind_isv_output(h, f, _ptr(arr) if hi > lo else None))
                    self._delivers_rows = True
                else:
                    block_rows = arr.reshape(self._n, dim)
                    for h, lo, hi, ptrs in self._blocks(block_rows):
                        self._chk(self._lib.dxm_bind_isv_output(h, f, ptrs[0] if hi > lo else None))
                self._delivered.add(name)

    @property
    def delivers_state_outputs(self):
        """Names of the internal state variables that every host-buffer ``integrate`` writes into bound arrays
        (``bind_state_outputs(..., deliver=True)``)."""
        return frozenset(self.__dict__.get("_delivered", ()))

    def scatter_rows(self, dst, rows, src):
        """``dst[rows] = src`` for ``(*, w)`` fp64 arrays on several threads (``dxm_host_scatter_rows``): what a map over a
        subset of the cells does with flux, tangent and state per update (``utils.py:136-143``; numpy's fancy assignment is
        one core, 1 s per 1e7 tangent blocks).  ``rows``: C-contiguous int64, each row once."""
        src = np.ascontiguousarray(src, dtype=np.float64)
        if not (dst.dtype == np.float64 and dst.flags.c_contiguous and dst.ndim == 2 and src.ndim == 2 and src.shape[1] == dst.shape[1]
                and rows.dtype == np.int64 and rows.flags.c_contiguous and len(rows) == len(src)):
            dst[rows] = src
            return
        self._chk(self._lib.dxm_host_scatter_rows(_ptr(dst), _ptr(src), rows.ctypes.data, len(rows), dst.shape[1], 16))

    def gather_rows(self, src, rows):
        """``src[rows]`` the same way (``quadrature_map.py:271``: ``_get_vals(field)[self.dofs]``)."""
        if not (src.dtype == np.float64 and src.flags.c_contiguous and src.ndim == 2 and rows.dtype == np.int64 and rows.flags.c_contiguous):
            return src[rows]
        out = np.empty((len(rows), src.shape[1]))
        self._chk(self._lib.dxm_host_gather_rows(_ptr(out), _ptr(src), rows.ctypes.data, len(rows), src.shape[1], 16))
        return out

    def pinned_array(self, shape):
        """A zero-initialised fp64 array in page-locked host memory that owns its block (``_lib.PinnedArray``)."""
        a = _lib.PinnedArray(shape).array
        a[...] = 0.0
        return a

    def read_final_state(self, name, out):
        """Field ``name`` of the final state s1 into the caller's C-contiguous ``(N, dim)`` fp64 array -- internal
        state variables come straight from the device into it (``dxm_get_state``); the flux is the array the last
        ``integrate`` delivered (nothing to do when ``out`` is that memory, i.e. a bound Function)."""
        self._handles()
        if name not in self.variables:
            raise ValueError(f"unknown field {name!r}")
        dim = max(1, self.variables[name])
        if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.flags.c_contiguous and out.size == self._n * dim):
            raise ValueError(f"out must be a C-contiguous float64 array with {self._n * dim} entries")
        if name in (self._fname, self._gname):
            src = (self._flux if name == self._fname else self._grad)[1]
            if src.ctypes.data != out.ctypes.data:
                out.reshape(src.shape)[...] = src
        elif self._n:
            f = self.internal_state_variable_names.index(name)
            rows = out.reshape(self._n, dim)
            self._run([lambda h=h, ptr=ptrs[0]: self._chk(self._lib.dxm_get_state(h, S1, f, ptr))
                       for h, lo, hi, ptrs in self._blocks(rows) if hi > lo])
        return out

    def _unbind(self, key=None):
        for k in ([key] if key else list(self._bound)):
            arr = self._bound.pop(k, None)
            if arr is None:
                continue
            if k.startswith("isv:") and k[4:] in self.__dict__.get("_delivered", ()):   # stop the deliveries before the page-lock goes
                f = self.internal_state_variable_names.index(k[4:])
                for h, *_ in getattr(self, "_parts", []):
                    self._lib.dxm_bind_isv_output(h, f, None)
                self._delivered.discard(k[4:])
                if not self._delivered:
                    self._delivers_rows = False
            if arr.nbytes:
                self._lib.dxm_host_unregister(_ptr(arr))
            # back to the material's own buffers (allocated when next needed)
            if k == "flux":
                self._flux_buf = []
            elif k == "tangent":
                self._out_ct = None

    def isv_device(self, which, isv_ptr, stream=0):
        self._chk(self._lib.dxm_isv_device(self._require(), which, int(isv_ptr), int(stream) or None))

    def stats(self):
        """Wait for the last integrate and return its per-batch status."""
        st = Stats()
        rc = self._chk(self._lib.dxm_get_stats(self._require(), C.byref(st)))
        self.last_stats = st.as_dict()
        return rc, self.last_stats
