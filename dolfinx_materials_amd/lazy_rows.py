"""Arrays that live on the device until somebody looks, and the rule by which the state dictionaries hand them out.

Split out of ``hip_material.py`` (round 5) so that the one piece of the Python layer whose correctness is about object LIFETIMES can
be read on its own.  Three kinds, all ``numpy``-like (``__array__``, ufuncs, indexing, attribute fall-through):

* :class:`LazyISV` -- the ``isv`` array ``integrate`` returns: a VIEW of the material's current final state s1, like the arrays
  ``generic.Material.integrate`` returns (``generic.py:185-189``);
* :class:`LazyInitialRows` / :class:`LazyFinalRows` -- gradient / flux of s0 / s1 that exist on the device only.  The material keeps
  ONE mirror per field and never gives it away; ``get_initial_state_dict()`` / ``get_final_state_dict()`` hand out *views*
  (``mirror._view()``), which the mirror knows through weak references.  When the state the mirror stands for is replaced and a view
  is still alive, the mirror downloads the rows once and all its views keep showing them -- the value semantics of the reference's
  dictionaries, which hold copies (``generic.py:212-213, :237-240, :265-277``).  No reference counting: the only question asked is
  whether a view object still exists (an interpreter without prompt finalisation answers "yes" for longer: one download more, same
  values).  Exercised without a GPU by ``tests/test_protocol_fuzz_cpu.py``.

:class:`_Reaper` frees large host arrays off the calling thread.
"""
from __future__ import annotations

import weakref

import numpy as np


class _Reaper:
    """Drops the last reference to large arrays on a helper thread.

    ``integrate`` keeps the caller's gradient array as the gradient of the final state (no copy), so the array of the
    PREVIOUS call dies inside the next ``integrate`` -- and ``QuadratureMap.update`` builds a new one per call
    (``quadrature_map.py:304-313``): freeing 480 MB (1e7 points) is a 17 ms ``munmap`` (28 ms without transparent huge
    pages) on the calling thread, more than half of what the whole PCIe-bound call takes
    (``profiles/archive/r03_hostpath_fresh_array.md``).  The array is handed to this thread when the call that replaced it
    returns: the free then runs beside whatever the caller does next (a ``munmap`` running beside the chunk pipeline
    itself slows that by 10-15 ms: it was tried)."""

    def __init__(self):
        self._q = None

    def drop(self, obj):
        if obj is None or getattr(obj, "nbytes", 0) < (8 << 20):
            return
        if self._q is None:
            import queue
            import threading

            self._q = queue.SimpleQueue()
            threading.Thread(target=self._run, name="dxm-array-reaper", daemon=True).start()
        self._q.put(obj)

    def _run(self):
        while True:
            item = self._q.get()
            del item   # the munmap happens here


_reaper = _Reaper()


class LazyISV(np.lib.mixins.NDArrayOperatorsMixin):
    """The ``isv`` array of ``integrate`` -- ``(N, sum isv)``, the ``_hcat_mixed`` of ``jaxmat.py:227-229`` --
    fetched from the device on first use.  Internal state variables are consumed when an increment has
    converged (``QuadratureMap.advance``, ``quadrature_map.py:350-360``), not in every Newton iteration, and
    they are 56 of the 392 B/point the host-buffer form would otherwise bring back over PCIe per call.
    Anything that looks at the values (``np.isnan(isv)``, ``isv[:, a:b]`` as in ``quadrature_map.py:323, :343-348``,
    ``np.asarray(isv)``) triggers one download.  Like the arrays ``generic.Material.integrate`` returns
    (``generic.py:185-189``) it is a VIEW of the material's current final state ``s1``: looked at after a later
    ``integrate`` it shows that call's values."""

    def __init__(self, material, shape):
        self._m, self.shape = material, tuple(shape)
        self._seen = -1           # serial of the integrate call whose state was last downloaded through this object
        self.dtype = np.dtype(np.float64)
        self.ndim = 2

    def _serial(self):
        return self._m._serial

    def _download(self):
        return self._m._fetch_isv()

    _frozen = False

    def _get(self):
        if self._frozen:
            return self._value
        if self._seen != self._serial():
            self._value = self._download()
            self._seen = self._serial()
        return self._value

    def _freeze(self):
        """Stop following the material: keep showing the state this view stands for now (downloads it if nobody has looked yet)."""
        if not self._frozen:
            self._value = self._get()
            self._frozen = True

    @property
    def fetched(self):
        return self._frozen or self._seen == self._serial()

    def __array__(self, dtype=None, copy=None):
        a = self._get()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):   # isv - other, np.isnan(isv), ...
        inputs = tuple(x._get() if isinstance(x, LazyISV) else x for x in inputs)
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __getitem__(self, idx):
        return self._get()[idx]

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        return iter(self._get())

    def __getattr__(self, name):   # .any(), .copy(), .reshape(...), .T ...: whatever an ndarray offers
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._get(), name)

    def __repr__(self):
        return f"{type(self).__name__}(shape={self.shape}, fetched={self.fetched})"


class LazyInitialRows(LazyISV):
    """Gradient (``kind`` 0) or flux (1) of the INITIAL state s0 -- ``get_initial_state_dict()["Strain"]``,
    ``generic.py:194-198`` -- kept on the device by ``dxm_advance`` (option ``keep_initial_io``) and downloaded when
    first looked at.  Used when the host array that held the accepted state is a bound Function that the next update
    overwrites: accepting an increment then costs a pointer swap instead of a 480 MB host copy per array (1e7 points).

    It stands for ONE initial state.  The material keeps one such object per field as its s0 mirror and never gives it away:
    the state dictionaries hand out *views* of it (:meth:`_view`; same class, ``_parent`` set), which the mirror knows through
    weak references.  When that state is replaced (``advance`` with a new state, ``set_initial_state_dict``) and a view is still
    alive -- in a local, a container, a closure: whoever holds it -- the mirror downloads its rows at that moment and all its
    views keep showing them, like the arrays the reference hands out, which are copies (``generic.py:212-213, :237-240, :265-277``);
    with no view alive nothing is downloaded.  No reference counting is involved: the only question asked is whether a view
    object still exists (an interpreter without prompt finalisation answers "yes" for longer: one download more, same values)."""

    _which = 0

    def __init__(self, material, shape, kind, parent=None):
        super().__init__(material, shape)
        self._kind = kind
        self._parent = parent
        self._views = [] if parent is None else None     # weak references to the views handed out (identity only: `==` on a view is an array operation)

    def _view(self):
        """A new view of this mirror for a caller (what the state dictionaries contain)."""
        v = type(self)(self._m, self.shape, self._kind, parent=self)
        self._views = [r for r in self._views if r() is not None]
        self._views.append(weakref.ref(v))
        return v

    def _handed_out(self):
        """Whether a view given to a caller still exists."""
        return any(r() is not None for r in self._views)

    def _get(self):
        return super()._get() if self._parent is None else self._parent._get()

    def _freeze(self):
        if self._parent is None:
            super()._freeze()
        else:
            self._parent._freeze()

    @property
    def fetched(self):
        top = self if self._parent is None else self._parent
        return top._frozen or top._seen == top._serial()

    def _serial(self):
        return self._m._serial0

    def _download(self):
        # read-only: every view of this mirror shows THIS array (and advance() may install it as the next s0 mirror); the
        # reference's dictionaries hold private copies (generic.py:265-277), so a caller who writes into one changes nothing but
        # the copy -- here such a write is refused instead of silently changing what every other view shows
        rows = self._m._fetch_io_rows(self._which, self._kind)
        rows.setflags(write=False)
        return rows


class LazyFinalRows(LazyInitialRows):
    """The flux of the FINAL state s1 after :meth:`HIPMaterial.integrate_rows`, whose results went to scattered rows of the
    caller's arrays: the contiguous ``(N, nf)`` array exists on the device only and is downloaded when somebody asks
    (``get_final_state_dict()["Stress"]``).  Same hand-out rule as :class:`LazyInitialRows`: the material keeps the mirror, the
    dictionaries hold views of it, and a view that is still alive when the next update (or ``revert``) replaces s1 keeps the rows
    of ITS state."""

    _which = 1

    def _serial(self):
        return self._m._serial
