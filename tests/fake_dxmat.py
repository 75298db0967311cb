"""A test double of ``libdxmat.so`` for the CPU suite: the HANDLE semantics of ``csrc/dxmat.hip`` (state buffers and the s1 alias,
which device copies of gradient / flux a handle holds and when it drops them, what the host-buffer calls deliver) restated in
Python over numpy arrays, with the arithmetic done by the C oracle.

TEST INFRASTRUCTURE ONLY.  It exists so that the Python layer above the C ABI -- ``hip_material.HIPMaterial``: the s0 / s1 mirrors,
the views the state dictionaries hand out, the bound-array paths, ``quadrature_map.AcceleratedUpdate`` on top of them -- runs its
protocol fuzz without a GPU (``tests/test_protocol_fuzz_cpu.py``); the product never sees it (``_lib.load`` is monkeypatched by the
test).  What it restates, with the lines it follows:

    launch / host-buffer call   csrc/dxmat.hip launch_range (io1_valid = 0), launch (s1_alias = false), run_and_download (io1_valid = 3)
    dxm_advance                 swap of the state buffers, s1_alias = true, copies of the accepted state become those of s0 (option
                                keep_initial_io) or are dropped, io1_valid = 0
    dxm_revert                  s1_alias = true, io1_valid = 0
    dxm_set_state               materialize_s1 first (an aliased s1 gets its own storage and brings no copies along)
    dxm_get_io / dxm_io_held    io_mask: s1 shows the copies of s0 while it is served from it
    dxm_bind_isv_output         fields of the final state into bound rows inside every host-buffer call (the rows forms: into row rows[i]
                                of the bound base)

Entry points that are pure host code (law table, threaded copies, row scatter / gather, index range) go to the real library, which
loads without a GPU.  Small-strain laws, full tangent layout, one device.
"""
import ctypes as C

import numpy as np

from dolfinx_materials_amd import _lib
from oracle import oracle_c

S0, S1 = _lib.S0, _lib.S1


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, int):
        return p
    if hasattr(p, "_obj"):          # ctypes.byref(x)
        return C.addressof(p._obj)
    if isinstance(p, C.c_void_p):
        return p.value or 0
    return C.cast(p, C.c_void_p).value or 0


def _rows(p, n, dim):
    """The caller's C-contiguous (n, dim) fp64 memory at pointer ``p`` as a writable numpy view."""
    a = _addr(p)
    if not a or n * dim == 0:
        return None
    return np.ctypeslib.as_array((C.c_double * (n * dim)).from_address(a)).reshape(n, dim)


class _Handle:
    def __init__(self, law, params, n):
        self.law, self.params, self.n = law, list(params), n
        self.state = [dict(p=np.zeros(n), epsp=np.zeros((n, 6))), dict(p=np.zeros(n), epsp=np.zeros((n, 6)))]   # [s0, s1]
        self.s1_alias = False                # (a new handle's s1 has its own storage: csrc/dxmat.hip `bool s1_alias = false`)
        self.io = [dict(grad=None, flux=None), dict(grad=None, flux=None)]   # device copies of (gradient, flux) for s0 / s1
        self.io_valid = [0, 0]
        self.keep_initial_io = 0
        self.isv_out = {}
        self.stats = dict(n_points=n, n_plastic=0, n_not_converged=0, n_nan=0, max_local_iters=0)
        self.epoch, self.parity = 0, 0       # dxm_launch_generation = epoch << 1 | parity (csrc/dxmat.hip)
        self.launched = False

    @property
    def generation(self):
        return (self.epoch << 1) | self.parity

    def state_of(self, which):
        return self.state[1 if (which == S1 and not self.s1_alias) else 0]

    def materialize_s1(self):
        if self.s1_alias:
            self.state[1] = {k: v.copy() for k, v in self.state[0].items()}
            self.s1_alias = False

    def io_mask(self, which):
        return self.io_valid[0] if (which == S0 or self.s1_alias) else self.io_valid[1]


class FakeDxmat:
    """Stands where ``_lib.load()`` returns the ``ctypes.CDLL`` of libdxmat.so."""

    def __init__(self, real):
        self._real = real
        self._handles = {}
        self._next = 1000
        self._error = b""
        self._pinned = {}
        self.downloads = []          # (which, kind) of every dxm_get_io: what a test counts

    def __getattr__(self, name):    # host-only entry points: law table, dxm_host_copy / scatter / gather / index_range ...
        if name.startswith("dxm_"):
            return getattr(self._real, name)
        raise AttributeError(name)

    def _fail(self, rc, msg):
        self._error = msg.encode()
        return rc

    def dxm_last_error(self):
        return self._error

    def dxm_abi_version(self):
        return self._real.dxm_abi_version()

    # ---- life cycle ---------------------------------------------------------------------------------------------------------
    def dxm_create(self, law, params, nparams, npoints, device):
        if law not in (_lib.LAW_ELASTIC_ISO, _lib.LAW_J2_LINEAR, _lib.LAW_J2_VOCE):
            self._fail(-1, "fake library: small-strain laws only")
            return None
        prm = [params[i] for i in range(nparams)]
        self._next += 8
        self._handles[self._next] = _Handle(law, prm, int(npoints))
        return self._next

    def _h(self, h):
        return self._handles[_addr(h)]

    def dxm_destroy(self, h):
        self._handles.pop(_addr(h), None)
        return 0

    def dxm_set_params(self, h, params, nparams):
        self._h(h).params = [params[i] for i in range(nparams)]
        self._h(h).epoch += 1                # dxm_set_params: ++m->epoch
        return 0

    def dxm_set_newton(self, h, maxit, rtol):
        self._h(h).epoch += 1
        return 0

    def dxm_set_tangent_layout(self, h, layout):
        if layout != 0:
            return self._fail(-1, "fake library: full tangent layout only")
        self._h(h).epoch += 1
        return 0

    def dxm_tangent_size(self, h):
        return 36

    def dxm_kernel_name(self, h):
        return b"fake (tests/fake_dxmat.py)"

    def dxm_launch_generation(self, h):
        return self._h(h).generation

    def dxm_notify_replay(self, h):
        return 0

    def dxm_set_option(self, h, name, value):
        m = self._h(h)
        if name == b"keep_initial_io":
            m.keep_initial_io = int(value != 0)
        m.epoch += 1                         # dxm_set_option: ++m->epoch
        return 0

    # ---- page-locked memory: ordinary memory here ------------------------------------------------------------------------------
    def dxm_host_alloc(self, nbytes):
        buf = (C.c_char * max(1, int(nbytes)))()
        self._pinned[C.addressof(buf)] = buf
        return C.addressof(buf)

    def dxm_host_free(self, p):
        self._pinned.pop(_addr(p), None)
        return 0

    def dxm_host_register(self, p, nbytes):
        return 0

    def dxm_host_unregister(self, p):
        return 0

    # ---- state ----------------------------------------------------------------------------------------------------------------
    FIELDS = (("p", 1), ("epsp", 6))

    def dxm_set_state(self, h, which, field, host):
        m = self._h(h)
        if m.law == _lib.LAW_ELASTIC_ISO or not 0 <= field < 2:
            return self._fail(-1, "state field out of range")
        m.materialize_s1()
        name, dim = self.FIELDS[field]
        m.state_of(which)[name][...] = _rows(host, m.n, dim).reshape(m.state_of(which)[name].shape)
        return 0

    def dxm_get_state(self, h, which, field, host):
        m = self._h(h)
        if m.law == _lib.LAW_ELASTIC_ISO or not 0 <= field < 2:
            return self._fail(-1, "state field out of range")
        name, dim = self.FIELDS[field]
        if m.n:
            _rows(host, m.n, dim)[...] = m.state_of(which)[name].reshape(m.n, dim)
        return 0

    def dxm_isv_host(self, h, which, host):
        m = self._h(h)
        if m.law == _lib.LAW_ELASTIC_ISO or m.n == 0:
            return 0
        out = _rows(host, m.n, 7)
        st = m.state_of(which)
        out[:, 0], out[:, 1:] = st["p"], st["epsp"]
        return 0

    def dxm_advance(self, h):
        m = self._h(h)
        if not m.s1_alias:
            m.state[0], m.state[1] = m.state[1], m.state[0]
            m.s1_alias = True
            m.parity ^= 1                    # launches read / write the other buffer now
            if m.keep_initial_io and m.io_valid[1]:
                for bit, key in ((1, "grad"), (2, "flux")):
                    if m.io_valid[1] & bit:
                        m.io[0][key], m.io[1][key] = m.io[1][key], m.io[0][key]
            m.io_valid[0] = m.io_valid[1] if m.keep_initial_io else 0
            m.io_valid[1] = 0
        return 0

    def dxm_revert(self, h):
        m = self._h(h)
        m.s1_alias = True
        m.io_valid[1] = 0
        return 0

    def dxm_io_held(self, h, which):
        return self._h(h).io_mask(which)

    def dxm_get_io(self, h, which, kind, host):
        m = self._h(h)
        if not m.io_mask(which) & (1 << kind):
            return self._fail(-1, f"the {'flux' if kind else 'gradient'} of that state is not held on the device")
        first = which == S0 or m.s1_alias
        self.downloads.append((which, kind))
        if m.n:
            _rows(host, m.n, 6)[...] = m.io[0 if first else 1]["flux" if kind else "grad"]
        return 0

    def dxm_bind_isv_output(self, h, field, host):
        m = self._h(h)
        if not 0 <= field < 2 or m.law == _lib.LAW_ELASTIC_ISO:
            return self._fail(-1, "state field out of range")
        m.isv_out[field] = _addr(host) or None
        return 0

    # ---- the hot call, host-buffer forms ------------------------------------------------------------------------------------------
    def _update(self, m, grad):
        E, nu = m.params[0], m.params[1]
        m.io_valid[1] = 0                    # launch_range: s1 is being rewritten
        s0 = m.state[0]
        if m.law == _lib.LAW_ELASTIC_ISO:
            sig, ct = oracle_c.elastic_iso(grad, E, nu)
            r = dict(sig=sig, Ct=ct, p=s0["p"], epsp=s0["epsp"], n_plastic=0, n_not_converged=0)
        else:
            kind = 0 if m.law == _lib.LAW_J2_LINEAR else 1
            r = oracle_c.j2(grad, s0["epsp"], s0["p"], E, nu, kind, *m.params[2:])
        m.state[1] = dict(p=r["p"].copy(), epsp=r["epsp"].copy())
        m.s1_alias = False
        m.launched = True
        nan = int(np.isnan(r["sig"]).any(axis=1).sum())
        m.stats = dict(n_points=m.n, n_plastic=int(r["n_plastic"]), n_not_converged=int(r["n_not_converged"]), n_nan=nan, max_local_iters=0)
        return r

    def _finish(self, m, grad, r, stats, idx=None):
        for field, addr in m.isv_out.items():
            if addr:
                name, dim = self.FIELDS[field]
                if idx is None:
                    _rows(addr, m.n, dim)[...] = m.state[1][name].reshape(m.n, dim)
                else:   # the rows forms: the bound pointer is the base of the array over all rows (run_and_download: isv_rows)
                    _rows(addr, int(idx.max()) + 1, dim)[idx] = m.state[1][name].reshape(m.n, dim)
        m.io[1] = dict(grad=np.array(grad), flux=r["sig"].copy())
        m.io_valid[1] = 3
        self._fill_stats(m, stats)
        return min(m.stats["n_not_converged"], 0x7FFFFFFF)

    def _fill_stats(self, m, stats):
        a = _addr(stats)
        if a:
            st = _lib.Stats.from_address(a)
            for k, v in m.stats.items():
                setattr(st, k, v)
            st.upload = 1

    def dxm_integrate(self, h, grad, dt, flux, isv, ct, stats):
        m = self._h(h)
        if m.n == 0:
            self._fill_stats(m, stats)
            return 0
        g = _rows(grad, m.n, 6)
        r = self._update(m, g)
        if _addr(flux):
            _rows(flux, m.n, 6)[...] = r["sig"]
        if _addr(ct):
            _rows(ct, m.n, 36)[...] = r["Ct"].reshape(m.n, 36)
        if _addr(isv) and m.law != _lib.LAW_ELASTIC_ISO:
            out = _rows(isv, m.n, 7)
            out[:, 0], out[:, 1:] = m.state[1]["p"], m.state[1]["epsp"]
        return self._finish(m, g, r, stats)

    def dxm_integrate_rows(self, h, grad, dt, flux_base, ct_base, rows, stats):
        m = self._h(h)
        if m.n == 0:
            self._fill_stats(m, stats)
            return 0
        idx = np.ctypeslib.as_array((C.c_int64 * m.n).from_address(_addr(rows)))
        g = _rows(grad, m.n, 6)
        r = self._update(m, g)
        top = int(idx.max()) + 1
        _rows(flux_base, top, 6)[idx] = r["sig"]
        _rows(ct_base, top, 36)[idx] = r["Ct"].reshape(m.n, 36)
        return self._finish(m, g, r, stats, idx)

    def dxm_get_stats(self, h, stats):
        self._fill_stats(self._h(h), stats)
        return min(self._h(h).stats["n_not_converged"], 0x7FFFFFFF)
