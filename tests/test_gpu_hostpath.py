"""The host-buffer form (what ``QuadratureMap.update`` hands over, ``quadrature_map.py:297-334``): packed
transfer of the symmetric tangent with the full block rebuilt on the host, internal state variables on
demand, results delivered into caller-owned (page-locked in place) arrays, options instead of environment
variables, and the launch-generation contract for HIP graphs."""
import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd import _lib
from dolfinx_materials_amd.hip_material import LazyISV
from dolfinx_materials_amd.jaxmat import JAXMaterial
from oracle import constitutive_np as onp

from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, j2_history

pytestmark = pytest.mark.gpu
from helpers import to_device, to_host  # noqa: E402,F401



def _j2(kind="linear", **kw):
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    hard = jm.LinearHardening(SIG0_LIN, H_LIN) if kind == "linear" else jm.VoceHardening(SIG0_V, SIGU_V, B_V)
    return JAXMaterial(jm.vonMisesIsotropicHardening(el, hard), **kw)


@pytest.mark.parametrize("n", [1, 255, 40_001, 300_001, 2_200_000])
def test_packed_tangent_transfer_is_bit_identical_to_the_full_download(n):
    """2.2e6 points = 33 chunks on two streams with 16 expansion threads behind them; 300001 and 40001 = ragged chunks
    of 32768 points; the two small sizes take the unpacked route (below option packed_min_points = 32768)."""
    a, b = _j2(), _j2()
    a.set_data_manager(n)
    b.set_data_manager(n)
    b.set_option("packed_transfer", 0)   # moves the full 36-entry block over PCIe, as in round 1
    for eps in j2_history(n, seed=5)[:3]:
        fa, ia, ca = a.integrate(eps)
        fb, ib, cb = b.integrate(eps)
        assert ca.shape == (n, 6, 6) and np.array_equal(fa, fb) and np.array_equal(ca, cb)
        assert np.array_equal(ca, ca.transpose(0, 2, 1)) and np.array_equal(np.asarray(ia), np.asarray(ib))
        assert a.last_stats == b.last_stats
        a.data_manager.update()
        b.data_manager.update()
    ref = onp.j2_update(j2_history(n, seed=5)[0], np.zeros((n, 6)), np.zeros(n), E, NU, onp.LinearHardening(SIG0_LIN, H_LIN))
    c = _j2()
    c.set_data_manager(n)
    c.set_option("host_threads", 3)
    c.set_option("max_chunks", 5)
    ct = c.integrate(j2_history(n, seed=5)[0])[2]
    safe = np.abs(ref["f_trial"]) > 1e-9 * SIG0_LIN
    assert np.abs(ct[safe] - ref["Ct"][safe]).max() <= 1e-12 * np.abs(ref["Ct"]).max()


@pytest.mark.parametrize("n,hard", [(33_001, "voce"), (300_001, "voce"), (70_000, "linear")])
def test_fefp_building_block_transfer_is_bit_identical_to_the_full_download(n, hard):
    """FeFp, host-buffer form: the 54 building blocks of the 9x9 tangent cross PCIe (432 instead of 648 B/point) and worker
    threads rebuild the block with the kernel's four-term expression; load path of tests/test_FeFp_jax.py:28-30 with
    perturbations, elastic and plastic steps, advance in between."""
    from helpers import SIG0_F, SIGU_F, B_F, fefp_path

    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    h = jm.VoceHardening(SIG0_F, SIGU_F, B_F) if hard == "voce" else jm.LinearHardening(SIG0_F, 2e3)
    a, b = JAXMaterial(jm.FeFpJ2Plasticity(el, h)), JAXMaterial(jm.FeFpJ2Plasticity(el, h))
    a.set_data_manager(n)
    b.set_data_manager(n)
    b.set_option("packed_transfer", 0)
    nplastic = 0
    for F in fefp_path(n, nsteps=6, eps=3e-2)[::2]:
        fa, ia, ca = a.integrate(F)
        fb, ib, cb = b.integrate(F)
        assert ca.shape == (n, 9, 9) and np.array_equal(fa, fb) and np.array_equal(ca, cb)
        assert np.array_equal(np.asarray(ia), np.asarray(ib)) and a.last_stats == b.last_stats
        nplastic += a.last_stats["n_plastic"]
        a.data_manager.update()
        b.data_manager.update()
    assert nplastic > 0


def test_isv_is_fetched_on_demand_and_equals_the_eager_download():
    n = 40_000
    lazy, eager = _j2("voce"), _j2("voce", lazy_isv=False)
    lazy.set_data_manager(n)
    eager.set_data_manager(n)
    h = j2_history(n, sig0=SIG0_V)
    for eps in h[:2]:
        _, il, _ = lazy.integrate(eps)
        _, ie, _ = eager.integrate(eps)
        assert isinstance(il, LazyISV) and not il.fetched and isinstance(ie, np.ndarray) and il.shape == ie.shape == (n, 7)
        assert not np.isnan(il).any() and il.fetched            # quadrature_map.py:323
        assert np.array_equal(il[:, 0:1], ie[:, 0:1]) and np.array_equal(il[:, 1:7], ie[:, 1:7])   # quadrature_map.py:343-348
        assert np.array_equal(np.asarray(il), ie) and np.abs(il - ie).max() == 0.0 and il.max() == ie.max()
        lazy.data_manager.update()
        eager.data_manager.update()
    # never looked at: the state dicts still serve the values (quadrature_map.py:355-360) ...
    _, il, _ = lazy.integrate(h[2])
    _, ie, _ = eager.integrate(h[2])
    lazy.data_manager.update()
    assert np.array_equal(lazy.get_final_state_dict()["p"][:, 0], ie[:, 0])
    # ... and, like the views generic.Material.integrate returns (generic.py:185-189), the array follows s1
    _, il_old, _ = lazy.integrate(h[3])
    first = np.asarray(il_old).copy()
    _, il_new, _ = lazy.integrate(h[3] * 0.5)   # unloads further: some points yield in reverse
    assert np.array_equal(np.asarray(il_old), np.asarray(il_new)) and not np.array_equal(first, np.asarray(il_new))


def test_bound_arrays_keep_the_initial_gradient_and_flux_on_the_device():
    """With the gradient / flux Functions bound (what the accelerated QuadratureMap does) accepting an increment copies
    nothing on the host: `dxm_advance` keeps the device copies of the last call's strain and stress as those of s0
    (option keep_initial_io, a pointer swap) and `get_initial_state_dict()` shows them through lazy views."""
    import ctypes as C

    from dolfinx_materials_amd.hip_material import LazyInitialRows

    n = 70_003
    m = _j2()
    m.set_data_manager(n)
    lib, handle = m._lib, m._handles()[0]
    flux_fn, jac_fn, grad_fn = np.zeros(n * 6), np.zeros(n * 36), np.zeros(n * 6)
    h = j2_history(n, seed=21)
    # without the option nothing is kept (and nothing is allocated for it)
    m.integrate(h[0])
    m.data_manager.update()
    assert lib.dxm_io_held(handle, 0) == 0
    assert lib.dxm_get_io(handle, 0, 1, flux_fn.ctypes.data_as(C.c_void_p)) < 0 and b"keep_initial_io" in lib.dxm_last_error()
    assert isinstance(m.get_initial_state_dict()["stress"], np.ndarray)
    m.bind_outputs(flux=flux_fn, tangent=jac_fn)
    m.bind_inputs(gradient=grad_fn)
    rows = grad_fn.reshape(n, 6)
    rows[...] = h[1]
    f1 = np.array(m.integrate(rows)[0])
    m.data_manager.update()
    assert lib.dxm_io_held(handle, 0) == 3
    s0 = m.get_initial_state_dict()
    assert isinstance(s0["stress"], LazyInitialRows) and isinstance(s0["strain"], LazyInitialRows) and not s0["stress"].fetched
    rows[...] = h[2]
    m.integrate(rows)                                    # overwrites both bound arrays
    assert not np.array_equal(flux_fn.reshape(n, 6), f1)
    assert np.array_equal(np.asarray(s0["stress"]), f1) and np.array_equal(np.asarray(s0["strain"]), h[1])
    assert s0["stress"].shape == (n, 6) and s0["strain"][5, 2] == h[1][5, 2]
    # a view that is held keeps showing ITS initial state when the next increment is accepted (the reference rebinds s0 on update,
    # generic.py:212-213: arrays handed out earlier keep their content) -- also one nobody has looked at yet
    m2 = _j2()
    m2.set_data_manager(n)
    fl2, jc2, gr2 = np.zeros(n * 6), np.zeros(n * 36), np.zeros(n * 6)
    m2.bind_outputs(flux=fl2, tangent=jc2)
    m2.bind_inputs(gradient=gr2)
    r2 = gr2.reshape(n, 6)
    r2[...] = h[0]
    fa = np.array(m2.integrate(r2)[0])
    m2.data_manager.update()
    held_unlooked = m2.get_initial_state_dict()["stress"]
    assert isinstance(held_unlooked, LazyInitialRows) and not held_unlooked.fetched
    r2[...] = h[1]
    fb = np.array(m2.integrate(r2)[0])
    m2.data_manager.update()                              # s0 <- the state of h[1]
    assert np.array_equal(np.asarray(held_unlooked), fa)  # ... the old view still shows the state of h[0]
    assert np.array_equal(np.asarray(m2.get_initial_state_dict()["stress"]), fb)
    m2.close()
    # revert: the final state shows the same arrays; a second advance without a new state keeps them
    m.data_manager.revert()
    assert np.array_equal(np.asarray(m.get_final_state_dict()["stress"]), f1)
    m.data_manager.update()
    m.data_manager.update()
    assert lib.dxm_io_held(handle, 0) == 3 and np.array_equal(np.asarray(m.get_initial_state_dict()["strain"]), h[1])
    # a state produced by a device-pointer form has no host arrays and brings no device copy along: its gradient / flux
    # mirrors say "unknown" (all NaN, no memory) instead of showing the arrays of an earlier increment, on both sides of the ABI
    import torch
    d_eps, d_flux, d_ct = to_device(h[3]), torch.empty((n, 6), dtype=torch.float64, device="cuda:0"), torch.empty((n, 36), dtype=torch.float64, device="cuda:0")
    m.integrate_device(d_eps.data_ptr(), d_flux.data_ptr(), d_ct.data_ptr())
    torch.cuda.synchronize()
    assert np.isnan(m.get_final_state_dict()["stress"]).all() and m.get_final_state_dict()["stress"].strides == (0, 0)
    assert np.array_equal(np.asarray(m.get_initial_state_dict()["stress"]), f1)          # s0 is still the accepted increment
    m.data_manager.update()
    assert lib.dxm_io_held(handle, 0) == 0
    assert lib.dxm_get_io(handle, 0, 1, flux_fn.ctypes.data_as(C.c_void_p)) < 0
    s0_after = m.get_initial_state_dict()
    assert np.isnan(s0_after["stress"]).all() and np.isnan(s0_after["strain"]).all() and s0_after["stress"].shape == (n, 6)
    assert np.isfinite(s0_after["p"]).all()                                              # the state proper is on the device, as always
    # the next host-buffer call and advance bring real mirrors back
    rows[...] = h[3]
    f3 = np.array(m.integrate(rows)[0])
    m.data_manager.update()
    assert lib.dxm_io_held(handle, 0) == 3 and np.array_equal(np.asarray(m.get_initial_state_dict()["stress"]), f3)
    # an explicit initial stress replaces the view
    m.set_initial_state_dict({"stress": np.ones((n, 6))})
    assert np.array_equal(np.asarray(m.get_initial_state_dict()["stress"]), np.ones((n, 6)))
    m.close()


def test_views_of_the_initial_state_keep_their_content_however_they_are_held():
    """A state dictionary's lazy strain / stress keeps showing the state it was taken from across later `update(); advance()`
    cycles -- held in a local, in a list, in a closure, in a dict under its id() -- like the copies the reference's dictionaries
    hold (generic.py:212-213, :237-240, :265-277).  Whether a view is still held is asked of weak references, not of reference
    counts; a dictionary nobody kept costs no download when its state is replaced."""
    import gc
    import weakref

    from dolfinx_materials_amd.hip_material import LazyInitialRows

    n = 50_001
    m = _j2()
    m.set_data_manager(n)
    flux_fn, jac_fn, grad_fn = np.zeros(n * 6), np.zeros(n * 36), np.zeros(n * 6)
    m.bind_outputs(flux=flux_fn, tangent=jac_fn)
    m.bind_inputs(gradient=grad_fn)
    rows = grad_fn.reshape(n, 6)
    h = j2_history(n, seed=33)
    downloads = []
    fetch = m._fetch_io_rows
    m._fetch_io_rows = lambda which, kind: (downloads.append((which, kind)), fetch(which, kind))[1]

    def cycle(k):
        rows[...] = h[k]
        f = np.array(m.integrate(rows)[0])
        m.data_manager.update()
        return f

    f0 = cycle(0)
    local = m.get_initial_state_dict()["stress"]                      # (i) a local
    bag = [m.get_initial_state_dict()["stress"]]                      # (ii) a list

    def closed_over():
        v = m.get_initial_state_dict()["stress"]
        return lambda: v

    inside = closed_over()                                            # (iii) a closure
    v = m.get_initial_state_dict()["strain"]
    book = {id(v): v}                                                 # (iv) known to the caller by its id() only
    del v
    gone = weakref.ref(m.get_initial_state_dict()["stress"])          # (v) a weak reference holds nothing
    gc.collect()
    assert all(isinstance(x, LazyInitialRows) and not x.fetched for x in (local, bag[0], inside(), *book.values())) and not downloads
    f1 = cycle(1)                                                     # s0 replaced: the views above settle now ...
    assert sorted(downloads) == [(0, 0), (0, 1)]                      # ... with ONE download per field, shared by its views
    f2 = cycle(2)
    assert len(downloads) == 2 and gone() is None
    for x in (local, bag[0], inside()):
        assert np.array_equal(np.asarray(x), f0) and x.fetched
    assert np.array_equal(np.asarray(next(iter(book.values()))), h[0])
    assert np.array_equal(np.asarray(m.get_initial_state_dict()["stress"]), f2) and not np.array_equal(f0, f2) and not np.array_equal(f1, f2)
    # dictionaries that were looked at and dropped, or never kept: nothing is downloaded when their state goes
    n_before = len(downloads)
    assert np.array_equal(np.asarray(m.get_initial_state_dict()["strain"]), h[2])
    m.get_initial_state_dict()
    gc.collect()
    n_looked = len(downloads)
    assert n_looked == n_before + 1
    f3 = cycle(3)
    assert len(downloads) == n_looked
    # after revert the final state shows the initial one: its dictionary entries are views of the same kind
    rows[...] = h[0]
    m.integrate(rows)
    m.data_manager.revert()
    fin = m.get_final_state_dict()["stress"]
    assert isinstance(fin, LazyInitialRows) and np.array_equal(np.asarray(fin), f3)
    # set_initial_state_dict replaces s0 as well: a held view of the state before keeps it
    keep = m.get_initial_state_dict()["strain"]
    m.set_initial_state_dict({"strain": np.full((n, 6), 2.0)})
    assert np.array_equal(np.asarray(m.get_initial_state_dict()["strain"]), np.full((n, 6), 2.0))
    rows[...] = h[1]
    m.integrate(rows)                         # overwrites the bound arrays
    m.data_manager.update()
    assert np.array_equal(np.asarray(keep), h[3]) and np.array_equal(np.asarray(fin), f3)
    # the unbound form: the material's two alternating flux buffers go out as copies
    m2 = _j2()
    m2.set_data_manager(n)
    g0 = np.array(m2.integrate(h[0])[0])
    m2.data_manager.update()
    kept_s0, kept_s1 = m2.get_initial_state_dict()["stress"], m2.get_final_state_dict()["stress"]
    for k in (1, 2, 3):
        m2.integrate(h[k])
        m2.data_manager.update()
    assert np.array_equal(kept_s0, g0) and np.array_equal(kept_s1, g0)
    m2.close()
    m.close()


@pytest.mark.parametrize("kind,n,total,devices", [("linear", 300_007, 700_000, None), ("voce", 2_200_000, 2_500_000, None), ("linear", 1, 5, None),
                                                  ("linear", 150_003, 200_000, [0, 0, 0])])
def test_integrate_rows_delivers_the_bound_state_fields_into_their_rows(kind, n, total, devices):
    """``bind_state_outputs(deliver=True, rows=True)`` + ``integrate_rows``: the internal state variables of point i land in row
    rows[i] of the ISV Functions over all cells inside the same call (the worker threads scatter them from the library's page-locked
    landing area behind the transfers), rows of other maps untouched -- what ``_update_vals(isv, values, cells)`` does per update
    (``quadrature_map.py:332, :343-348``); several chunks, one and three blocks."""
    rng = np.random.default_rng(n + 1)
    rows = np.ascontiguousarray(rng.permutation(total)[:n], dtype=np.int64)
    others = np.setdiff1d(np.arange(total), rows)
    ref_m, m = _j2(kind), _j2(kind, devices=devices)
    ref_m.set_data_manager(n)
    m.set_data_manager(n)
    flux_fn, jac_fn = np.full(total * 6, -7.0), np.full((total, 36), -7.0)
    fields = {"p": np.full(total, -7.0), "epsp": np.full(total * 6, -7.0)}
    m.bind_state_outputs(fields, deliver=True, rows=True)
    assert m.delivers_state_outputs == {"p", "epsp"}
    h = j2_history(n, seed=14, sig0=SIG0_V if kind == "voce" else SIG0_LIN)
    for k, eps in enumerate(h[:3]):
        f0, i0, c0 = ref_m.integrate(eps)
        i0 = np.asarray(i0)
        isv = m.integrate_rows(eps, rows, flux_fn, jac_fn)
        assert np.array_equal(flux_fn.reshape(total, 6)[rows], f0) and np.array_equal(jac_fn[rows], c0.reshape(n, 36)), k
        assert np.array_equal(fields["p"][rows], i0[:, 0]) and np.array_equal(fields["epsp"].reshape(total, 6)[rows], i0[:, 1:]), k
        assert (fields["p"][others] == -7.0).all() and (fields["epsp"].reshape(total, 6)[others] == -7.0).all()
        assert np.array_equal(np.asarray(isv), i0) and m.last_stats == ref_m.last_stats
        ref_m.data_manager.update()
        m.data_manager.update()
    with pytest.raises(_lib.DxmError, match="ROWS"):
        m.integrate(h[3])
    m.close()
    ref_m.close()


@pytest.mark.parametrize("kind,n,total,devices", [("linear", 300_007, 700_000, None), ("voce", 70_001, 70_001, None), ("linear", 1, 5, None),
                                                  ("linear", 150_003, 200_000, [0, 0, 0])])
def test_integrate_rows_delivers_every_point_into_its_row(kind, n, total, devices):
    """`dxm_integrate_rows` (a QuadratureMap over a subset of the cells): stress and tangent block of point i in row rows[i] of
    arrays over all cells, bit-identical to `integrate` followed by the fancy assignment of `utils.py:136-143`; rows of other
    maps untouched; the contiguous flux of the state dictionaries served from the device."""
    from dolfinx_materials_amd.hip_material import LazyFinalRows, LazyInitialRows

    rng = np.random.default_rng(n)
    rows = np.ascontiguousarray(rng.permutation(total)[:n], dtype=np.int64)
    ref_m, m = _j2(kind), _j2(kind, devices=devices)
    ref_m.set_data_manager(n)
    m.set_data_manager(n)
    assert m.supports_row_outputs
    flux_fn, jac_fn = np.full(total * 6, -7.0), np.full((total, 36), -7.0)     # flat and 2-D both accepted
    want_f, want_c = flux_fn.reshape(total, 6).copy(), jac_fn.copy()
    h = j2_history(n, seed=4, sig0=SIG0_V if kind == "voce" else SIG0_LIN)
    for k, eps in enumerate(h[:3]):
        f0, i0, c0 = ref_m.integrate(eps)
        isv = m.integrate_rows(eps, rows, flux_fn, jac_fn)
        want_f[rows], want_c[rows] = f0, c0.reshape(n, 36)
        assert np.array_equal(flux_fn.reshape(total, 6), want_f) and np.array_equal(jac_fn, want_c), k
        assert np.array_equal(np.asarray(isv), np.asarray(i0)) and m.last_stats == ref_m.last_stats
        s1 = m.get_final_state_dict()
        assert isinstance(s1["stress"], LazyFinalRows) and np.array_equal(np.asarray(s1["stress"]), f0) and np.array_equal(s1["strain"], eps)
        ref_m.data_manager.update()
        m.data_manager.update()
        s0 = m.get_initial_state_dict()
        assert isinstance(s0["stress"], LazyInitialRows) and np.array_equal(np.asarray(s0["stress"]), f0)
        assert np.array_equal(np.asarray(m.get_final_state_dict()["stress"]), f0)      # s1 is served from s0 after advance
        assert np.array_equal(s0["p"], ref_m.get_initial_state_dict()["p"])
    # several updates from one s0, then revert: the state dictionaries follow
    f3 = np.array(ref_m.integrate(h[3])[0])
    m.integrate_rows(h[3], rows, flux_fn, jac_fn)
    assert np.array_equal(np.asarray(m.get_final_state_dict()["stress"]), f3) and np.array_equal(np.asarray(m.get_initial_state_dict()["stress"]), f0)
    m.data_manager.revert()
    assert np.array_equal(np.asarray(m.get_final_state_dict()["stress"]), f0)
    # the ordinary form keeps working on the same handle
    f1, _, c1 = m.integrate(h[3])
    assert np.array_equal(f1, f3) and isinstance(m.get_final_state_dict()["stress"], np.ndarray)
    with pytest.raises(ValueError):
        m.integrate_rows(h[3], rows.astype(np.int32), flux_fn, jac_fn)  # the index is int64
    with pytest.raises(ValueError):
        m.integrate_rows(h[3], rows + total, flux_fn, jac_fn)          # out of range
    if n > 1:   # the SAME index array, changed in place since it was last checked: caught (the check runs in every call)
        keep = rows[-1]
        rows[-1] = total + 5
        with pytest.raises(ValueError):
            m.integrate_rows(h[3], rows, flux_fn, jac_fn)
        rows[-1] = -1
        with pytest.raises(ValueError):
            m.integrate_rows(h[3], rows, flux_fn, jac_fn)
        rows[-1] = keep
        m.integrate_rows(h[3], rows, flux_fn, jac_fn)
    m.close()
    ref_m.close()


@pytest.mark.parametrize("law,n,total", [("elastic", 70_001, 90_000), ("elastic", 300, 300), ("fefp", 40_003, 50_000), ("fefp", 129, 700)])
def test_integrate_rows_for_the_elastic_and_the_finite_strain_law(law, n, total):
    """The same for the laws whose tangent does not come from (c1, c2, c3, w): the elastic law's constant block is filled in at
    the rows, the FeFp law's 9x9 block is rebuilt from its 54 building blocks there; both sides of the packed-transfer threshold."""
    from helpers import SIG0_F, SIGU_F, B_F, fefp_path

    def make():
        el = jm.LinearElasticIsotropic(E=E, nu=NU)
        return JAXMaterial(jm.ElasticBehavior(el) if law == "elastic" else jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F)))

    rng = np.random.default_rng(n)
    rows = np.ascontiguousarray(rng.permutation(total)[:n], dtype=np.int64)
    ref_m, m = make(), make()
    ref_m.set_data_manager(n)
    m.set_data_manager(n)
    nf = 6 if law == "elastic" else 9
    hist = j2_history(n, seed=2)[:3] if law == "elastic" else fefp_path(n, nsteps=4, eps=3e-2)
    flux_fn, jac_fn = np.full((total, nf), 3.0), np.full((total, nf * nf), 3.0)
    want_f, want_c = flux_fn.copy(), jac_fn.copy()
    for k, g in enumerate(hist):
        f0, i0, c0 = ref_m.integrate(g)
        isv = m.integrate_rows(g, rows, flux_fn, jac_fn)
        want_f[rows], want_c[rows] = f0, np.asarray(c0).reshape(n, nf * nf)
        assert np.array_equal(flux_fn, want_f) and np.array_equal(jac_fn, want_c), k
        assert np.array_equal(np.asarray(isv), np.asarray(i0)) and m.last_stats == ref_m.last_stats
        assert np.array_equal(np.asarray(m.get_final_state_dict()[m._fname]), f0)
        ref_m.data_manager.update()
        m.data_manager.update()
        assert np.array_equal(np.asarray(m.get_initial_state_dict()[m._fname]), f0)
    m.close()
    ref_m.close()


@pytest.mark.parametrize("law,layout,n,total", [("linear", "pack4", 300_007, 700_000), ("voce", "coef", 70_001, 70_001), ("linear", "sym", 150_003, 200_000),
                                                ("elastic", "sym", 40_001, 50_000), ("linear", "pack4", 1, 3)])
def test_integrate_rows_moves_the_rows_of_a_packed_tangent_layout(law, layout, n, total):
    """The rows forms for a handle whose own tangent layout is packed (round 6): the kernel's 21 / 9 / 4 numbers per point land in the
    library's page-locked area and are MOVED to row rows[i] of an ``(M, tangent_size)`` array -- nothing is rebuilt; stress and (bound)
    state fields go with them; bit-identical to ``integrate`` of the same layout followed by a fancy assignment."""
    rng = np.random.default_rng(n + 7)
    rows = np.ascontiguousarray(rng.permutation(total)[:n], dtype=np.int64)
    others = np.setdiff1d(np.arange(total), rows)
    if law == "elastic":
        make = lambda: JAXMaterial(jm.ElasticBehavior(jm.LinearElasticIsotropic(E=E, nu=NU)), tangent_layout=layout)   # noqa: E731
    else:
        make = lambda: _j2(law, tangent_layout=layout)   # noqa: E731
    ref_m, m = make(), make()
    ref_m.set_data_manager(n)
    m.set_data_manager(n)
    width = m.tangent_size
    assert m.supports_row_outputs and width == {"pack4": 4, "coef": 9, "sym": 21}[layout]
    flux_fn, jac_fn = np.full((total, 6), -7.0), np.full(total * width, -7.0)
    fields = {name: np.full(total * max(1, dim), -7.0) for name, dim in m.internal_state_variables.items()}
    if fields:
        m.bind_state_outputs(fields, deliver=True, rows=True)
    h = j2_history(n, seed=4, sig0=SIG0_V if law == "voce" else SIG0_LIN)
    for k, eps in enumerate(h[:3]):
        f0, i0, c0 = ref_m.integrate(eps)
        i0 = np.asarray(i0)
        m.integrate_rows(eps, rows, flux_fn, jac_fn)
        assert np.array_equal(flux_fn[rows], f0) and np.array_equal(jac_fn.reshape(total, width)[rows], np.asarray(c0).reshape(n, width)), k
        assert (flux_fn[others] == -7.0).all() and (jac_fn.reshape(total, width)[others] == -7.0).all()
        col = 0
        for name, dim in m.internal_state_variables.items():
            assert np.array_equal(fields[name].reshape(total, dim)[rows], i0[:, col:col + dim]), (k, name)
            col += dim
        assert m.last_stats == ref_m.last_stats
        ref_m.data_manager.update()
        m.data_manager.update()
    with pytest.raises(ValueError):
        m.integrate_rows(h[3], rows, flux_fn, np.zeros(total * 36 + 1))        # whole rows of tangent_size doubles
    m.close()
    ref_m.close()


def test_results_are_delivered_into_bound_caller_arrays():
    n = 70_000
    ref_m, m = _j2(), _j2()
    ref_m.set_data_manager(n)
    m.set_data_manager(n)
    flux_fn = np.full(n * 6, np.nan)      # the x.array of the flux Function
    jac_fn = np.full(n * 36, np.nan)      # the x.array of jacobian_flatten (quadrature_map.py:83-105)
    m.bind_outputs(flux=flux_fn, tangent=jac_fn)
    h = j2_history(n, seed=9)
    for eps in h[:3]:
        f0, _, c0 = ref_m.integrate(eps)
        f1, _, c1 = m.integrate(eps)
        assert np.shares_memory(f1, flux_fn) and np.shares_memory(c1, jac_fn)
        assert np.array_equal(flux_fn.reshape(n, 6), f0) and np.array_equal(jac_fn.reshape(n, 6, 6), c0)
        s0_flux = m.get_initial_state_dict()["stress"].copy()
        ref_m.data_manager.update()
        m.data_manager.update()
        assert np.array_equal(m.get_initial_state_dict()["stress"], f0)   # the s0 mirror is the converged flux ...
    m.integrate(h[3])
    assert np.array_equal(m.get_initial_state_dict()["stress"], f0)       # ... and survives the next integrate
    with pytest.raises(ValueError):
        m.bind_outputs(flux=np.zeros(5))
    m.close()
    flux_fn[:] = 0.0   # unregistered again: ordinary memory


@pytest.mark.parametrize("pinned_isv", [False, True])
@pytest.mark.parametrize("kind,n", [("linear", 300_001), ("voce", 2_200_000)])
def test_eager_isv_rows_and_bound_field_deliveries_in_one_call(kind, n, pinned_isv):
    """``lazy_isv=False`` (isv_aos handed to ``dxm_integrate``: interleaved (N, 7) rows) together with
    ``bind_state_outputs(deliver=True)`` (``dxm_bind_isv_output``: field-major rows into the ISV Functions) in a call of many
    chunks on two streams: the two layouts have their own device scratch (``dxmat.hip``: ``d_isv`` / ``d_isv_fields``), both for a
    pageable isv_aos (filled after the chunk loop) and a page-locked one (filled by DMA inside it)."""
    import ctypes as C

    ref_m, m = _j2(kind), _j2(kind, lazy_isv=False)
    ref_m.set_data_manager(n)
    m.set_data_manager(n)
    fields = {name: np.full(n * dim, np.nan) for name, dim in m.internal_state_variables.items()}
    m.bind_state_outputs(fields, deliver=True)
    assert m.delivers_state_outputs == frozenset(fields)
    total = sum(m.internal_state_variables.values())
    lib, h = m._lib, m._handles()[0]
    own = _lib.PinnedArray((n, total)) if pinned_isv else None
    isv_rows = own.array if pinned_isv else np.empty((n, total))
    flux, ct = np.empty((n, 6)), np.empty((n, 6, 6))
    st = _lib.Stats()
    for eps in j2_history(n, seed=17, sig0=SIG0_LIN if kind == "linear" else SIG0_V)[:3]:
        f0, i0, c0 = ref_m.integrate(eps)
        i0 = np.asarray(i0)
        # through the protocol: eager rows returned AND the Functions' memory written by the same call
        f1, i1, c1 = m.integrate(eps)
        assert isinstance(i1, np.ndarray) and np.array_equal(i1, i0) and np.array_equal(f1, f0) and np.array_equal(c1, c0)
        col = 0
        for name, dim in m.internal_state_variables.items():
            assert np.array_equal(fields[name].reshape(n, dim), i0[:, col:col + dim]), name
            fields[name][:] = np.nan
            col += dim
        # straight through the C ABI with the caller's own isv_aos (pageable / page-locked)
        isv_rows[...] = np.nan
        rc = lib.dxm_integrate(h, eps.ctypes.data_as(C.c_void_p), 0.0, flux.ctypes.data_as(C.c_void_p), isv_rows.ctypes.data_as(C.c_void_p),
                               ct.ctypes.data_as(C.c_void_p), C.byref(st))
        assert rc == 0 and np.array_equal(isv_rows, i0) and np.array_equal(flux, f0) and np.array_equal(ct, c0)
        col = 0
        for name, dim in m.internal_state_variables.items():
            assert np.array_equal(fields[name].reshape(n, dim), i0[:, col:col + dim]), name
            col += dim
        ref_m.data_manager.update()
        m.data_manager.update()
    m.close()
    ref_m.close()


@pytest.mark.parametrize("layout", ["full", "pack4", "sym"])
@pytest.mark.parametrize("kind,n", [("linear", 300_001), ("voce", 2_200_000)])
def test_the_three_stream_scheme_delivers_the_bits_of_the_alternating_chunks(kind, n, layout):
    """Option ``split_streams`` (default 1 for page-locked inputs: uploads + kernels on one stream, the downloads of chunk c on one
    of two others behind an event, at most 24 chunks) against 0 (whole chunks alternating on two streams, up to 64): same stress,
    tangent, delivered state fields and status record, with bound (page-locked in place) input and output arrays as the accelerated
    map sets them up; a pageable gradient array takes the staged route, which keeps the alternating streams either way."""
    width = {"full": 36, "pack4": 4, "sym": 21}[layout]
    mats, outs = [], []
    h = j2_history(n, seed=23, sig0=SIG0_LIN if kind == "linear" else SIG0_V)
    for split in (1, 0):
        m = _j2(kind, tangent_layout=layout)
        m.set_data_manager(n)
        m.set_option("split_streams", split)
        flux_fn, jac_fn, grad_fn = np.full(n * 6, np.nan), np.full(n * width, np.nan), np.zeros(n * 6)
        fields = {name: np.full(n * dim, np.nan) for name, dim in m.internal_state_variables.items()}
        m.bind_outputs(flux=flux_fn, tangent=jac_fn)
        m.bind_inputs(gradient=grad_fn)
        m.bind_state_outputs(fields, deliver=True)
        mats.append(m)
        outs.append((flux_fn, jac_fn, grad_fn, fields))
    for k in range(3):
        stats = []
        for m, (flux_fn, jac_fn, grad_fn, fields) in zip(mats, outs):
            grad_fn[...] = h[k].ravel()
            m.integrate(grad_fn.reshape(n, 6))
            stats.append(dict(m.last_stats))
            m.data_manager.update()
        (fa, ja, _, sa), (fb, jb, _, sb) = outs
        assert np.array_equal(fa, fb) and np.array_equal(ja, jb) and not np.isnan(ja).any(), k
        assert all(np.array_equal(sa[name], sb[name]) for name in sa) and stats[0] == stats[1], k
    # a pageable array (a new one per call, like the reference's update()): staged through the ring on both handles, same bits
    res = [np.array(m.integrate(np.array(h[3]))[0]) for m in mats]
    assert np.array_equal(res[0], res[1])
    for m in mats:
        m.close()


@pytest.mark.parametrize("kind,n", [("linear", 40_001), ("voce", 300_001), ("linear", 2_200_000)])
def test_sym_layout_host_calls_rebuild_the_21_entries_from_the_four_coefficients(kind, n):
    """A handle with the ``"sym"`` tangent layout in the host-buffer form: with ``packed_transfer = 2`` (default) only (c1, c2, c3, w)
    cross PCIe -- 32 instead of 168 B/point -- and the worker threads rebuild the 21 upper-triangle entries from them and the stress
    (``host_side.hpp::expand_pack4_tangent_sym``); ``packed_transfer = 0`` downloads the kernel's own 21 entries.  Same bits, with the
    material's own arrays and with bound ones; and they are the upper triangle of the full-layout handle's block."""
    from dolfinx_materials_amd.conventions import pack_sym_tangent

    hist = j2_history(n, seed=9, sig0=SIG0_V if kind == "voce" else SIG0_LIN)
    mats = []
    for level, bind in ((2, False), (2, True), (0, False)):
        m = _j2(kind, tangent_layout="sym")
        m.set_data_manager(n)
        m.set_option("packed_transfer", level)
        if bind:
            keep = (np.zeros(n * 6), np.zeros(n * 21))
            m.bind_outputs(flux=keep[0], tangent=keep[1])
        mats.append(m)
    full = _j2(kind)
    full.set_data_manager(n)
    for eps in hist[:3]:
        out = [m.integrate(eps) for m in mats]
        assert out[0][2].shape == (n, 21)
        for f, i, c in out[1:]:
            assert np.array_equal(f, out[0][0]) and np.array_equal(c, out[0][2]) and np.array_equal(np.asarray(i), np.asarray(out[0][1]))
        ff, _, cf = full.integrate(eps)
        assert np.array_equal(ff, out[0][0]) and np.array_equal(pack_sym_tangent(cf), out[0][2])
        for m in mats + [full]:
            m.data_manager.update()
    for m in mats + [full]:
        m.close()


def test_options_replace_environment_variables():
    m = _j2()
    m.set_data_manager(1000)
    g0 = m.launch_generation
    for name, value in (("pipeline", 0), ("packed_transfer", 0), ("host_threads", 2), ("max_chunks", 4), ("fused_gradient", 0),
                        ("blocks_per_cu", 8), ("verbose", 0)):
        m.set_option(name, value)
    assert m.launch_generation > g0
    sig = m.integrate(j2_history(1000)[2])[0].copy()
    m.set_option("blocks_per_cu", 32)
    assert np.array_equal(m.integrate(j2_history(1000)[2])[0], sig)
    for name, value in (("no_such_option", 1), ("host_threads", 0), ("max_chunks", 999), ("blocks_per_cu", 1e6)):
        with pytest.raises(_lib.DxmError):
            m.set_option(name, value)


def test_launch_generation_tells_when_a_captured_graph_is_stale():
    """A captured dxm_integrate_device bakes in the two state buffers and the parameters: advance() swaps the
    buffers (the generation's low bit flips and comes back at the next advance), parameter changes bump it."""
    torch = pytest.importorskip("torch")
    n = 30_000
    dev = torch.device("cuda:0")
    m = _j2()
    m.set_data_manager(n)
    h = j2_history(n, seed=21)
    g = [to_device(x) for x in h]
    f = torch.zeros((n, 6), dtype=torch.float64, device=dev)
    c = torch.zeros((n, 36), dtype=torch.float64, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    gin = torch.empty_like(g[0])

    def capture():
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            m.integrate_device(gin.data_ptr(), f.data_ptr(), c.data_ptr(), st())
        return gr, m.launch_generation

    graphs = {}
    hard = onp.LinearHardening(SIG0_LIN, H_LIN)
    epsp, p = np.zeros((n, 6)), np.zeros(n)
    for k in range(4):
        gen = m.launch_generation
        if gen not in graphs:                      # at most two captures serve the whole history
            graphs[gen] = capture()[0]
        gin.copy_(g[k])
        graphs[gen].replay()
        m.notify_replay()                          # the replay is invisible to the library
        rc, stats = m.stats()                      # no event of the replay exists: falls back to a device sync
        assert rc == 0 and stats["n_nan"] == 0
        ref = onp.j2_update(h[k], epsp, p, E, NU, hard)
        safe = np.abs(ref["f_trial"]) > 1e-9 * SIG0_LIN
        assert np.abs(to_host(f)[safe] - ref["sig"][safe]).max() <= 1e-12 * np.abs(ref["sig"]).max()
        assert np.abs(to_host(c).reshape(n, 6, 6)[safe] - ref["Ct"][safe]).max() <= 1e-12 * np.abs(ref["Ct"]).max()
        epsp, p = ref["epsp"], ref["p"]
        m.data_manager.update()
        assert m.launch_generation != gen and (m.launch_generation ^ gen) == 1
    assert len(graphs) == 2
    gen = m.launch_generation
    m.data_manager.revert()
    assert m.launch_generation == gen              # revert does not move the buffers
    m.update_material_property("yield_stress.H", 6e3)
    assert m.launch_generation not in graphs       # a stale graph would integrate with the old hardening modulus


def test_stats_after_the_launch_stream_is_gone():
    """The handle waits on its own event, not on the caller's stream handle (which may have been destroyed)."""
    torch = pytest.importorskip("torch")
    n = 20_000
    dev = torch.device("cuda:0")
    m = _j2()
    m.set_data_manager(n)
    g = to_device(j2_history(n)[2])
    f = torch.zeros((n, 6), dtype=torch.float64, device=dev)
    c = torch.zeros((n, 36), dtype=torch.float64, device=dev)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        m.integrate_device(g.data_ptr(), f.data_ptr(), c.data_ptr(), side.cuda_stream)
    side.synchronize()
    del side
    import gc

    gc.collect()
    rc, stats = m.stats()
    assert rc == 0 and stats["n_plastic"] > 0 and m.get_final_state_dict()["p"].max() > 0


def test_returned_arrays_outlive_the_material():
    """The arrays integrate() / the state dicts hand out own their page-locked memory: close(), a second
    set_data_manager() (QuadratureMap re-binding cells) or dropping the material must not invalidate them
    (round 1 freed the buffers under live numpy views)."""
    import gc
    import sys
    import os

    n = 3000
    m = _j2(lazy_isv=False)
    m.set_data_manager(n)
    eps = j2_history(n, seed=2)[2]
    sig, isv, ct = m.integrate(eps)
    keep = (sig.copy(), isv.copy(), ct.copy())
    fin = m.get_final_state_dict()["epsp"]
    fin_copy = fin.copy()
    m.set_data_manager(2 * n)             # frees the old handle and drops the old buffers
    m.integrate(np.concatenate([eps, eps]))
    m.close()
    del m
    gc.collect()
    junk = [np.full(50_000, 7.0) for _ in range(20)]   # churn the allocator
    assert np.array_equal(sig, keep[0]) and np.array_equal(isv, keep[1]) and np.array_equal(ct, keep[2])
    assert np.array_equal(fin, fin_copy) and len(junk) == 20
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from standalone_batch import main

    P, isv2, Ct2 = main(6)                # returns after its material went out of scope
    gc.collect()
    assert P.shape == (6, 9) and Ct2.shape == (6, 9, 9) and np.isfinite(P).all() and np.isfinite(Ct2).all() and isv2[0, 0] > 1e-2


def test_elastic_host_path_fills_the_constant_block_without_moving_it():
    """Elastic law, full layout, >= 32768 points: no tangent bytes cross PCIe, worker threads fill the constant block;
    equal to the device-computed block and to python_materials/elasticity.py:15-19."""
    n = 300_001
    el = jm.ElasticBehavior(jm.LinearElasticIsotropic(E=E, nu=NU))
    a, b = JAXMaterial(el), JAXMaterial(el)
    a.set_data_manager(n)
    b.set_data_manager(n)
    b.set_option("packed_transfer", 0)
    eps = j2_history(n, seed=3)[2]
    fa, ia, ca = a.integrate(eps)
    fb, ib, cb = b.integrate(eps)
    assert np.array_equal(fa, fb) and np.array_equal(ca, cb) and np.asarray(ia).shape == (n, 0)
    assert np.array_equal(ca[12345], onp.elastic_matrix(E, NU))


@pytest.mark.parametrize("n", [1000, 300_001])
def test_staged_and_runtime_uploads_give_the_same_bits(n):
    """Default: the pageable strain array goes through the library's page-locked ring (worker-thread copies + copy
    kernel); option pageable_dma = 1: the runtime transfers it itself.  Same results, bit for bit."""
    a, b = _j2("voce"), _j2("voce")
    a.set_data_manager(n)
    b.set_data_manager(n)
    b.set_option("pageable_dma", 1)
    for eps in j2_history(n, seed=9, sig0=SIG0_V)[:3]:
        fa, ia, ca = a.integrate(eps)
        fb, ib, cb = b.integrate(eps)
        assert np.array_equal(fa, fb) and np.array_equal(ca, cb) and np.array_equal(np.asarray(ia), np.asarray(ib))
        a.data_manager.update()
        b.data_manager.update()


def test_host_path_beyond_one_status_record_per_workgroup_of_a_single_launch():
    """2e7 points through the chunked host-buffer form: 32 chunks of 2442 workgroups each append 78 144 block records,
    more than the largest single launch (256 CUs x 256) the status buffer used to be sized for -- `dxm_integrate` then
    failed part-way with "internal: stats buffer too small".  Strided sample against the oracle, whole batch through
    the status record."""
    n = 20_000_000
    rng = np.random.default_rng(77)
    eps = rng.standard_normal((n, 6))
    eps *= (rng.uniform(0.0, 4.0, n) * SIG0_LIN / (2 * (E / 2 / (1 + NU))) * np.sqrt(2.0 / 3.0) / np.linalg.norm(eps, axis=1))[:, None]
    m = _j2(tangent_layout="coef")        # 72 B/point of tangent: keeps the host arrays of this test at ~4 GB
    m.set_data_manager(n)
    sig, isv, coef = m.integrate(eps)
    assert m.last_stats["n_points"] == n and m.last_stats["n_nan"] == 0 and m.last_stats["n_not_converged"] == 0
    idx = np.concatenate([np.arange(0, n, 40_009), [n - 1, n - 255, n - 256, n - 257]])
    ref = onp.j2_update(eps[idx], np.zeros((len(idx), 6)), np.zeros(len(idx)), E, NU, onp.LinearHardening(SIG0_LIN, H_LIN))
    safe = np.abs(ref["f_trial"]) > 1e-9 * SIG0_LIN
    assert np.abs(sig[idx][safe] - ref["sig"][safe]).max() <= 1e-12 * np.abs(ref["sig"]).max()
    from dolfinx_materials_amd.conventions import tangent_from_coefficients
    assert np.abs(tangent_from_coefficients(coef[idx])[safe] - ref["Ct"][safe]).max() <= 1e-12 * np.abs(ref["Ct"]).max()
    # the plastic count of the whole batch: same yield test on the host (points at the surface excluded from neither side)
    mu = E / 2 / (1 + NU)
    dev = eps.copy()
    dev[:, :3] -= eps[:, :3].mean(axis=1)[:, None]
    seq = np.sqrt(1.5) * 2 * mu * np.linalg.norm(dev, axis=1)
    assert abs(m.last_stats["n_plastic"] - int((seq > SIG0_LIN).sum())) <= int((np.abs(seq - SIG0_LIN) < 1e-9 * SIG0_LIN).sum())
    m.close()


@pytest.mark.parametrize("kind,n", [("linear", 40_001), ("voce", 300_001)])
def test_every_level_of_packed_transfer_delivers_the_same_bits(kind, n):
    """``packed_transfer`` 2 (default: (c1, c2, c3, w) cross PCIe, 32 B/point, the flow direction is rebuilt from the
    stress), 1 (the nine coefficients, 72 B/point), 0 (the full block, 288 B/point): identical stress, tangent and state,
    with own and with bound output arrays; level 2 falls back to level 1 when the caller does not take the stress."""
    import ctypes as C

    hist = j2_history(n, seed=9, sig0=SIG0_V if kind == "voce" else SIG0_LIN)
    mats = []
    for level, bind in ((2, False), (2, True), (1, False), (0, False)):
        m = _j2(kind)
        m.set_data_manager(n)
        m.set_option("packed_transfer", level)
        if bind:
            keep = (np.zeros(n * 6), np.zeros(n * 36))
            m.bind_outputs(flux=keep[0], tangent=keep[1])
        mats.append(m)
    for eps in hist:
        out = [m.integrate(eps) for m in mats]
        for f, i, c in out[1:]:
            assert np.array_equal(f, out[0][0]) and np.array_equal(c, out[0][2]) and np.array_equal(np.asarray(i), np.asarray(out[0][1]))
        assert np.array_equal(out[0][2], out[0][2].transpose(0, 2, 1))
        for m in mats:
            m.data_manager.update()
    # no stress destination: the C ABI still delivers the tangent (through the nine coefficients)
    m = mats[0]
    ct = np.zeros((n, 6, 6))
    st = _lib.Stats()
    rc = m._lib.dxm_integrate(m._handle, hist[2].ctypes.data, 0.0, None, None, ct.ctypes.data, C.byref(st))
    assert rc == 0
    ref = mats[3].integrate(hist[2])[2]
    assert np.array_equal(ct, ref)
    with pytest.raises(_lib.DxmError):
        m.set_option("packed_transfer", 3)
    for m in mats:
        m.close()


def test_the_handle_measures_how_a_pageable_gradient_array_goes_up_fastest():
    """Default (`register_input` = 1): the first call page-locks, calls 2-5 alternate between page-locking for the call and the
    staging ring, then the faster way is kept (which one depends on the host) with one trial of the other in 32 calls; every call
    delivers the same bits, and `last_upload` says what it did."""
    n = 400_003
    hist = j2_history(n, seed=8)
    a, b = _j2(), _j2()
    a.set_data_manager(n)
    b.set_data_manager(n)
    b.set_option("register_input", 0)
    ways = []
    for k in range(45):
        eps = hist[1 + k % 3]
        fa, _, ca = a.integrate(np.array(eps))
        ways.append(a.last_upload)
        if k < 8 or k % 9 == 0:
            fb, _, cb = b.integrate(np.array(eps))
            assert np.array_equal(fa, fb) and np.array_equal(ca, cb), k
    locked, staged = "dma (page-locked for the call)", "staged through the ring"
    assert ways[:5] == [locked, locked, staged, locked, staged]
    kept = ways[5]
    assert kept in (locked, staged)
    other = staged if kept == locked else locked
    assert ways[5:36] == [kept] * 31 and ways[36] == other            # the 32nd call after the calibration tries the other way
    assert all(w in (locked, staged) for w in ways[37:])
    a.set_option("register_input", 1)                                  # setting the option starts the measurement again
    a.integrate(np.array(hist[1]))
    a.integrate(np.array(hist[1]))
    a.integrate(np.array(hist[1]))
    assert a.last_upload == staged
    a.close()
    b.close()


def test_pageable_gradient_is_page_locked_for_the_call_only():
    """`register_input` = 2: a gradient array in ordinary memory is registered for the duration of `dxm_integrate`
    and uploaded by DMA; staged through the ring with the option off.  Same bits either way, and the range is unregistered
    again when the call returns: registering it explicitly afterwards succeeds (a range that is still registered is
    refused by the runtime), also after a call that fails part-way."""
    n = 200_003
    hist = j2_history(n, seed=21)
    a, b = _j2(), _j2()
    a.set_data_manager(n)
    b.set_data_manager(n)
    a.set_option("register_input", 2)
    b.set_option("register_input", 0)
    lib = a._lib
    for eps in hist[:3]:
        g = np.array(eps)                       # a fresh array per call, as QuadratureMap.update hands over
        fa, ia, ca = a.integrate(g)
        fb, ib, cb = b.integrate(np.array(eps))
        assert np.array_equal(fa, fb) and np.array_equal(ca, cb) and np.array_equal(np.asarray(ia), np.asarray(ib))
        assert a.last_upload == "dma (page-locked for the call)" and b.last_upload == "staged through the ring"
        assert lib.dxm_host_register(g.ctypes.data, g.nbytes) == 0, _lib.last_error(lib)   # not registered any more
        assert lib.dxm_host_unregister(g.ctypes.data) == 0
        a.data_manager.update()
        b.data_manager.update()
    # a view that does not start at the array's first byte, and a read-only array
    big = np.zeros((n + 7, 6))
    big[7:] = hist[3]
    ro = np.array(hist[3])
    ro.setflags(write=False)
    ref = b.integrate(np.array(hist[3]))[0].copy()
    assert np.array_equal(a.integrate(big[7:])[0], ref)
    assert np.array_equal(a.integrate(ro)[0], ref)
    pin = _lib.PinnedArray(hist[3].shape)
    pin.array[...] = hist[3]
    assert np.array_equal(a.integrate(pin.array)[0], ref) and a.last_upload == "dma (caller's array page-locked)"
    small = _j2()
    small.set_data_manager(1000)
    small.integrate(np.array(hist[0][:1000]))
    assert small.last_upload == "staged through the ring"     # below 1 MB the registration is not worth a system call
    small.close()
    a.close()
    b.close()
