"""CPU: the Python layer above the C ABI -- ``hip_material.HIPMaterial`` (s0 / s1 mirrors, the views its state dictionaries hand
out, bound arrays, rows mode, lazy ISVs) and ``quadrature_map.AcceleratedUpdate`` on top of it -- driven through the SAME random
protocol sequences as ``tests/test_gpu_fuzz_protocol.py``, against ``tests/fake_dxmat.py``: a test double of libdxmat.so that restates
the handle semantics of ``csrc/dxmat.hip`` (which device copies a handle holds, when it drops them) over numpy with the C oracle's
arithmetic.  This is what found "revert(); set_initial_state_dict(isv); advance()" leaving a lazy mirror without its device copy
only on the GPU box in round 5; now the host logic is exercised on every CPU run.  The product path is untouched: ``_lib.load`` is
monkeypatched for the duration of a test."""
import gc

import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd import _lib
from dolfinx_materials_amd.jaxmat import JAXMaterial
from fake_dxmat import FakeDxmat
from helpers import E, NU, SIG0_LIN, H_LIN, j2_history


@pytest.fixture
def fake(monkeypatch):
    lib = FakeDxmat(_lib.load())
    monkeypatch.setattr(_lib, "load", lambda *a, **k: lib)
    yield lib
    gc.collect()


@pytest.mark.parametrize("seed,n,bound,lazy", [(0, 77, False, True), (1, 500, True, True), (2, 64, False, False), (3, 400, True, False),
                                               (4, 1, False, True), (6, 300, "io", True), (7, 660, "io", True), (8, 130, "io", False),
                                               (9, 250, "rows", True), (10, 703, "rows", True), (11, 1, "rows", True),
                                               (21, 90, "io", True), (22, 90, "rows", True), (23, 90, True, True), (24, 90, False, True),
                                               (12, 310, "rows_isv", True), (25, 90, "rows_isv", True)])
def test_random_operation_sequences_on_the_test_double(fake, seed, n, bound, lazy):
    from test_gpu_fuzz_protocol import run_operation_sequence

    run_operation_sequence(seed, n, bound, lazy, device_ops=False, nops=96 if seed > 20 else 64)


def _j2():
    return JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)))


def test_views_are_counted_by_weak_references_not_reference_counts(fake):
    """The CPU twin of tests/test_gpu_hostpath.py::test_views_of_the_initial_state_keep_their_content_however_they_are_held: one
    download per field when a held view's state is replaced, none for dictionaries nobody kept."""
    from dolfinx_materials_amd.hip_material import LazyInitialRows

    n = 501
    m = _j2()
    m.set_data_manager(n)
    flux_fn, jac_fn, grad_fn = np.zeros(n * 6), np.zeros(n * 36), np.zeros(n * 6)
    m.bind_outputs(flux=flux_fn, tangent=jac_fn)
    m.bind_inputs(gradient=grad_fn)
    rows = grad_fn.reshape(n, 6)
    h = j2_history(n, seed=33)

    def cycle(k):
        rows[...] = h[k]
        f = np.array(m.integrate(rows)[0])
        m.data_manager.update()
        return f

    f0 = cycle(0)
    local = m.get_initial_state_dict()["stress"]
    bag = [m.get_initial_state_dict()["stress"]]
    book = {}
    v = m.get_initial_state_dict()["strain"]
    book[id(v)] = v
    del v
    m.get_initial_state_dict()                       # nobody keeps this one
    gc.collect()
    assert all(isinstance(x, LazyInitialRows) and not x.fetched for x in (local, bag[0], *book.values())) and not fake.downloads
    cycle(1)
    assert sorted(fake.downloads) == [(0, 0), (0, 1)]
    cycle(2)
    assert len(fake.downloads) == 2
    assert np.array_equal(np.asarray(local), f0) and np.array_equal(np.asarray(bag[0]), f0)
    assert np.array_equal(np.asarray(next(iter(book.values()))), h[0])
    del local, bag, book
    gc.collect()
    cycle(3)
    assert len(fake.downloads) == 2                   # nothing held any more: nothing downloaded
    # revert + an ISV set + advance: s1 gets its own storage back and the device drops the copies the mirror stood for
    f3 = np.array(flux_fn.reshape(n, 6))
    rows[...] = h[0]
    m.integrate(rows)
    m.data_manager.revert()
    kept = m.get_final_state_dict()["stress"]
    m.set_initial_state_dict({"p": np.full(n, 1e-3)})
    m.data_manager.update()
    rows[...] = h[1]
    m.integrate(rows)
    m.data_manager.update()
    assert np.array_equal(np.asarray(kept), f3)
    m.close()


def test_accelerated_map_on_the_test_double_writes_isvs_inside_integrate(fake):
    """``AcceleratedUpdate`` over ``HIPMaterial`` over the test double: default mode = ISV Functions written by the host-buffer call
    itself (``dxm_bind_isv_output``), bit-identical fields to the reference cadence around a second material."""
    from bench import as_reference_advance, as_reference_update
    from dolfinx_materials_amd.field_map import FieldMapBase, QuadratureFieldMap

    ncell, nqp = 60, 4
    n = ncell * nqp
    hist = j2_history(n, seed=5)
    now = {"g": hist[0]}
    ev = lambda c: now["g"].reshape(ncell, nqp, 6)[c].reshape(-1, 6)   # noqa: E731
    fast, slow = QuadratureFieldMap(ncell, nqp, _j2()), FieldMapBase(ncell, nqp, _j2())
    for q in (fast, slow):
        q.register_gradient("strain", ev)
    held = dict(fast._isv_functions())
    reads = []
    inner = fast.material.read_final_state
    fast.material.read_final_state = lambda name, out: (reads.append(name), inner(name, out))[1]
    for k, g in enumerate(hist):
        now["g"] = g
        fast.update()
        as_reference_update(slow)
        assert fast.material.delivers_state_outputs == {"p", "epsp"} and not reads
        for name in ("p", "epsp"):
            assert np.array_equal(held[name].x.array, slow.internal_state_variables[name].x.array), (k, name)
        assert np.array_equal(fast.fluxes["stress"].x.array, slow.fluxes["stress"].x.array)
        assert np.array_equal(fast.jacobian_flatten.x.array, slow.jacobian_flatten.x.array)
        fast.advance()
        as_reference_advance(slow)
        del reads[:]
    fast.close()
    for q in (fast, slow):
        q.material.close()


@pytest.mark.parametrize("seed,subset", [(20, False), (21, True), (22, True), (23, False), (30, False), (31, True), (32, False), (33, True)])
def test_field_map_sequences_on_the_test_double(fake, seed, subset):
    """tests/test_gpu_fuzz_protocol.py's differential test one level up (the same random QuadratureFieldMap operations with
    HIPMaterial behind one map and the oracle-backed material behind the other), with the test double under HIPMaterial: bound
    Functions, ISV delivery inside `integrate` (the default mode), rows mode for maps over a subset of the cells."""
    from test_gpu_fuzz_protocol import test_field_map_driven_by_the_engine_equals_field_map_driven_by_the_oracle as run

    run(seed, subset)


def test_state_fields_bound_for_row_delivery_serve_the_rows_forms_only(fake):
    """``bind_state_outputs(deliver=True, rows=True)``: the ISV Functions over ALL cells of a map over a subset; ``integrate_rows`` puts
    the fields of point i into row rows[i] (here through the test double, on the GPU by the worker threads of ``dxm_integrate_rows``),
    ``integrate`` refuses while the binding is in place, unbinding restores it."""
    from dolfinx_materials_amd._lib import DxmError
    from oracle import oracle_c

    n, total = 300, 337
    m = _j2()
    m.set_data_manager(n)
    rows = np.ascontiguousarray(np.random.default_rng(5).permutation(total)[:n], dtype=np.int64)
    flux_all, jac_all = np.full((total, 6), 9.0), np.full((total, 36), 9.0)
    fields = {"p": np.full(total, 9.0), "epsp": np.full(total * 6, 9.0)}
    with pytest.raises(ValueError):
        m.bind_state_outputs(fields, rows=True)                      # rows describes deliveries
    with pytest.raises(ValueError):
        m.bind_state_outputs({"p": np.zeros(n - 1)}, deliver=True, rows=True)
    m.bind_state_outputs(fields, deliver=True, rows=True)
    assert m.delivers_state_outputs == {"p", "epsp"}
    eps = j2_history(n, seed=12)[2]
    m.integrate_rows(eps, rows, flux_all, jac_all)
    ref = oracle_c.j2(eps, np.zeros((n, 6)), np.zeros(n), E, NU, 0, SIG0_LIN, H_LIN)
    others = np.setdiff1d(np.arange(total), rows)
    assert np.array_equal(fields["p"][rows], ref["p"]) and np.array_equal(fields["epsp"].reshape(total, 6)[rows], ref["epsp"])
    assert (fields["p"][others] == 9.0).all() and (fields["epsp"].reshape(total, 6)[others] == 9.0).all() and ref["n_plastic"] > 0
    with pytest.raises(DxmError, match="ROWS"):
        m.integrate(eps)
    for name in fields:
        m._unbind("isv:" + name)
    assert m.delivers_state_outputs == frozenset()
    m.integrate(eps)
    m.close()
