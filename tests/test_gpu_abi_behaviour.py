"""GPU: behaviour of the C ABI beyond numerics -- errors, stream semantics, graph capture (hard errors are negative return
codes surfaced as DxmError; the reference raises Python exceptions at the same places)."""
import ctypes as C

import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd import _lib
from dolfinx_materials_amd.jaxmat import JAXMaterial

pytestmark = pytest.mark.gpu
from helpers import to_device, to_host  # noqa: E402,F401



def _mat(n=16):
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=70e3, nu=0.3), jm.LinearHardening(250.0, 5e3)))
    m.set_data_manager(n)
    return m


def test_create_rejects_bad_arguments():
    lib = _lib.load()
    p = (C.c_double * 4)(70e3, 0.3, 250.0, 5e3)
    assert not lib.dxm_create(1, p, 3, 8, 0) and "expects 4 parameters" in _lib.last_error()
    assert not lib.dxm_create(9, p, 4, 8, 0) and "unknown law" in _lib.last_error()
    assert not lib.dxm_create(1, p, 4, 8, 99) and "out of range" in _lib.last_error()
    bad = (C.c_double * 4)(70e3, 0.5, 250.0, 5e3)
    assert not lib.dxm_create(1, bad, 4, 8, 0) and "invalid elastic constants" in _lib.last_error()
    assert not lib.dxm_create(1, p, 4, -1, 0)


def test_wrong_shapes_and_pointers():
    torch = pytest.importorskip("torch")
    m = _mat(16)
    with pytest.raises(ValueError):
        m.integrate(np.zeros((15, 6)))
    with pytest.raises(ValueError):
        m.integrate(np.zeros((16, 9)))
    g = torch.zeros(16 * 6 + 1, dtype=torch.float64, device="cuda:0")
    f = torch.zeros((16, 6), dtype=torch.float64, device="cuda:0")
    c = torch.zeros((16, 36), dtype=torch.float64, device="cuda:0")
    with pytest.raises(_lib.DxmError, match="16-byte aligned"):
        m.integrate_device(g.data_ptr() + 8, f.data_ptr(), c.data_ptr())
    with pytest.raises(_lib.DxmError, match="null device pointer"):
        m.integrate_device(g.data_ptr(), 0, c.data_ptr())
    lib = _lib.load()
    buf = np.zeros((16, 6))
    assert lib.dxm_get_state(m._handle, 0, 7, buf.ctypes.data) < 0 and "no state field" in _lib.last_error()
    assert lib.dxm_get_state(m._handle, 5, 0, buf.ctypes.data) < 0
    assert lib.dxm_set_newton(m._handle, 0, 1e-14) < 0
    assert lib.dxm_set_tangent_layout(m._handle, 7) < 0


def test_non_convergence_is_reported_not_hidden():
    """A Newton cap that is too small must surface as a positive return code / warning with the
    number of affected points (the reference's JAX path reports nothing)."""
    from helpers import SIG0_V, SIGU_V, B_V, j2_history

    n = 2000
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=70e3, nu=0.3), jm.VoceHardening(SIG0_V, SIGU_V, B_V)))
    m.set_data_manager(n)
    m.set_newton(maxit=1, rtol=1e-14)
    with pytest.warns(RuntimeWarning, match="did not converge"):
        m.integrate(j2_history(n, sig0=SIG0_V)[2])
    assert 0 < m.last_stats["n_not_converged"] <= m.last_stats["n_plastic"]
    m.set_newton(maxit=25, rtol=1e-14)
    m.integrate(j2_history(n, sig0=SIG0_V)[2])
    assert m.last_stats["n_not_converged"] == 0


def test_nan_input_is_counted():
    m = _mat(64)
    eps = np.zeros((64, 6))
    eps[5, 2] = np.nan
    m.integrate(eps)
    assert m.last_stats["n_nan"] == 1


def test_unknown_property_and_callable_hardening():
    m = _mat(4)
    with pytest.raises(ValueError):
        m.update_material_property("elasticity.G", 1.0)
    with pytest.raises(NotImplementedError, match="varies from point to point"):
        m.update_material_property("elasticity.E", np.array([1.0, 2.0, 3.0, 4.0]))
    # what QuadratureMap.update_material_properties hands over (quadrature_map.py:160-172): 0-d arrays for
    # numbers, one value per Gauss point for UFL-valued properties; a uniform field is a number
    m.update_material_property("elasticity.E", np.asarray(71e3))
    m.update_material_property("elasticity.E", np.full(4, 72e3))
    assert m.material_properties["elasticity.E"] == 72e3 and m.behavior.elasticity.E == 72e3


def test_device_path_on_a_non_default_torch_stream():
    """The stream handle passed through the ABI is honoured: work enqueued on a side stream is
    ordered with torch work on that stream and invisible to the others until synchronised."""
    torch = pytest.importorskip("torch")
    from oracle import constitutive_np as onp
    from helpers import j2_history

    n = 200_000
    dev = torch.device("cuda:0")
    m = _mat(n)
    eps_h = j2_history(n)[2]
    side = torch.cuda.Stream()
    f = torch.zeros((n, 6), dtype=torch.float64, device=dev)
    c = torch.zeros((n, 36), dtype=torch.float64, device=dev)
    with torch.cuda.stream(side):
        g = to_device(eps_h) * 1.0   # produced on the side stream
        m.integrate_device(g.data_ptr(), f.data_ptr(), c.data_ptr(), side.cuda_stream)
        total = f.sum()                                                 # consumed on the side stream
    side.synchronize()
    ref = onp.j2_update(eps_h, np.zeros((n, 6)), np.zeros(n), 70e3, 0.3, onp.LinearHardening(250.0, 5e3))
    assert abs(float(total) - ref["sig"].sum()) < 1e-6 * np.abs(ref["sig"]).sum()
    assert np.abs(to_host(f) - ref["sig"]).max() < 1e-9 * np.abs(ref["sig"]).max()


def test_device_path_is_graph_capturable():
    """dxm_integrate_device makes no allocation and no synchronisation, so a caller can capture a
    whole Newton-iteration cadence (several updates) in a HIP graph and replay it."""
    torch = pytest.importorskip("torch")
    from helpers import j2_history

    n = 50_000
    dev = torch.device("cuda:0")
    m = _mat(n)
    h = j2_history(n)
    g = [to_device(x) for x in h[:3]]
    f = torch.zeros((n, 6), dtype=torch.float64, device=dev)
    c = torch.zeros((n, 36), dtype=torch.float64, device=dev)
    # eager reference: three updates from the same s0, last one wins
    for x in g:
        m.integrate_device(x.data_ptr(), f.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    f_ref, c_ref = f.clone(), c.clone()
    f.zero_()
    c.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for x in g:
            m.integrate_device(x.data_ptr(), f.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert float(f.abs().max()) == 0.0  # nothing ran during capture
    graph.replay()
    m.notify_replay()
    torch.cuda.synchronize()
    assert torch.equal(f, f_ref) and torch.equal(c, c_ref)


def test_create_destroy_cycles_do_not_leak_device_memory():
    torch = pytest.importorskip("torch")
    from helpers import j2_history

    torch.cuda.synchronize()
    n = 200_000
    eps = j2_history(n)[2]
    free0, _ = torch.cuda.mem_get_info()
    for k in range(30):
        m = _mat(n)
        m.integrate(eps)
        m.data_manager.update()
        m.get_final_state_dict()
        m.close()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, f"device memory shrank by {(free0 - free1) >> 20} MiB over 30 create/destroy cycles"


def test_several_handles_side_by_side():
    """Independent handles (e.g. one QuadratureMap per material region) do not interfere."""
    from oracle import constitutive_np as onp
    from helpers import j2_history

    n = 5000
    hs = j2_history(n, seed=3)
    mats = [_mat(n) for _ in range(4)]
    outs = [m.integrate(hs[k % 3 + 0] * (1 + 0.1 * k))[0].copy() for k, m in enumerate(mats)]
    for k, (m, got) in enumerate(zip(mats, outs)):
        ref = onp.j2_update(hs[k % 3] * (1 + 0.1 * k), np.zeros((n, 6)), np.zeros(n), 70e3, 0.3, onp.LinearHardening(250.0, 5e3))
        assert np.abs(got - ref["sig"]).max() < 1e-9 * np.abs(ref["sig"]).max()
        m.close()


def test_plain_c_host_runs():
    """examples/c_host/j2_batch.c: the constitutive update driven from C through include/dxmat.h only; exits 0 when
    four load steps match the closed-form radial return to 1e-10."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples", "c_host")], check=True, capture_output=True)
    r = subprocess.run([os.path.join(root, "examples", "c_host", "j2_batch"), "300001"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "plastic points = 300001" in r.stdout and "0 entries differ" in r.stdout      # ... and dxm_integrate_rows == dxm_integrate


@pytest.mark.parametrize("law,n", [("j2", 100), ("j2", 50_001), ("j2", 2_300_000), ("fefp", 40_001)])
def test_pageable_output_arrays_are_filled_through_the_staging_path(law, n):
    """A C caller may hand ordinary (pageable) arrays to dxm_integrate / dxm_isv_host / dxm_get_state: the GPU never
    writes into them directly (page-locked staging + CPU copy, dxmat.hip::download_to_host); same numbers as the
    page-locked route of the Python layer, for the unpacked, the packed and the chunked (two staging rounds) cases."""
    from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_F, SIGU_F, B_F, fefp_path, j2_history

    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if law == "j2":
        mk = lambda: JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN)), lazy_isv=False)   # noqa: E731
        g, nf, ng, nisv = j2_history(n, seed=2)[2], 6, 6, 7
    else:
        mk = lambda: JAXMaterial(jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F)), lazy_isv=False)   # noqa: E731
        g, nf, ng, nisv = fefp_path(n, nsteps=3, eps=3e-2)[2], 9, 9, 7
    a, b = mk(), mk()
    a.set_data_manager(n)
    b.set_data_manager(n)
    fa, ia, ca = a.integrate(g)
    lib = _lib.load()
    flux, isv, ct = np.full((n, nf), np.nan), np.full((n, nisv), np.nan), np.full((n, nf * ng), np.nan)   # plain numpy memory
    st = _lib.Stats()
    rc = lib.dxm_integrate(b._handle, np.ascontiguousarray(g).ctypes.data, 0.0, flux.ctypes.data, isv.ctypes.data, ct.ctypes.data, C.byref(st))
    assert rc == 0 and st.as_dict() == a.last_stats
    assert np.array_equal(flux, fa) and np.array_equal(ct.reshape(np.asarray(ca).shape), ca) and np.array_equal(isv, np.asarray(ia))
    isv2 = np.full((n, nisv), np.nan)
    assert lib.dxm_isv_host(b._handle, 1, isv2.ctypes.data) == 0 and np.array_equal(isv2, isv)
    p = np.full((n, 1), np.nan)
    assert lib.dxm_get_state(b._handle, 1, 0, p.ctypes.data) == 0 and np.array_equal(p[:, 0], isv[:, 0])




def test_isv_outputs_bound_at_the_c_level_are_written_by_the_host_buffer_call():
    """`dxm_bind_isv_output`: page-locked (npoints, dim) rows per field receive that field of the final state inside
    `dxm_integrate`'s chunk pipeline -- bit-identical to `dxm_get_state` afterwards, for a chunked batch and a single-chunk one;
    pageable memory is refused (the rows are written by DMA); NULL unbinds."""
    from helpers import E, NU, j2_history

    lib = _lib.load()
    for n in (300_007, 1000):
        prm = (C.c_double * 5)(E, NU, 350.0, 500.0, 1e3)
        h = lib.dxm_create(_lib.LAW_J2_VOCE, prm, 5, n, 0)
        assert h
        hist = j2_history(n, seed=1, sig0=350.0)
        bufs = {name: _lib.PinnedArray(shape) for name, shape in (("g", (n, 6)), ("f", (n, 6)), ("c", (n, 36)), ("p", (n, 1)), ("ep", (n, 6)))}
        g, f, c, p, ep = (bufs[k].array for k in ("g", "f", "c", "p", "ep"))
        pageable = np.zeros((n, 6))
        assert lib.dxm_bind_isv_output(h, 1, pageable.ctypes.data_as(C.c_void_p)) < 0 and b"page-locked" in lib.dxm_last_error()
        assert lib.dxm_bind_isv_output(h, 2, ep.ctypes.data_as(C.c_void_p)) < 0        # two fields: p, epsp
        _lib.check(lib.dxm_bind_isv_output(h, 0, p.ctypes.data_as(C.c_void_p)), lib)
        _lib.check(lib.dxm_bind_isv_output(h, 1, ep.ctypes.data_as(C.c_void_p)), lib)
        st = _lib.Stats()
        for k in range(3):
            g[...] = hist[k]
            p[...] = -1.0
            ep[...] = -1.0
            assert lib.dxm_integrate(h, g.ctypes.data_as(C.c_void_p), 0.0, f.ctypes.data_as(C.c_void_p), None, c.ctypes.data_as(C.c_void_p), C.byref(st)) == 0
            want_p, want_ep = np.empty((n, 1)), np.empty((n, 6))
            _lib.check(lib.dxm_get_state(h, _lib.S1, 0, want_p.ctypes.data_as(C.c_void_p)), lib)
            _lib.check(lib.dxm_get_state(h, _lib.S1, 1, want_ep.ctypes.data_as(C.c_void_p)), lib)
            assert np.array_equal(p, want_p) and np.array_equal(ep, want_ep) and (k == 0 or want_p.any())
            _lib.check(lib.dxm_advance(h), lib)
        _lib.check(lib.dxm_bind_isv_output(h, 0, None), lib)
        p[...] = -1.0
        g[...] = hist[3]
        assert lib.dxm_integrate(h, g.ctypes.data_as(C.c_void_p), 0.0, f.ctypes.data_as(C.c_void_p), None, c.ctypes.data_as(C.c_void_p), C.byref(st)) == 0
        assert (p == -1.0).all() and not (ep == -1.0).any()
        lib.dxm_destroy(h)
        for b in bufs.values():
            b.release()
