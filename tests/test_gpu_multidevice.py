"""One process, several GPUs: ``HIPMaterial(behavior, devices=[...])`` cuts the points into contiguous blocks, one
``dxm_material`` handle per block, and every host-buffer call runs the blocks side by side, each GPU's DMA delivering into
its rows of the one host array (north_star: stress / tangent reassembled in the host memory of the one process that owns
the quadrature Functions, ``quadrature_map.py:66-70``; no collective).  The box has one GPU, so the blocks here live on
``devices=[0, 0]`` / ``[0, 0, 0]``: two or three handles, threads, pipelines and block offsets -- everything except the
second PCIe link.  The results must be bit-identical to the single-handle material through the whole state life cycle."""
import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd._lib import DxmError
from dolfinx_materials_amd.field_map import QuadratureFieldMap
from dolfinx_materials_amd.jaxmat import JAXMaterial
from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, SIG0_F, SIGU_F, B_F, fefp_path, j2_history

pytestmark = pytest.mark.gpu


def _beh(law):
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if law == "linear":
        return jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))
    if law == "voce":
        return jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V))
    return jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F))


@pytest.mark.parametrize("law,n,devices", [("linear", 1001, [0, 0]), ("voce", 70_001, [0, 0]), ("voce", 100_003, [0, 0, 0]),
                                           ("fefp", 66_001, [0, 0]), ("linear", 1, [0, 0]), ("voce", 0, [0, 0])])
def test_blocks_on_several_handles_equal_one_handle_bit_for_bit(law, n, devices):
    hist = fefp_path(n, nsteps=6, eps=3e-2)[::2] if law == "fefp" else j2_history(n, seed=12, sig0=SIG0_V if law == "voce" else SIG0_LIN)
    one, many = JAXMaterial(_beh(law)), JAXMaterial(_beh(law), devices=devices)
    for m in (one, many):
        m.set_data_manager(n)
    assert len(many._parts) == len(devices) and [p[1] for p in many._parts][0] == 0 and many._parts[-1][2] == n
    for k, g in enumerate(hist):
        fa, ia, ca = one.integrate(g)
        fb, ib, cb = many.integrate(g)
        assert np.array_equal(fa, fb) and np.array_equal(ca, cb) and np.array_equal(np.asarray(ia), np.asarray(ib)), k
        assert one.last_stats == many.last_stats
        if g.nbytes >= (1 << 20):   # the one gradient array is page-locked once for all blocks, not staged for every second one
            assert many.last_upload == "dma (caller's array page-locked)"
            assert many._lib.dxm_host_register(g.ctypes.data, g.nbytes) == 0 and many._lib.dxm_host_unregister(g.ctypes.data) == 0   # and released
        if k == 1:   # a second Newton iteration from the same initial state, then back to it and forward again
            for m in (one, many):
                m.data_manager.revert()
            sa, sb = one.get_final_state_dict(), many.get_final_state_dict()
            assert all(np.array_equal(sa[key], sb[key]) for key in sa)
            fa, _, ca = one.integrate(g)
            fb, _, cb = many.integrate(g)
            assert np.array_equal(fa, fb) and np.array_equal(ca, cb)
        for m in (one, many):
            m.data_manager.update()
        sa, sb = one.get_initial_state_dict(), many.get_initial_state_dict()
        assert set(sa) == set(sb) and all(np.array_equal(sa[key], sb[key]) for key in sa)
    one.close()
    many.close()


def test_state_written_by_the_caller_reaches_the_right_block():
    n = 5001
    rng = np.random.default_rng(3)
    eps = j2_history(n, seed=4)[2]
    p0 = rng.uniform(0, 2e-3, (n, 1))
    ep0 = 1e-4 * rng.standard_normal((n, 6))
    ep0[:, :3] -= ep0[:, :3].mean(axis=1)[:, None]
    one, many = JAXMaterial(_beh("linear")), JAXMaterial(_beh("linear"), devices=[0, 0, 0])
    for m in (one, many):
        m.set_data_manager(n)
        m.set_initial_state_dict({"p": p0, "epsp": ep0})
        m.update_material_property("yield_stress.sig0", 200.0)
        m.set_newton(30, 1e-13)
        m.set_option("max_chunks", 4)
    sb = many.get_initial_state_dict()
    assert np.array_equal(sb["p"], p0) and np.array_equal(sb["epsp"], ep0)
    fa, ia, ca = one.integrate(eps)
    fb, ib, cb = many.integrate(eps)
    assert np.array_equal(fa, fb) and np.array_equal(ca, cb) and np.array_equal(np.asarray(ia), np.asarray(ib))
    out = np.empty((n, 6))
    assert np.array_equal(many.read_final_state("epsp", out), np.asarray(ia)[:, 1:])
    one.close()
    many.close()


def test_field_map_on_a_multi_device_material_binds_and_matches():
    ncell, nqp = 9001, 8
    n = ncell * nqp
    hist = j2_history(n, seed=6)
    now = {"g": hist[0]}
    maps = [QuadratureFieldMap(ncell, nqp, JAXMaterial(_beh("linear"), devices=d)) for d in ([0], [0, 0])]
    for q in maps:
        q.register_gradient("strain", lambda c: now["g"].reshape(ncell, nqp, 6)[c].reshape(-1, 6))
    for g in hist[:3]:
        now["g"] = g
        for q in maps:
            q.update()
        assert np.array_equal(maps[0].fluxes["stress"].x.array, maps[1].fluxes["stress"].x.array)
        assert np.array_equal(maps[0].jacobian_flatten.x.array, maps[1].jacobian_flatten.x.array)
        for q in maps:
            q.advance()
        assert np.array_equal(maps[0].internal_state_variables["epsp"].x.array, maps[1].internal_state_variables["epsp"].x.array)
    assert maps[1]._bound and set(maps[1].material._bound) == {"flux", "tangent", "gradient", "isv:p", "isv:epsp"}
    for q in maps:
        q.close()
        q.material.close()


def test_calls_that_belong_to_one_gpu_say_so():
    m = JAXMaterial(_beh("linear"), devices=[0, 0])
    m.set_data_manager(100)
    for call in (lambda: m.integrate_device(0, 0, 0), lambda: m.launch_generation,
                 lambda: m.stats(), lambda: m.isv_device(1, 0)):
        with pytest.raises(DxmError, match="one GPU"):
            call()
    with pytest.raises(DxmError):
        JAXMaterial(_beh("linear"), devices=[0, 99]).set_data_manager(10)
    with pytest.raises(ValueError):
        JAXMaterial(_beh("linear"), devices=[])
    m.close()
