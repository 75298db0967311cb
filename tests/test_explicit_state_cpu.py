"""The explicit-state callables of the reference's material protocol -- ``sig, new_state = material.constitutive_update(eps, state, dt)``
(``jaxmat.py:158-164``, ``docs/jax.md:46-50``) and ``Ct, new_state = material.batched_constitutive_update(gradients, state, dt)``
(``jaxmat.py:147-155``, ``generic.py:115-117``) -- on ``HIPMaterial``.  The ``check_*`` functions are the test bodies: run here on CPU over
the test double of libdxmat.so (``tests/fake_dxmat.py``, small-strain laws) and on the GPU through the real library
(``tests/test_gpu_protocol.py``, FeFp included)."""
import gc
import os

import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd import _lib
from dolfinx_materials_amd.jaxmat import JAXMaterial
from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, j2_history, random_j2_state
from oracle import constitutive_np as onp

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _j2(kind="linear", **kw):
    hard = jm.LinearHardening(SIG0_LIN, H_LIN) if kind == "linear" else jm.VoceHardening(SIG0_V, SIGU_V, B_V)
    return JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), hard), **kw)


def check_recorded_calls_of_the_reference_protocol(tol_scale=1.0):
    """Every ``integrate`` the imported reference classes recorded in ``protocol_ref.npz`` (``tests/golden/make_golden.py``), replayed as
    ONE explicit-state call from the recorded initial state: same tangent, flux and state as the reference's
    ``batched_constitutive_update`` produced inside its ``integrate`` (``generic.py:176-190``) -- with a material whose own state
    was never set up for it."""
    g = np.load(os.path.join(GOLDEN, "protocol_ref.npz"))
    n = g["eps_hat"].shape[0]
    mat = _j2()
    mat.set_data_manager(3)                                  # its own batch has another size and stays as it is
    own0 = {k: np.array(v) for k, v in mat.get_initial_state_dict().items()}
    seen = 0
    for k, (op, sc) in enumerate(zip(g["script"], g["scale"])):
        if op != "integrate":
            continue
        state = {"p": g[f"s0_p_{k}"], "epsp": g[f"s0_epsp_{k}"], "stress": g[f"s0_stress_{k}"]}
        Ct, new = mat.batched_constitutive_update(sc * g["eps_hat"], state, 0.0)
        assert Ct.shape == (n, 6, 6) and set(new) == {"strain", "stress", "p", "epsp"}
        assert np.allclose(Ct, g[f"Ct_{k}"], rtol=1e-12, atol=1e-8 * tol_scale)
        assert np.allclose(new["stress"], g[f"flux_{k}"], rtol=1e-12, atol=1e-9 * tol_scale)
        assert np.allclose(np.hstack([new["p"], new["epsp"]]), g[f"isv_{k}"], rtol=1e-12, atol=1e-18)
        assert np.array_equal(new["strain"], sc * g["eps_hat"])
        seen += 1
    assert seen >= 4
    for k, v in mat.get_initial_state_dict().items():
        assert np.array_equal(np.asarray(v), own0[k]), k
    mat.close()


def check_explicit_state_update_against_the_oracle(kind, n=257):
    """Random initial states: tangent / stress / state of the oracle's ``j2_update``; the material's own s0 and s1 (mid-increment,
    with a lazily held final state) are exactly what they were; the one-point form is row i of the batched one."""
    hard = onp.LinearHardening(SIG0_LIN, H_LIN) if kind == "linear" else onp.VoceHardening(SIG0_V, SIGU_V, B_V)
    sig0 = SIG0_LIN if kind == "linear" else SIG0_V
    mat = _j2(kind)
    mat.set_data_manager(n)
    hist = j2_history(n, seed=41, sig0=sig0)
    mat.integrate(hist[1])
    mat.data_manager.update()
    f_mid = np.array(mat.integrate(hist[2])[0])               # s1 of an unfinished increment
    before = ({k: np.array(v) for k, v in mat.get_initial_state_dict().items()}, {k: np.array(v) for k, v in mat.get_final_state_dict().items()})
    epsp, p = random_j2_state(n, seed=2, sig0=sig0)
    eps = hist[2] + epsp
    ref = onp.j2_update(eps, epsp, p, E, NU, hard)
    assert ref["plastic"].any() and not ref["plastic"].all()
    Ct, new = mat.batched_constitutive_update(eps, {"epsp": epsp, "p": p[:, None]}, 0)
    safe = np.abs(ref["f_trial"]) > 1e-9 * sig0
    assert np.abs(new["stress"] - ref["sig"]).max() <= 1e-12 * sig0
    assert np.abs(new["epsp"] - ref["epsp"]).max() <= 1e-15 and np.abs(new["p"][:, 0] - ref["p"]).max() <= 1e-15
    assert np.abs(Ct[safe] - ref["Ct"][safe]).max() <= 1e-12 * np.abs(ref["Ct"]).max()
    # absent keys: the natural state (a virgin point)
    Ct0, new0 = mat.batched_constitutive_update(eps[:5], {}, 0)
    ref0 = onp.j2_update(eps[:5], np.zeros((5, 6)), np.zeros(5), E, NU, hard)
    assert np.abs(new0["stress"] - ref0["sig"]).max() <= 1e-12 * sig0 and new0["p"].shape == (5, 1)
    # one material point (jaxmat.py:158-164): flat arrays in, flat arrays out
    i = int(np.nonzero(ref["plastic"])[0][0])
    sig_i, new_i = mat.constitutive_update(eps[i], {"epsp": epsp[i], "p": p[i]}, 0.0)
    assert sig_i.shape == (6,) and np.array_equal(sig_i, new["stress"][i]) and np.array_equal(new_i["epsp"], new["epsp"][i])
    assert new_i["p"].shape == (1,) and new_i["p"][0] == new["p"][i, 0]
    # the material itself: untouched
    after = (mat.get_initial_state_dict(), mat.get_final_state_dict())
    for a, b in zip(before, after):
        for k in a:
            assert np.array_equal(a[k], np.asarray(b[k])), k
    assert np.array_equal(np.asarray(mat.get_final_state_dict()["stress"]), f_mid)
    with pytest.raises(AssertionError, match="unknown field"):
        mat.batched_constitutive_update(eps, {"no_such_field": p}, 0)
    with pytest.raises(ValueError):
        mat.batched_constitutive_update(eps[:, :5], {}, 0)
    # parameters changed through the protocol reach the explicit-state update too (quadrature_map.py:160-172)
    mat.update_material_property("yield_stress.sig0", 1.2 * sig0)
    _, new2 = mat.batched_constitutive_update(eps, {"epsp": epsp, "p": p[:, None]}, 0)
    hard2 = onp.LinearHardening(1.2 * SIG0_LIN, H_LIN) if kind == "linear" else onp.VoceHardening(1.2 * SIG0_V, SIGU_V, B_V)
    ref2 = onp.j2_update(eps, epsp, p, E, NU, hard2)
    assert np.abs(new2["stress"] - ref2["sig"]).max() <= 1e-12 * sig0
    mat.close()


@pytest.fixture
def fake(monkeypatch):
    from fake_dxmat import FakeDxmat

    lib = FakeDxmat(_lib.load())
    monkeypatch.setattr(_lib, "load", lambda *a, **k: lib)
    yield lib
    gc.collect()


def test_recorded_reference_calls_as_explicit_state_updates_on_the_test_double(fake):
    check_recorded_calls_of_the_reference_protocol()


@pytest.mark.parametrize("kind", ["linear", "voce"])
def test_explicit_state_update_on_the_test_double(fake, kind):
    check_explicit_state_update_against_the_oracle(kind)


def test_natural_state_has_the_shapes_of_the_state_dictionaries(fake):
    mat = _j2()
    st = mat.natural_state(4)
    assert {k: v.shape for k, v in st.items()} == {"strain": (4, 6), "stress": (4, 6), "p": (4, 1), "epsp": (4, 6)}
    assert all(not v.any() for v in st.values())


def test_python_materials_callable_follows_the_generic_convention(fake):
    check_python_materials_callable()


def check_python_materials_callable():
    """``python_materials.LinearElasticIsotropic.constitutive_update(eps, state, dt)`` returns ``(C, state)`` with ``state["Stress"]``
    set, like the reference's (``python_materials/elasticity.py:21-24``; the JAX materials return ``(stress, new_state)``); the batched
    form is the ``_vmap`` of it: ``(Ct (N,6,6), state of (N, dim) arrays)`` (``generic.py:115-117``)."""
    from dolfinx_materials_amd.python_materials import LinearElasticIsotropic

    mat = LinearElasticIsotropic(70e3, 0.3)
    mat.set_data_manager(2)
    eps = np.array([1e-3, -2e-4, 3e-4, 5e-4, 0.0, -1e-4])
    state = {}
    C, out = mat.constitutive_update(eps, state, 0)
    assert out is state and C.shape == (6, 6) and np.allclose(C, onp.elastic_matrix(70e3, 0.3), rtol=1e-14) and np.array_equal(mat.C, onp.elastic_matrix(70e3, 0.3))
    assert np.allclose(state["Stress"], onp.elastic_matrix(70e3, 0.3) @ eps, rtol=1e-13)
    Ct, new = mat.batched_constitutive_update(np.tile(eps, (5, 1)), {}, 0)
    assert Ct.shape == (5, 6, 6) and new["Stress"].shape == (5, 6) and np.allclose(new["Stress"][3], state["Stress"], rtol=1e-14)
    mat.close()
