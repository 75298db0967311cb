"""GPU: random SEQUENCES of protocol operations (integrate / update / revert / state dicts in and out / lazy ISVs /
bound output arrays / several integrates from one s0) against a model that keeps s0 and s1 in numpy and computes with
the C oracle -- the life-cycle of generic.py:176-216 / jaxmat.py:30-43 / quadrature_map.py:281-360 exercised in orders
no hand-written test uses (pointer-swap advance with the s1 alias, state set between integrates, revert after advance)."""
import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from dolfinx_materials_amd.jaxmat import JAXMaterial
from oracle import oracle_c

from helpers import E, NU, SIG0_V, SIGU_V, B_V

pytestmark = pytest.mark.gpu
from helpers import to_device, to_host  # noqa: E402,F401

TOL = 1e-11


class Model:
    """What the reference's DataManager holds: s0 and s1, each (p, epsp, stress)."""

    def __init__(self, n):
        self.n = n
        self.sig0 = SIG0_V
        self.s0 = dict(p=np.zeros(n), epsp=np.zeros((n, 6)), stress=np.zeros((n, 6)), strain=np.zeros((n, 6)))
        self.s1 = {k: v.copy() for k, v in self.s0.items()}

    def integrate(self, eps):
        r = oracle_c.j2(eps, self.s0["epsp"], self.s0["p"], E, NU, 1, self.sig0, SIGU_V, B_V)
        self.s1 = dict(p=r["p"].copy(), epsp=r["epsp"].copy(), stress=r["sig"].copy(), strain=np.array(eps))
        return r

    def update(self):
        self.s0 = {k: v.copy() for k, v in self.s1.items()}

    def revert(self):
        self.s1 = {k: v.copy() for k, v in self.s0.items()}


def ref_plastic_mismatch(ct, ct_ref):
    """Points where the two sides took different branches (only legitimate ON the yield surface)."""
    return np.abs(ct - ct_ref).reshape(len(ct), -1).max(axis=1) > 1e-6 * np.abs(ct_ref).max()


def close(a, b, scale):
    if np.asarray(b).size == 0:
        return True
    return np.abs(np.asarray(a).reshape(np.asarray(b).shape) - b).max() <= TOL * scale


@pytest.mark.parametrize("seed,n,bound,lazy", [(0, 777, False, True), (1, 5000, True, True), (2, 64, False, False),
                                               (3, 40_000, True, False), (4, 1, False, True), (5, 70_001, False, True),
                                               (6, 3000, "io", True), (7, 66_000, "io", True), (8, 130, "io", False),
                                               (9, 2500, "rows", True), (10, 70_003, "rows", True), (11, 1, "rows", True),
                                               (12, 3100, "rows_isv", True), (13, 70_009, "rows_isv", True)])
def test_random_operation_sequences_match_the_state_model(seed, n, bound, lazy):
    run_operation_sequence(seed, n, bound, lazy)


def run_operation_sequence(seed, n, bound, lazy, device_ops=True, tol=TOL, nops=64, trace=None):
    """The sequence itself (also run on the CPU against the test double of the library, tests/test_protocol_fuzz_cpu.py, where the
    device-pointer launches become host-buffer ones).  ``trace``: a list that receives, after EVERY operation, what the library
    says about its handle -- ``dxm_io_held`` of s0 and s1, ``dxm_launch_generation`` -- and every array the operation returned:
    the same seed against the test double and against libdxmat.so must leave the same trace
    (``test_the_test_double_and_the_library_leave_the_same_trace``)."""
    rng = np.random.default_rng(seed)
    beh = jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(SIG0_V, SIGU_V, B_V))
    m = JAXMaterial(beh, lazy_isv=lazy)
    m.set_data_manager(n)
    isv_rows = None
    if bound in ("rows", "rows_isv"):   # a map over a subset of the cells: results into rows `rows` of arrays over all cells (integrate_rows)
        total = n + 37
        rows = np.ascontiguousarray(np.random.default_rng(seed + 100).permutation(total)[:n], dtype=np.int64)
        flux_all, jac_all = np.full((total, 6), 9.0), np.full((total, 36), 9.0)
        others = np.setdiff1d(np.arange(total), rows)
        if bound == "rows_isv":   # ... and the ISV Functions over all cells as row destinations (what the accelerated map binds by default)
            isv_rows = {"p": np.full(total, 9.0), "epsp": np.full(total * 6, 9.0)}
            m.bind_state_outputs(isv_rows, deliver=True, rows=True)
            bound = "rows"
    elif bound:
        flux_fn, jac_fn = np.zeros(n * 6), np.zeros(n * 36)
        m.bind_outputs(flux=flux_fn, tangent=jac_fn)
    if bound == "io":   # the gradient Function too (what the accelerated QuadratureMap binds): every update overwrites it, the
        grad_fn = np.zeros(n * 6)   # s0 mirrors of strain and stress are views of the device copies dxm_advance keeps
        m.bind_inputs(gradient=grad_fn)
    model = Model(n)
    mu = E / 2 / (1 + NU)
    ey = SIG0_V / (2 * mu) * np.sqrt(2.0 / 3.0)
    eps = np.zeros((n, 6))
    held = None           # (isv object of an earlier integrate, what it must show when looked at)
    ops = rng.choice(["integrate", "integrate", "integrate", "update", "update", "revert", "get", "set", "look", "device", "device", "prop",
                      "hold", "hold", "drop"], size=nops)
    # state dictionaries taken at random steps and KEPT (in a list, in a dict under their id(), inside a closure): each must go on
    # showing the state it was taken from whatever happens to the material afterwards (the reference's dictionaries are copies,
    # generic.py:265-277) -- lazy_rows.LazyInitialRows views, copies of the alternating flux buffers
    handouts, by_id, closures = [], {}, []

    def keep(got, want, how):
        if how == 0:
            handouts.append((got, want))
        elif how == 1:
            by_id[id(got)] = (got, want)
        else:
            closures.append(lambda got=got, want=want: (got, want))

    def kept():
        return handouts + list(by_id.values()) + [c() for c in closures]

    def check_kept():
        for got, want in kept():
            for name, ref_rows in want.items():
                assert np.array_equal(np.asarray(got[name]).reshape(ref_rows.shape), ref_rows), name
    known = {"initial": False, "final": False}   # whether the host-side stress mirror of s0 / s1 is meaningful
    if device_ops:
        import torch

        dev = torch.device("cuda:0")
        d_flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
        d_ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    sig0_now = SIG0_V
    for op in ops:
        if op == "device" and not device_ops:
            op = "integrate"
        returned = {}
        if op == "integrate":
            d = rng.standard_normal((n, 6))
            eps = 0.6 * eps + d * (rng.uniform(0, 3.0, n) * ey / np.linalg.norm(d, axis=1))[:, None]
            if bound == "io":
                grad_fn[...] = eps.ravel()
                sig, isv, ct = m.integrate(grad_fn.reshape(n, 6))
            elif bound == "rows":
                isv = m.integrate_rows(eps, rows, flux_all, jac_all)
                sig, ct = flux_all[rows], jac_all[rows]
                assert (flux_all[others] == 9.0).all() and (jac_all[others] == 9.0).all()
                if isv_rows is not None:   # the state fields of this very call, in their rows; nobody else's rows touched
                    r_now = oracle_c.j2(eps, model.s0["epsp"], model.s0["p"], E, NU, 1, model.sig0, SIGU_V, B_V)
                    assert close(isv_rows["p"][rows], r_now["p"], max(r_now["p"].max(), 1e-300) + 1e-30)
                    assert close(isv_rows["epsp"].reshape(total, 6)[rows], r_now["epsp"], max(np.abs(r_now["epsp"]).max(), 1e-300) + 1e-30)
                    assert (isv_rows["p"][others] == 9.0).all() and (isv_rows["epsp"].reshape(total, 6)[others] == 9.0).all()
            else:
                sig, isv, ct = m.integrate(eps)
            ref = model.integrate(eps)
            scale = max(np.abs(ref["sig"]).max(), SIG0_V)
            assert close(sig, ref["sig"], scale) and close(ct, ref["Ct"], np.abs(ref["Ct"]).max())
            if bound and bound != "rows":
                assert close(flux_fn, ref["sig"].ravel(), scale) and close(jac_fn, ref["Ct"].ravel(), np.abs(ref["Ct"]).max())
            assert m.last_stats["n_plastic"] == ref["n_plastic"] and m.last_stats["n_nan"] == 0
            held = isv
            known["final"] = True
            returned = {"sig": np.array(sig), "ct": np.array(ct).reshape(n, 36), "n_plastic": np.array(m.last_stats["n_plastic"])}
            if not lazy:
                returned["isv"] = np.array(np.asarray(isv))
        elif op == "device":
            # device-pointer form: asynchronous launch on torch's stream
            d = rng.standard_normal((n, 6))
            eps = 0.6 * eps + d * (rng.uniform(0, 3.0, n) * ey / np.linalg.norm(d, axis=1))[:, None]
            d_eps = to_device(eps)
            st = torch.cuda.current_stream().cuda_stream
            m.integrate_device(d_eps.data_ptr(), d_flux.data_ptr(), d_ct.data_ptr(), st)
            torch.cuda.synchronize()
            ref = model.integrate(eps)
            scale = max(np.abs(ref["sig"]).max(), SIG0_V)
            assert close(to_host(d_flux), ref["sig"], scale) and close(to_host(d_ct), ref["Ct"].reshape(n, 36), np.abs(ref["Ct"]).max())
            assert m.stats()[1]["n_plastic"] == ref["n_plastic"]
            known["final"] = False      # the host-side stress mirror does not follow the device-pointer forms
            held = None
        elif op == "prop":
            # QuadratureMap.update_material_properties hands 0-d arrays (quadrature_map.py:160-172)
            sig0_now = float(rng.uniform(0.8, 1.2)) * SIG0_V
            m.update_material_property("yield_stress.sig0", np.asarray(sig0_now))
            model.sig0 = sig0_now
        elif op == "update":
            m.data_manager.update()
            model.update()
            known["initial"] = known["final"]
        elif op == "revert":
            m.data_manager.revert()
            model.revert()
            known["final"] = known["initial"]
        elif op == "get":
            for which, st in (("initial", model.s0), ("final", model.s1)):
                got = m.get_initial_state_dict() if which == "initial" else m.get_final_state_dict()
                assert close(got["p"], st["p"][:, None], max(st["p"].max(), 1e-300) + 1e-30), which
                assert close(got["epsp"], st["epsp"], max(np.abs(st["epsp"]).max(), 1e-300) + 1e-30), which
                assert np.asarray(got["stress"]).shape == (n, 6) and np.asarray(got["strain"]).shape == (n, 6)   # whatever they hold, they can be looked at
                returned[which + "_p"], returned[which + "_epsp"] = np.array(np.asarray(got["p"])), np.array(np.asarray(got["epsp"]))
                if known[which]:
                    assert close(got["stress"], st["stress"], max(np.abs(st["stress"]).max(), SIG0_V)), which
                    assert np.array_equal(np.asarray(got["strain"]), st["strain"]), which
                    returned[which + "_stress"], returned[which + "_strain"] = np.array(np.asarray(got["stress"])), np.array(np.asarray(got["strain"]))
        elif op == "set":
            # a consistent plastic state: p >= 0 and a deviatoric plastic strain
            p = rng.uniform(0, 2e-3, n)
            ep = rng.standard_normal((n, 6)) * 1e-3
            ep[:, :3] -= ep[:, :3].mean(axis=1, keepdims=True)
            m.set_initial_state_dict({"p": p, "epsp": ep})
            model.s0["p"], model.s0["epsp"] = p.copy(), ep.copy()
        elif op == "hold":
            which = "initial" if rng.random() < 0.7 else "final"
            getter = m.get_initial_state_dict if which == "initial" else m.get_final_state_dict
            names = ["p", "epsp"] + (["stress", "strain"] if known[which] else [])
            got, twin = getter(), getter()          # two dictionaries of one state: one kept, one looked at now and thrown away
            want = {k: np.array(np.asarray(twin[k])) for k in names}
            del twin
            st = model.s0 if which == "initial" else model.s1
            for k in names:
                assert close(want[k], st[k].reshape(want[k].shape), max(np.abs(st[k]).max(), SIG0_V if k == "stress" else 1e-300) + 1e-30), (which, k)
            if rng.random() < 0.5:                  # some kept dictionaries are looked at when taken, some only later
                assert all(np.array_equal(np.asarray(got[k]), want[k]) for k in names)
            keep({k: got[k] for k in names}, want, int(rng.integers(0, 3)))
        elif op == "drop":
            for bag in (handouts, closures):
                if bag and rng.random() < 0.5:
                    bag.pop(int(rng.integers(0, len(bag))))
            if by_id and rng.random() < 0.5:
                by_id.pop(next(iter(by_id)))
        elif op == "look" and held is not None and lazy:
            # a lazy ISV array is a VIEW of s1 as it is NOW (lazy_rows.LazyISV); the eager one is a snapshot
            a = np.asarray(held)
            returned["looked_isv"] = np.array(a)
            assert a.shape == (n, 7)
            assert close(a[:, 0], model.s1["p"], max(model.s1["p"].max(), 1e-300) + 1e-30)
            assert close(a[:, 1:], model.s1["epsp"], max(np.abs(model.s1["epsp"]).max(), 1e-300) + 1e-30)
        if op in ("update", "integrate", "set", "revert"):
            check_kept()
        if trace is not None:
            h = m._handles()[0]
            trace.append({"op": str(op), "held": (int(m._lib.dxm_io_held(h, 0)), int(m._lib.dxm_io_held(h, 1))),
                          "generation": int(m._lib.dxm_launch_generation(h)), "arrays": returned})
    check_kept()
    m.close()


@pytest.mark.parametrize("seed,n,bound,lazy", [(0, 77, False, True), (1, 500, True, True), (2, 64, False, False), (3, 400, True, False),
                                               (6, 300, "io", True), (7, 660, "io", True), (8, 130, "io", False), (9, 250, "rows", True),
                                               (10, 703, "rows", True), (21, 90, "io", True), (22, 90, "rows", True), (23, 90, True, True),
                                               (12, 310, "rows_isv", True), (25, 90, "rows_isv", True)])
def test_the_test_double_and_the_library_leave_the_same_trace(monkeypatch, seed, n, bound, lazy):
    """``tests/fake_dxmat.py`` restates the handle semantics of ``csrc/dxmat.hip`` in Python so that the layer above the C ABI is
    fuzzed on every CPU run (``tests/test_protocol_fuzz_cpu.py``, these very seeds).  Here the SAME seed runs against the double and
    against libdxmat.so, and the two traces are compared operation by operation: which copies of gradient / flux each state holds
    on the device (``dxm_io_held``), the launch generation, and every array an operation returned (to the 1e-11 at which the C
    oracle behind the double and the kernels agree).  A drift between the double and the C code fails here."""
    from dolfinx_materials_amd import _lib
    from fake_dxmat import FakeDxmat

    real_trace, fake_trace = [], []
    run_operation_sequence(seed, n, bound, lazy, device_ops=False, nops=96 if seed > 20 else 64, trace=real_trace)
    real_load = _lib.load
    fake = FakeDxmat(real_load())
    with monkeypatch.context() as mp:
        mp.setattr(_lib, "load", lambda *a, **k: fake)
        run_operation_sequence(seed, n, bound, lazy, device_ops=False, nops=96 if seed > 20 else 64, trace=fake_trace)
    assert len(real_trace) == len(fake_trace) > 0
    for k, (a, b) in enumerate(zip(real_trace, fake_trace)):
        assert a["op"] == b["op"]
        assert a["held"] == b["held"], (k, a["op"], a["held"], b["held"])
        assert a["generation"] == b["generation"], (k, a["op"], a["generation"], b["generation"])
        assert set(a["arrays"]) == set(b["arrays"]), (k, a["op"])
        for name, x in a["arrays"].items():
            y = b["arrays"][name]
            assert x.shape == y.shape, (k, a["op"], name)
            if name == "n_plastic" or name.endswith("_strain"):
                assert np.array_equal(x, y), (k, a["op"], name)
                continue
            floor = SIG0_V if name == "sig" or name.endswith("_stress") else 1e-300      # (state variables: relative to their own size)
            assert np.abs(x - y).max(initial=0.0) <= TOL * max(np.abs(y).max(initial=0.0), floor), (k, a["op"], name)


@pytest.mark.parametrize("seed,n", [(10, 500), (11, 64), (12, 3001)])
def test_random_operation_sequences_fefp(seed, n):
    """The same for the finite-strain law, whose user-visible state (p, be_bar) is not what the kernel keeps: the hidden
    isochoric Cp^-1 follows from (F_n, be_bar_n), so `set_initial_state_dict` with `be_bar` and / or `F` goes through
    `conventions.cp_bar_inv_from_be_bar` with the s0 gradient mirror (hip_material.py)."""
    from helpers import SIG0_F, SIGU_F, B_F
    from oracle import constitutive_np as onp

    rng = np.random.default_rng(seed)
    m = JAXMaterial(jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(SIG0_F, SIGU_F, B_F)))
    m.set_data_manager(n)
    ident6 = np.zeros((n, 6))
    ident6[:, :3] = 1.0
    ident9 = np.zeros((n, 9))
    ident9[:, :3] = 1.0
    s0 = dict(p=np.zeros(n), be_bar=ident6.copy(), cpinv=ident6.copy(), F=ident9.copy())
    s1 = {k: v.copy() for k, v in s0.items()}
    A = rng.standard_normal((n, 3, 3)) * 0.03
    t = 0.0
    for op in rng.choice(["integrate", "integrate", "update", "revert", "get", "set_be", "set_F_be"], size=30):
        if op == "integrate":
            t = float(np.clip(t + rng.uniform(-0.3, 0.6), -1.0, 2.0))
            F9 = onp.tensor_to_nsym(np.eye(3)[None] + t * A)
            P, isv, Ct = m.integrate(F9)
            ref = oracle_c.fefp(F9, s0["cpinv"], s0["p"], E, NU, SIG0_F, SIGU_F, B_F, kind=1)
            assert ref["n_not_converged"] == 0 and m.last_stats["n_nan"] == 0
            # re-integrating the deformation a state was advanced with puts the trial state ON the yield surface:
            # either branch is right there (same exclusion as tests/test_gpu_parity.py)
            ftr = onp.fefp_update(F9, s0["cpinv"], s0["p"], E, NU, onp.VoceHardening(SIG0_F, SIGU_F, B_F), tangent=False)["f_trial"]
            safe = np.abs(ftr) > 1e-9 * SIG0_F
            if not safe.any():
                safe[:] = ~ref_plastic_mismatch(np.asarray(Ct), ref["Ct"])
            assert close(np.asarray(P)[safe], ref["P"][safe], max(np.abs(ref["P"]).max(), SIG0_F))
            assert close(np.asarray(Ct)[safe], ref["Ct"][safe], np.abs(ref["Ct"]).max() * 10)
            s1 = dict(p=ref["p"].copy(), be_bar=ref["be_bar"].copy(), cpinv=ref["cpinv"].copy(), F=F9.copy())
        elif op == "update":
            m.data_manager.update()
            s0 = {k: v.copy() for k, v in s1.items()}
        elif op == "revert":
            m.data_manager.revert()
            s1 = {k: v.copy() for k, v in s0.items()}
        elif op == "get":
            for got, st in ((m.get_initial_state_dict(), s0), (m.get_final_state_dict(), s1)):
                assert close(got["p"], st["p"][:, None], max(st["p"].max(), 1e-300) + 1e-30)
                assert close(got["be_bar"], st["be_bar"], 1.0) and close(got["F"], st["F"], 1.0)
        else:
            # a consistent visible state: be_bar symmetric positive definite with unit determinant, p >= 0
            B = np.eye(3)[None] + 0.02 * rng.standard_normal((n, 3, 3))
            be = B @ B.transpose(0, 2, 1)
            be /= np.cbrt(np.linalg.det(be))[:, None, None]
            be6 = onp.tensor_to_mandel(be)
            p = rng.uniform(0, 5e-3, n)
            state = {"p": p, "be_bar": be6}
            if op == "set_F_be":
                Fn = onp.tensor_to_nsym(np.eye(3)[None] + 0.02 * rng.standard_normal((n, 3, 3)))
                state["F"] = Fn
                s0["F"] = Fn.copy()
            m.set_initial_state_dict(state)
            s0["p"], s0["be_bar"] = p.copy(), be6.copy()
            s0["cpinv"] = onp.cpinv_from_be_bar(s0["F"], be6)
    m.close()


@pytest.mark.parametrize("seed,subset", [(20, False), (21, True), (22, True), (23, False)])
def test_field_map_driven_by_the_engine_equals_field_map_driven_by_the_oracle(seed, subset):
    """Differential test one level up: the SAME random sequence of QuadratureFieldMap operations (update, advance,
    update_initial_state with numbers / rows / current content, refresh of the ISV fields, several updates per
    increment) once with HIPMaterial behind the map and once with the oracle-backed material of tests/oracle_material.py;
    every field must agree after every operation (maps over all cells bind the engine's output arrays, maps over a
    subset scatter rows: field_map.py)."""
    from dolfinx_materials_amd.field_map import QuadratureFieldMap
    from oracle import constitutive_np as onp
    from oracle_material import OracleJ2Material

    rng = np.random.default_rng(seed)
    ncell, nqp = int(rng.integers(3, 40)), int(rng.choice([1, 4, 8]))
    cells = np.sort(rng.choice(ncell, size=max(1, ncell // 2), replace=False)).astype(np.int32) if subset else None
    beh = jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(SIG0_V, SIGU_V, B_V))
    maps = [QuadratureFieldMap(ncell, nqp, JAXMaterial(beh), cells=cells),
            QuadratureFieldMap(ncell, nqp, OracleJ2Material(E, NU, onp.VoceHardening(SIG0_V, SIGU_V, B_V)), cells=cells)]
    mu = E / 2 / (1 + NU)
    ey = SIG0_V / (2 * mu) * np.sqrt(2.0 / 3.0)
    strain = {"all": np.zeros((ncell * nqp, 6))}
    for qm in maps:
        qm.register_gradient("strain", lambda c: strain["all"].reshape(ncell, nqp, 6)[c].reshape(-1, 6))

    def same(what):
        a, b = maps
        for name in ("stress",):
            assert close(a.fluxes[name].values, b.fluxes[name].values, SIG0_V), (what, name)
        assert close(a.jacobian_flatten.values, b.jacobian_flatten.values.reshape(a.jacobian_flatten.values.shape), E), (what, "jacobian")
        for name in ("p", "epsp"):
            fa, fb = a.internal_state_variables[name].values, b.internal_state_variables[name].values
            assert close(fa, fb, max(np.abs(fb).max(), 1e-6)), (what, name)

    for op in rng.choice(["update", "update", "advance", "init_number", "init_rows", "init_current", "refresh"], size=30):
        if op == "update":
            d = rng.standard_normal((ncell * nqp, 6))
            strain["all"] = 0.7 * strain["all"] + d * (rng.uniform(0, 2.5, ncell * nqp) * ey / np.linalg.norm(d, axis=1))[:, None]
            for qm in maps:
                qm.update()
        elif op == "advance":
            if not all(getattr(qm, "_initialized") and hasattr(qm, "_last_isv") for qm in maps):
                continue
            for qm in maps:
                qm.advance()
        elif op == "refresh":
            if not all(hasattr(qm, "_last_isv") for qm in maps):
                continue
            for qm in maps:
                qm.refresh_internal_state_variables()
        elif op == "init_number":
            v = float(rng.uniform(0, 3e-3))
            for qm in maps:
                qm.update_initial_state("p", v)
        elif op == "init_rows":
            row = rng.standard_normal(6) * 1e-3
            row[:3] -= row[:3].mean()
            for qm in maps:
                qm.update_initial_state("epsp", row)
        else:
            for qm in maps:
                qm.update_initial_state("p")
        same(op)
    maps[0].material.close()
