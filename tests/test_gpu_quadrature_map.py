"""``quadrature_map.AcceleratedUpdate`` with the engine behind it (``field_map.QuadratureFieldMap`` = the mixin over the
dolfinx-free stand-in base): after every operation the flux / tangent / internal-state fields must be BIT-identical to
what the reference's update cadence (``bench.as_reference_update``: ``quadrature_map.py:297-360`` + ``utils.py:136-143``)
leaves in the same fields with a second ``HIPMaterial`` behind it, and within 1e-12 of the same cadence with the oracle
material.  Maps over all cells deliver straight into the fields' memory (bound, page-locked in place), maps over a subset
scatter rows; sizes on both sides of the packed-transfer threshold (32 768 points)."""
import numpy as np
import pytest

import dolfinx_materials_amd.materials as jm
from bench import as_reference_advance, as_reference_update
from dolfinx_materials_amd.field_map import FieldMapBase, QuadratureFieldMap
from dolfinx_materials_amd.hip_material import LazyISV
from dolfinx_materials_amd.jaxmat import JAXMaterial
from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, SIG0_F, SIGU_F, B_F, fefp_path, j2_history
from oracle import constitutive_np as onp
from oracle_material import OracleJ2Material

pytestmark = pytest.mark.gpu


def _behavior(law):
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if law == "j2_linear":
        return jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))
    if law == "j2_voce":
        return jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V))
    return jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F))


def _fields(q):
    return {**{k: f.x.array for k, f in q.fluxes.items()}, **{k: f.x.array for k, f in q.internal_state_variables.items()},
            "jacobian": q.jacobian_flatten.x.array}


@pytest.mark.parametrize("law,ncell,nqp,subset", [
    ("j2_linear", 37, 4, False), ("j2_linear", 37, 4, True), ("j2_voce", 5001, 8, False), ("j2_voce", 9001, 8, True),
    ("fefp", 23, 4, False), ("fefp", 4601, 8, True)])
def test_accelerated_map_is_bit_identical_to_the_reference_cadence(law, ncell, nqp, subset):
    n = ncell * nqp
    rng = np.random.default_rng(11)
    cells = np.sort(rng.choice(ncell, size=(ncell * 9) // 10, replace=False)).astype(np.int32) if subset else None
    if law == "fefp":
        hist, gname, ng = fefp_path(n, nsteps=6, eps=3e-2)[::2], "F", 9
    else:
        hist, gname, ng = j2_history(n, seed=3, sig0=SIG0_V if law == "j2_voce" else SIG0_LIN), "strain", 6
    now = {"g": hist[0]}
    ev = lambda c: now["g"].reshape(ncell, nqp, ng)[c].reshape(-1, ng)   # noqa: E731
    fast = QuadratureFieldMap(ncell, nqp, JAXMaterial(_behavior(law)), cells=cells)
    slow = FieldMapBase(ncell, nqp, JAXMaterial(_behavior(law)), cells=cells)
    fast.isv_every_update = True
    for q in (fast, slow):
        q.register_gradient(gname, ev)
    if law == "fefp":   # first call at F = I, as the reference demos do (finite_strain_elastoplasticity.py:181-184)
        now["g"] = np.tile(np.array([1.0, 1, 1, 0, 0, 0, 0, 0, 0]), (n, 1))
        fast.update()
        as_reference_update(slow)
    for k, g in enumerate(hist):
        now["g"] = g
        for rep in range(2):
            fast.update()
            as_reference_update(slow)
            for name in _fields(slow):
                assert np.array_equal(_fields(fast)[name], _fields(slow)[name]), (k, rep, name)
        fast.advance()
        as_reference_advance(slow)
        for name in _fields(slow):
            assert np.array_equal(_fields(fast)[name], _fields(slow)[name]), (k, "advance", name)
    assert fast._bound == (not subset)
    # default mode: the state fields were written by the engine inside every update -- into the bound Functions of a map over all
    # cells, into the rows `dofs` of the Functions of a map over a subset (bind_state_outputs(rows=True)) -- never by a second pass
    assert fast.material.delivers_state_outputs == frozenset(fast.material.internal_state_variables)
    fast.close()
    assert fast.material.delivers_state_outputs == frozenset()
    fast.material.close()
    slow.material.close()


@pytest.mark.parametrize("subset", [False, True])
def test_accelerated_map_matches_the_oracle_material_behind_the_reference_cadence(subset):
    ncell, nqp = 4200, 8          # 33 600 points: the packed transfer of the bound path
    n = ncell * nqp
    cells = np.arange(0, ncell, 2, dtype=np.int32) if subset else None
    hist = j2_history(n, seed=8, sig0=SIG0_V)
    now = {"g": hist[0]}
    ev = lambda c: now["g"].reshape(ncell, nqp, 6)[c].reshape(-1, 6)   # noqa: E731
    fast = QuadratureFieldMap(ncell, nqp, JAXMaterial(_behavior("j2_voce")), cells=cells)
    ref = FieldMapBase(ncell, nqp, OracleJ2Material(E, NU, onp.VoceHardening(SIG0_V, SIGU_V, B_V)), cells=cells)
    for q in (fast, ref):
        q.register_gradient("strain", ev)
    scale = {"stress": SIG0_V, "jacobian": E, "p": 1e-2, "epsp": 1e-2}
    for g in hist:
        now["g"] = g
        fast.update()
        as_reference_update(ref)
        for name in ("stress", "jacobian"):
            assert np.abs(_fields(fast)[name] - _fields(ref)[name]).max() <= 1e-12 * scale[name], name
        fast.advance()
        as_reference_advance(ref)
        for name in _fields(ref):
            assert np.abs(_fields(fast)[name] - _fields(ref)[name]).max() <= 1e-12 * scale[name], name
    fast.close()
    fast.material.close()


def test_bound_map_delivers_into_the_fields_memory_and_fetches_isvs_when_looked_at():
    ncell, nqp = 5000, 8
    n = ncell * nqp
    hist = j2_history(n, seed=2)
    now = {"g": hist[1]}
    m = JAXMaterial(_behavior("j2_linear"))
    q = QuadratureFieldMap(ncell, nqp, m)
    q.isv_every_update = "lazy"   # opt-in (the default writes the ISV Functions in every update like the reference)
    q.register_gradient("strain", lambda c: now["g"].reshape(ncell, nqp, 6)[c].reshape(-1, 6))
    q.update()
    assert q._bound and set(m._bound) == {"flux", "tangent", "gradient", "isv:p", "isv:epsp"}
    assert m._bound["flux"].ctypes.data == q.fluxes["stress"].x.array.ctypes.data
    assert m._bound["tangent"].ctypes.data == q.jacobian_flatten.x.array.ctypes.data
    assert m._bound["gradient"].ctypes.data == q.gradients["strain"].function.x.array.ctypes.data
    assert isinstance(q._last_isv, LazyISV) and not q._last_isv.fetched        # nothing looked at the ISVs
    held = q._isv_functions()["p"]                                             # (no refresh through this accessor)
    assert not held.x.array.any() and m.last_stats["n_plastic"] > 0
    # the first look at the dict downloads them straight into the Functions' page-locked memory (isv_every_update = "lazy")
    ref1 = onp.j2_update(hist[1], np.zeros((n, 6)), np.zeros(n), E, NU, onp.LinearHardening(SIG0_LIN, H_LIN))
    assert np.abs(q.internal_state_variables["p"].x.array - ref1["p"]).max() < 1e-14 and held.x.array.any()
    assert np.abs(q.variables["epsp"].values - ref1["epsp"]).max() < 1e-14
    assert not q._last_isv.fetched                                             # ... not through a staged (N, 7) array
    now["g"] = hist[2]
    q.update()
    assert q.__dict__["_accel_isv_stale"] and np.abs(held.x.array - ref1["p"]).max() < 1e-14   # stale until somebody looks / advance
    q.advance()
    ref = onp.j2_update(hist[2], np.zeros((n, 6)), np.zeros(n), E, NU, onp.LinearHardening(SIG0_LIN, H_LIN))
    assert np.abs(q.internal_state_variables["p"].x.array - ref["p"]).max() < 1e-14
    assert np.abs(q.internal_state_variables["epsp"].values - ref["epsp"]).max() < 1e-14
    assert np.abs(q.fluxes["stress"].values - ref["sig"]).max() < 1e-12 * SIG0_LIN
    # the s0 mirrors are the material's own copies: the next update overwrites the bound arrays, not them
    s0 = m.get_initial_state_dict()
    now["g"] = hist[3]
    q.update()
    assert np.array_equal(s0["strain"], hist[2]) and np.abs(s0["stress"] - ref["sig"]).max() < 1e-12 * SIG0_LIN
    q.close()
    assert not m._bound
    m.close()


def test_device_gradient_through_the_map_equals_the_host_gradient_route():
    from dolfinx_materials_amd.gradient import Hex8Mesh

    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from hex_fem import HexMesh

    mesh = HexMesh(6)
    rng = np.random.default_rng(5)
    u = 2e-3 * rng.standard_normal(mesh.ndof)
    maps = []
    for device in (False, True):
        q = QuadratureFieldMap(mesh.num_cells, mesh.nqp, JAXMaterial(_behavior("j2_linear")))
        q.register_gradient("strain", lambda cells: mesh.strain(u, cells))
        if device:
            q.register_device_gradient(Hex8Mesh(mesh.coords, mesh.conn), lambda: u)
        q.update()
        q.advance()
        maps.append(q)
    for name in _fields(maps[0]):
        a, b = _fields(maps[0])[name], _fields(maps[1])[name]
        assert np.abs(a - b).max() <= 1e-11 * max(np.abs(a).max(), 1e-300), name
    with pytest.raises(ValueError):
        QuadratureFieldMap(mesh.num_cells, mesh.nqp, JAXMaterial(_behavior("j2_linear")), cells=np.array([0, 1], dtype=np.int32)
                           ).register_device_gradient(Hex8Mesh(mesh.coords, mesh.conn), lambda: u)
    for q in maps:
        q.close()
        q.material.close()


@pytest.mark.parametrize("layout", ["full", "pack4", "sym"])
def test_two_materials_on_disjoint_cells_with_device_gradients_fill_the_same_functions(layout):
    """A multi-material problem (demos/multimaterials/multimaterials.py:253-257: one QuadratureMap per material, each over its
    cells): both maps evaluate their strain on the GPU from the one displacement vector (a mesh object per map, connectivity
    restricted to its cells) and the engine stores stress and tangent blocks in the rows of that map -- against the same two
    maps with host-evaluated strains, and against their restated reference cadence.  ``layout``: the rows forms move the rows of
    a packed tangent layout as they are (``dxm_integrate_rows`` / ``dxm_integrate_displacement_rows`` with a 4- / 21-wide Function)."""
    from dolfinx_materials_amd.gradient import Hex8Mesh

    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from hex_fem import HexMesh

    mesh = HexMesh(7)
    rng = np.random.default_rng(6)
    u = {"now": 2e-3 * rng.standard_normal(mesh.ndof)}
    cells_a = np.sort(rng.choice(mesh.num_cells, size=mesh.num_cells // 3, replace=False)).astype(np.int32)
    cells_b = np.setdiff1d(np.arange(mesh.num_cells, dtype=np.int32), cells_a)
    laws = {"a": ("j2_voce", cells_a), "b": ("j2_linear", cells_b)}
    fields = {}
    for route in ("host", "device", "cadence"):
        maps = []
        for law, cells in laws.values():
            cls = FieldMapBase if route == "cadence" else QuadratureFieldMap
            q = cls(mesh.num_cells, mesh.nqp, JAXMaterial(_behavior(law), tangent_layout=layout), cells=cells)
            q.register_gradient("strain", lambda c: mesh.strain(u["now"], c))
            if route != "cadence":
                assert q._accel_plan().row_outputs
            if route == "device":
                q.register_device_gradient(Hex8Mesh(mesh.coords, mesh.conn[cells]), lambda: u["now"])
                assert q._accel_plan().row_outputs and not q._accel_plan().identity
            maps.append(q)
        total = {}
        for scale in (1.0, 1.7):
            u_now = u["now"]
            u["now"] = scale * u_now
            for q in maps:
                (as_reference_update(q) if route == "cadence" else q.update())
                (as_reference_advance(q) if route == "cadence" else q.advance())
            u["now"] = u_now
        for name in _fields(maps[0]):   # the two maps own disjoint rows: the sum of their fields is the mesh-wide field
            total[name] = _fields(maps[0])[name] + _fields(maps[1])[name]
        fields[route] = total
        for q in maps:
            if hasattr(q, "close"):
                q.close()
            q.material.close()
    for name in fields["host"]:
        assert np.array_equal(fields["host"][name], fields["cadence"][name]), name
        scale = max(np.abs(fields["host"][name]).max(), 1e-300)
        assert np.abs(fields["device"][name] - fields["host"][name]).max() <= 1e-11 * scale, name
    assert np.abs(fields["host"]["p"]).max() > 0 and (fields["host"]["stress"].reshape(mesh.num_cells, -1) != 0).any(axis=1).all()


@pytest.mark.parametrize("case", ["full", "subset"])
def test_engine_behind_the_map_reproduces_the_reference_classs_fields(case):
    """``tests/golden/quadrature_map_ref.npz``: what the REFERENCE's own ``QuadratureMap.update() / advance()`` left in
    its quadrature Functions (generated from ``/root/reference`` over numpy-backed doubles of dolfinx, with the oracle law
    behind it); the accelerated map with the HIP kernels behind it must land every field in the same place, to 1e-12."""
    import os

    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quadrature_map_ref.npz"))
    ncell, nqp = int(gold["ncell"]), int(gold["nqp"])
    cells = gold["subset"] if case == "subset" else None
    beh = jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=float(gold["E"]), nu=float(gold["nu"])),
                                        jm.VoceHardening(float(gold["sig0"]), float(gold["sigu"]), float(gold["b"])))
    now = {"k": 0}
    q = QuadratureFieldMap(ncell, nqp, JAXMaterial(beh), cells=cells)
    q.isv_every_update = True
    q.register_gradient("strain", lambda c: gold["strains"][now["k"]].reshape(ncell, nqp * 6)[c])
    scale = {"stress": float(gold["sig0"]), "jacobian": float(gold["E"]), "p": 1e-2, "epsp": 1e-2}
    for i, (op, k) in enumerate(zip(gold["ops"], gold["strain_of_op"])):
        if op == "update":
            now["k"] = int(k)
            q.update()
        else:
            q.advance()
        for name in scale:
            assert np.abs(_fields(q)[name] - gold[f"{case}_{i}_{name}"]).max() <= 1e-12 * scale[name], (case, i, op, name)
    q.close()
    q.material.close()


def test_close_gives_back_only_what_the_map_bound():
    """`close()` un-page-locks the arrays THIS map bound; a binding the caller made on the same material stays."""
    ncell, nqp = 300, 8
    n = ncell * nqp
    cells = np.arange(0, ncell, 2, dtype=np.int32)
    m = JAXMaterial(_behavior("j2_linear"))
    q = QuadratureFieldMap(ncell, nqp, m, cells=cells)          # a subset map binds its ISV Functions (row deliveries), nothing else
    npts = len(cells) * nqp
    mine = np.zeros(npts * 6)
    m.bind_inputs(gradient=mine)                                # the caller's own page-locked gradient buffer
    eps = j2_history(npts, seed=5)[2]
    q.register_gradient("strain", lambda c: eps.reshape(len(cells), nqp, 6))
    q.update()
    assert set(m._bound) == {"gradient", "isv:p", "isv:epsp"} and m.delivers_state_outputs == {"p", "epsp"}
    q.close()
    assert set(m._bound) == {"gradient"} and m._bound["gradient"] is mine
    # ... and a map over everything gives back its five, not the sixth
    m2 = JAXMaterial(_behavior("j2_linear"))
    q2 = QuadratureFieldMap(ncell, nqp, m2)
    e2 = j2_history(n, seed=6)[2]
    q2.register_gradient("strain", lambda c: e2.reshape(ncell, nqp, 6)[c].reshape(-1, 6))
    q2.update()
    assert set(m2._bound) == {"flux", "tangent", "gradient", "isv:p", "isv:epsp"}
    q2.close()
    assert not m2._bound
    m.close()
    m2.close()


def test_update_keeps_the_references_timer_rows_with_the_engine(monkeypatch):
    import contextlib

    import dolfinx_materials_amd.hip_material as hm
    import dolfinx_materials_amd.quadrature_map as qm

    seen = []

    @contextlib.contextmanager
    def recorder(name):
        seen.append(name)
        yield

    monkeypatch.setattr(qm, "_Timer", recorder)
    monkeypatch.setattr(hm, "_Timer", recorder)
    ncell, nqp = 64, 8
    eps = j2_history(ncell * nqp, seed=1)[2]
    m = JAXMaterial(_behavior("j2_voce"))
    q = QuadratureFieldMap(ncell, nqp, m)
    q.register_gradient("strain", lambda c: eps.reshape(ncell, nqp, 6)[c].reshape(-1, 6))
    q.update()
    q.update()
    dx = [s for s in seen if s.startswith("dx_mat:")]
    assert dx == 2 * ["dx_mat: External state variable update", "dx_mat: Gradients evaluation", "dx_mat: Material integration",
                      "dx_mat: Update values and tangent operators"]
    # the material's own rows nest inside "Material integration" (jaxmat.py:209-223)
    i0, i1 = seen.index("dx_mat: Material integration"), seen.index("dx_mat: Update values and tangent operators")
    assert [s for s in seen[i0:i1] if s.startswith("jaxmat:")] == ["jaxmat: dolfinx to jaxmat conversion", "jaxmat: First pass (includes jit compilation)",
                                                                    "jaxmat: jaxmat to dolfinx conversion"]
    q.close()
    m.close()


@pytest.mark.parametrize("law,ncell,nqp,subset", [("j2_linear", 5000, 8, False), ("j2_voce", 601, 4, False), ("fefp", 4600, 8, False),
                                                  ("j2_linear", 9000, 8, True), ("fefp", 4600, 8, True)])
def test_default_mode_writes_the_isv_functions_inside_the_update_itself(law, ncell, nqp, subset):
    """`isv_every_update = True` (the default: the reference writes the ISV Functions in every update, quadrature_map.py:332) with
    the engine and a map over all cells: the fields of the final state travel inside `integrate`'s own transfer pipeline into the
    Functions' page-locked memory (`dxm_bind_isv_output`) -- no second pass over the state (`read_final_state` is not called by
    `update()`), a Function object taken out of the dict earlier is current when `update()` returns, and the values are the bits
    the lazy mode downloads.  Switching the mode switches the delivery.  ``subset``: a map over every other cell -- the Functions over
    all cells are bound as ROW destinations and ``integrate_rows`` writes the fields of point i into row ``dofs[i]``."""
    n = ncell * nqp
    cells = np.arange(0, ncell, 2, dtype=np.int32) if subset else None
    if law == "fefp":
        hist, gname, ng = fefp_path(n, nsteps=6, eps=3e-2)[::2], "F", 9
    else:
        hist, gname, ng = j2_history(n, seed=8, sig0=SIG0_V if law == "j2_voce" else SIG0_LIN), "strain", 6
    now = {"g": hist[0]}
    ev = lambda c: now["g"].reshape(ncell, nqp, ng)[c].reshape(-1, ng)   # noqa: E731
    maps = {}
    for mode in (True, "lazy"):
        q = QuadratureFieldMap(ncell, nqp, JAXMaterial(_behavior(law)), cells=cells)
        q.isv_every_update = mode
        q.register_gradient(gname, ev)
        maps[mode] = q
    fast, lazy = maps[True], maps["lazy"]
    assert QuadratureFieldMap.isv_every_update is True
    if law == "fefp":
        now["g"] = np.tile(np.array([1.0, 1, 1, 0, 0, 0, 0, 0, 0]), (n, 1))
        for q in maps.values():
            q.update()
    assert fast._accel_plan().identity == (not subset)
    reads = []
    inner = fast.material.read_final_state
    fast.material.read_final_state = lambda name, out: (reads.append(name), inner(name, out))[1]
    held = dict(fast._isv_functions())                       # Function objects taken out before any update
    names = set(fast.material.internal_state_variables)
    for k, g in enumerate(hist):
        now["g"] = g
        fast.update()
        assert fast.material.delivers_state_outputs == names and not reads and not fast.__dict__["_accel_isv_stale"]
        lazy.update()
        assert lazy.material.delivers_state_outputs == frozenset()
        for name in names:
            assert np.array_equal(held[name].x.array, lazy.internal_state_variables[name].x.array), (k, name)
        assert held["p"].x.array.any() or k == 0
        if k % 2:
            for q in maps.values():
                q.advance()
            assert reads and set(reads) >= names             # advance() writes the final state as the reference does (:350-360)
            del reads[:]
            for name, f in _fields(fast).items():
                assert np.array_equal(f, _fields(lazy)[name]), name
    # opting in to "lazy" on the same map stops the deliveries; back to True resumes them
    fast.isv_every_update = "lazy"
    now["g"] = hist[-1] * 0.5 if law != "fefp" else hist[-2]
    before = {name: held[name].x.array.copy() for name in names}
    fast.update()
    assert fast.material.delivers_state_outputs == frozenset() and fast.__dict__["_accel_isv_stale"]
    assert all(np.array_equal(held[name].x.array, before[name]) for name in names)      # nobody has looked yet
    lazy.update()
    assert all(np.array_equal(fast.internal_state_variables[name].x.array, lazy.internal_state_variables[name].x.array) for name in names)
    fast.isv_every_update = True
    fast.update()
    assert fast.material.delivers_state_outputs == names and not fast.__dict__["_accel_isv_stale"]
    for q in maps.values():
        q.close()
        q.material.close()
    assert fast.material.delivers_state_outputs == frozenset()


@pytest.mark.parametrize("law,layout,ncell,nqp,subset", [
    ("j2_linear", "sym", 37, 4, False), ("j2_voce", "sym", 5001, 8, True), ("elastic", "sym", 4200, 8, False),
    ("j2_linear", "coef", 4200, 8, False), ("j2_voce", "coef", 37, 4, True),
    ("j2_linear", "pack4", 37, 4, True), ("j2_voce", "pack4", 5001, 8, False), ("j2_linear", "pack4", 40_000, 8, False)])
def test_jacobians_of_a_packed_map_evaluate_to_the_block_of_the_full_map(law, layout, ncell, nqp, subset):
    """The assembly-side consumer: a map whose material hands the tangent out packed holds a ``jacobian_flatten``
    of 21 / 9 / 4 doubles per point and ``jacobians[block]`` in terms of it (and of the stress field for ``"pack4"``) -- the
    expression ``derivative()`` contracts.  Evaluated at the quadrature points it must be the 6x6 block of the full-layout map:
    bit for bit for ``"sym"`` (same kernel arithmetic, 21 of the 36 entries stored) and for ``"coef"`` / ``"pack4"`` to the rounding of
    the one fused multiply-add the kernel uses per entry (1e-15 of the block's scale); flux and state fields are identical bit
    for bit.  320 000 points: many chunks on two streams with the 80 B/point download and no host rebuild."""
    n = ncell * nqp
    rng = np.random.default_rng(5)
    cells = np.sort(rng.choice(ncell, size=(ncell * 9) // 10, replace=False)).astype(np.int32) if subset else None
    sig0 = SIG0_V if law == "j2_voce" else SIG0_LIN
    hist = j2_history(n, seed=3, sig0=sig0)
    now = {"g": hist[0]}
    ev = lambda c: now["g"].reshape(ncell, nqp, 6)[c].reshape(-1, 6)   # noqa: E731
    behavior = (lambda: jm.ElasticBehavior(jm.LinearElasticIsotropic(E=E, nu=NU))) if law == "elastic" else (lambda: _behavior(law))
    full = QuadratureFieldMap(ncell, nqp, JAXMaterial(behavior()), cells=cells)
    packed = QuadratureFieldMap(ncell, nqp, JAXMaterial(behavior(), tangent_layout=layout), cells=cells)
    width = {"sym": 21, "coef": 9, "pack4": 4}[layout]
    assert packed.jacobian_flatten.x.array.size == n * width and packed.material.tangent_size == width
    for q in (full, packed):
        q.register_gradient("strain", ev)
    rows = full.dofs
    plastic = 0
    for k, g in enumerate(hist):
        now["g"] = g
        for q in (full, packed):
            q.update()
        for name in _fields(full):
            if name != "jacobian":
                assert np.array_equal(_fields(packed)[name], _fields(full)[name]), (k, name)
        want = full.tangent_block_values(rows=rows)
        assert np.array_equal(want, full.jacobian_flatten.x.array.reshape(-1, 6, 6)[rows])
        got = packed.tangent_block_values(rows=rows)
        if layout == "sym":
            assert np.array_equal(got, want), k
        else:
            assert np.abs(got - want).max() <= 1e-15 * np.abs(want).max(), (k, np.abs(got - want).max())
        plastic += full.material.last_stats["n_plastic"]
        assert packed.material.last_stats == full.material.last_stats
        for q in (full, packed):
            q.advance()
        for name in _fields(full):
            if name != "jacobian":
                assert np.array_equal(_fields(packed)[name], _fields(full)[name]), (k, "advance", name)
    assert plastic > 0 or law == "elastic"
    assert packed._bound == (not subset)     # the packed Function's memory IS the material's output array for a map over all cells
    for q in (full, packed):
        q.close()
        q.material.close()
