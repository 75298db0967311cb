"""The caller of the hot path, pinned by the reference's own code.

``tests/golden/quadrature_map_ref.npz`` holds the quadrature fields that the REFERENCE's ``QuadratureMap`` leaves after a
sequence of ``update()`` / ``advance()`` calls (generated in the build container by
``tests/golden/make_quadrature_map_golden.py``: the reference's ``quadrature_map.py`` / ``quadrature_function.py`` /
``utils.py`` run unmodified over numpy-backed doubles of what they call into dolfinx, ``oracle/dolfinx_doubles.py``).

* everywhere: ``field_map.QuadratureFieldMap`` (= ``quadrature_map.AcceleratedUpdate`` over the stand-in base) with the same
  oracle material replays the sequence and must reproduce the fixture bit for bit -- maps over all cells and over a subset;
* where ``/root/reference`` exists: ``quadrature_map.accelerate(reference.QuadratureMap)`` -- the class a dolfinx user gets --
  is built over the real class and must agree with the real class after every operation, also with a material that offers
  the engine's optional members (bound outputs, page-locked inputs, status record)."""
import os

import numpy as np
import pytest

from dolfinx_materials_amd.field_map import QuadratureFieldMap
from dolfinx_materials_amd.quadrature_map import AcceleratedUpdate, accelerate
from oracle import constitutive_np as onp
from oracle.ref_import import REFERENCE_ROOT, reference_available
from oracle_material import OracleJ2Material

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quadrature_map_ref.npz"))
NCELL, NQP = int(GOLD["ncell"]), int(GOLD["nqp"])
FIELDS = ("stress", "jacobian", "p", "epsp")


def _material():
    return OracleJ2Material(float(GOLD["E"]), float(GOLD["nu"]), onp.VoceHardening(float(GOLD["sig0"]), float(GOLD["sigu"]), float(GOLD["b"])))


def _fields(q):
    return {"stress": q.fluxes["stress"].x.array, "jacobian": q.jacobian_flatten.x.array, "p": q.internal_state_variables["p"].x.array,
            "epsp": q.internal_state_variables["epsp"].x.array}


@pytest.mark.parametrize("isv_mode", [True, "lazy"])
@pytest.mark.parametrize("case", ["full", "subset"])
def test_field_map_reproduces_what_the_reference_class_left_in_its_functions(case, isv_mode):
    cells = GOLD["subset"] if case == "subset" else None
    now = {"k": 0}
    q = QuadratureFieldMap(NCELL, NQP, _material(), cells=cells)
    # True: the reference writes the internal state variables in every update (quadrature_map.py:332); "lazy" (opt-in): they are
    # refreshed when `_fields` below looks at `q.internal_state_variables[...]` -- the same content either way
    q.isv_every_update = isv_mode
    q.register_gradient("strain", lambda c: GOLD["strains"][now["k"]].reshape(NCELL, NQP * 6)[c])
    for i, (op, k) in enumerate(zip(GOLD["ops"], GOLD["strain_of_op"])):
        if op == "update":
            now["k"] = int(k)
            q.update()
        else:
            q.advance()
        for name in FIELDS:
            assert np.array_equal(_fields(q)[name], GOLD[f"{case}_{i}_{name}"]), (case, i, op, name)


@pytest.mark.skipif(not reference_available(), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("isv_mode", [True, "lazy"])
@pytest.mark.parametrize("case,engine_like", [("full", False), ("subset", False), ("full", True), ("subset", True), ("subset", "rows"), ("full", "rows")])
def test_accelerated_class_over_the_real_reference_class_equals_the_real_class(case, engine_like, isv_mode):
    from oracle import dolfinx_doubles as dd
    from test_quadrature_map import EngineLikeMaterial, RowDeliveringMaterial

    cells = GOLD["subset"] if case == "subset" else None
    with dd.installed(REFERENCE_ROOT) as qm:
        Accelerated = accelerate(qm.QuadratureMap)
        assert issubclass(Accelerated, qm.QuadratureMap) and Accelerated.update is AcceleratedUpdate.update
        assert Accelerated.advance is AcceleratedUpdate.advance and Accelerated.initialize_state is AcceleratedUpdate.initialize_state
        assert Accelerated.register_gradient is qm.QuadratureMap.register_gradient and Accelerated.derivative is qm.QuadratureMap.derivative
        now = {"k": 0}
        maps = []
        for cls in (qm.QuadratureMap, Accelerated):
            mat = _material()
            if engine_like and cls is Accelerated:
                mat = (RowDeliveringMaterial if engine_like == "rows" else EngineLikeMaterial)(float(GOLD["E"]), float(GOLD["nu"]), onp.VoceHardening(float(GOLD["sig0"]), float(GOLD["sigu"]), float(GOLD["b"])))
            q = cls(dd.Mesh(NCELL, "hexahedron", 3), 2, mat, cells=cells)
            q.register_gradient("strain", dd.PointwiseExpression(lambda c: GOLD["strains"][now["k"]].reshape(NCELL, NQP * 6)[c], 6))
            maps.append(q)
        ref, acc = maps
        acc.isv_every_update = isv_mode   # "lazy": the reference's own `variables` / dict accesses trigger the refresh
        for i, (op, k) in enumerate(zip(GOLD["ops"], GOLD["strain_of_op"])):
            for q in maps:
                if op == "update":
                    now["k"] = int(k)
                    q.update()
                else:
                    q.advance()
            for name in FIELDS:
                assert np.array_equal(_fields(acc)[name], _fields(ref)[name]), (i, op, name)
                assert np.array_equal(_fields(ref)[name], GOLD[f"{case}_{i}_{name}"])   # the fixture is what the real class does
        if engine_like == "rows" and case == "subset":   # stress and tangent went into the rows `dofs` of the real class's Functions
            assert "integrate_rows" in acc.material.calls and "bind_outputs" not in acc.material.calls
        if engine_like and case == "full":
            assert acc._accel_plan().bound and "bind_outputs" in acc.material.calls and "bind_inputs" in acc.material.calls
            # the gradient went straight into the reference's QuadratureExpression Function (Expression.eval(values=...))
            assert np.array_equal(acc.gradients["strain"].function.x.array, GOLD["strains"][int(GOLD["strain_of_op"][-2])].ravel())
        # update_initial_state stays the reference's method and feeds the accelerated update
        for q in maps:
            q.update_initial_state("p", 1e-3)
            now["k"] = 2
            q.update()
        assert np.array_equal(_fields(acc)["stress"], _fields(ref)["stress"]) and np.array_equal(_fields(acc)["jacobian"], _fields(ref)["jacobian"])
        acc.close()


# ---- the assembly-side consumer of the packed tangent (quadrature_map.py:83-105, :132-158) ----------
def _packed_material(layout):
    from oracle_material import PackedOracleJ2Material

    return PackedOracleJ2Material(float(GOLD["E"]), float(GOLD["nu"]), onp.VoceHardening(float(GOLD["sig0"]), float(GOLD["sigu"]), float(GOLD["b"])), layout)


def _replay(q, now, check):
    for i, (op, k) in enumerate(zip(GOLD["ops"], GOLD["strain_of_op"])):
        if op == "update":
            now["k"] = int(k)
            q.update()
        else:
            q.advance()
        check(i, op)


@pytest.mark.parametrize("layout", ["sym", "coef", "pack4"])
@pytest.mark.parametrize("case", ["full", "subset"])
def test_field_map_jacobians_of_a_packed_layout_evaluate_to_the_full_block_of_the_fixture(case, layout):
    """``QuadratureFieldMap.jacobians[block]`` -- ``tangent_entries`` over the packed ``jacobian_flatten`` (and the stress Field for
    ``"pack4"``) -- evaluated at the quadrature points against the 36-wide Function the REFERENCE class left (the committed
    fixture): exactly for the index-only ``"sym"``, to rounding for the two coefficient forms.  Stress and state Functions are
    the fixture's bit for bit: the layout changes the tangent Function only."""
    cells = GOLD["subset"] if case == "subset" else None
    now = {"k": 0}
    q = QuadratureFieldMap(NCELL, NQP, _packed_material(layout), cells=cells)
    assert q.jacobian_flatten.x.array.size == NCELL * NQP * {"sym": 21, "coef": 9, "pack4": 4}[layout]
    q.register_gradient("strain", lambda c: GOLD["strains"][now["k"]].reshape(NCELL, NQP * 6)[c])
    seen_plastic = []

    def check(i, op):
        for name in ("stress", "p", "epsp"):
            assert np.array_equal(_fields(q)[name], GOLD[f"{case}_{i}_{name}"]), (case, i, op, name)
        if op != "update":
            return
        rows = q.dofs
        want = GOLD[f"{case}_{i}_jacobian"].reshape(-1, 6, 6)[rows]
        got = q.tangent_block_values(rows=rows)
        if layout == "sym":   # index-only: the upper triangle is the fixture's, the lower one its mirror (the oracle's own lower
            # triangle differs from it in the last bit: (c3 n_i) n_j is not (c3 n_j) n_i)
            iu = np.triu_indices(6)
            assert np.array_equal(got[:, iu[0], iu[1]], want[:, iu[0], iu[1]]) and np.array_equal(got, got.transpose(0, 2, 1))
        else:
            assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()
        seen_plastic.append(int(q.material.last_stats["n_plastic"]))

    _replay(q, now, check)
    assert max(seen_plastic) > 0   # the n x n term took part


@pytest.mark.skipif(not reference_available(), reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("layout", ["sym", "coef", "pack4"])
@pytest.mark.parametrize("case", ["full", "subset"])
def test_accelerated_class_builds_the_reference_jacobians_over_a_packed_tangent_function(case, layout):
    """``accelerate(reference.QuadratureMap)`` with a packed-tangent material: the constructor re-creates ``WJ`` /
    ``jacobian_flatten`` at the packed width and ``jacobians[block]`` as the UFL matrix of ``tangent_entries``; evaluated per point
    (the doubles' expression tree reads the Functions' memory like a compiled form) it equals the 6x6 block that the REAL class
    with the full-layout material holds in its 36-wide Function after every update -- while flux and state Functions are
    identical bit for bit.  ``derivative`` stays the reference's method."""
    from oracle import dolfinx_doubles as dd

    cells = GOLD["subset"] if case == "subset" else None
    with dd.installed(REFERENCE_ROOT) as qm:
        Accelerated = accelerate(qm.QuadratureMap)
        assert Accelerated.derivative is qm.QuadratureMap.derivative
        now = {"k": 0}
        ref = qm.QuadratureMap(dd.Mesh(NCELL, "hexahedron", 3), 2, _material(), cells=cells)
        acc = Accelerated(dd.Mesh(NCELL, "hexahedron", 3), 2, _packed_material(layout), cells=cells)
        width = {"sym": 21, "coef": 9, "pack4": 4}[layout]
        assert acc.jacobian_flatten.x.array.size == NCELL * NQP * width and acc.WJ.value_size == width
        assert set(acc.jacobians) == {("stress", "strain")} and acc.jacobians[("stress", "strain")].ufl_shape == (6, 6)
        for q in (ref, acc):
            q.register_gradient("strain", dd.PointwiseExpression(lambda c: GOLD["strains"][now["k"]].reshape(NCELL, NQP * 6)[c], 6))
        rows = ref.dofs
        for i, (op, k) in enumerate(zip(GOLD["ops"], GOLD["strain_of_op"])):
            for q in (ref, acc):
                if op == "update":
                    now["k"] = int(k)
                    q.update()
                else:
                    q.advance()
            for name in ("stress", "p", "epsp"):
                assert np.array_equal(_fields(acc)[name], _fields(ref)[name]), (i, op, name)
            # the reference's own jacobians[block] over its 36-wide Function is, per point, that Function's row reshaped (6, 6)
            full = ref.jacobians[("stress", "strain")].evaluate()[rows]
            assert np.array_equal(full, ref.jacobian_flatten.x.array.reshape(-1, 6, 6)[rows])
            got = acc.jacobians[("stress", "strain")].evaluate()[rows]
            if layout == "sym":
                iu = np.triu_indices(6)
                assert np.array_equal(got[:, iu[0], iu[1]], full[:, iu[0], iu[1]]) and np.array_equal(got, got.transpose(0, 2, 1)), (i, op)
            else:
                assert np.abs(got - full).max() <= 1e-13 * np.abs(full).max(), (i, op)
        acc.close()


def test_a_packed_layout_is_refused_where_it_has_no_meaning():
    from dolfinx_materials_amd.quadrature_map import sym_position, tangent_entries

    with pytest.raises(ValueError):
        tangent_entries("pack4", np.zeros(4))            # no stress to take the direction from
    with pytest.raises(ValueError):
        tangent_entries("coef", np.zeros(9), n=9)        # the coefficient forms are those of the 6x6 J2 block
    assert [sym_position(i, j) for i in range(6) for j in range(i, 6)] == list(range(21))
    assert all(sym_position(i, j) == sym_position(j, i) for i in range(6) for j in range(6))
