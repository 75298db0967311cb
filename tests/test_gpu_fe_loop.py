"""GPU: the engine inside a real global Newton loop (stand-in FE driver of examples/, the 3-D
analogue of the reference's tests/uniaxial_tension.py driver; BASELINE.json configs[4])."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_uniaxial_tension_3d_j2_closed_form_and_quadratic_convergence():
    from uniaxial_tension_3d import run

    out = run(n=6, steps=8, law="j2_linear", verbose=False)
    h = out["history"][-1]
    expect = (out["sig0"] + out["H"] * h["exx"]) / (1 + out["H"] / out["E"])
    assert abs(h["sxx"] - expect) < 1e-8 * expect and h["sxx_spread"] < 1e-6
    assert abs(h["p"] - (h["exx"] - expect / out["E"])) < 1e-10
    for step in out["history"]:
        # (an elastic first increment converges at the linear predictor: one residual evaluation, already below the tolerance)
        assert step["iters"] <= 6 and (step["norms"][-1] < 1e-6 * step["norms"][0] or step["norms"][-1] < 1e-8)


def test_uniaxial_tension_3d_fefp_runs_and_saturates():
    from uniaxial_tension_3d import run

    out = run(n=4, steps=10, law="fefp", exx_max=5e-2, verbose=False)
    for step in out["history"]:
        assert step["iters"] <= 8
    h = out["history"][-1]
    # homogeneous uniaxial Kirchhoff stress tau_xx = R(p); PK1_xx = tau_xx / F_xx
    R = 250.0 + 250.0 * (1 - np.exp(-100.0 * h["p"]))
    assert abs(h["sxx"] * (1 + h["exx"]) - R) < 1e-6 * R and h["sxx_spread"] < 1e-6


def test_device_gradient_run_equals_host_gradient_run():
    """f3 of SURVEY.md section 8: the gradient evaluated on the GPU from the displacement vector gives
    the same Newton history as the host evaluation."""
    from uniaxial_tension_3d import run

    a = run(n=4, steps=4, law="j2_linear", verbose=False)
    b = run(n=4, steps=4, law="j2_linear", verbose=False, device_gradient=True)
    for sa, sb in zip(a["history"], b["history"]):
        assert sa["iters"] == sb["iters"]
        assert abs(sa["sxx"] - sb["sxx"]) < 1e-9 * abs(sa["sxx"]) and abs(sa["p"] - sb["p"]) < 1e-12


@pytest.mark.parametrize("layout", ["sym", "coef", "pack4"])
def test_assembly_from_the_packed_tangent_layouts_gives_the_same_newton_history(layout):
    """the host assembly consumes the 21-entry upper triangle / the nine coefficients of the
    J2 tangent directly (examples/hex_fem.py); same iterates as with the full (N,6,6) block, multigrid-CG solves."""
    from uniaxial_tension_3d import run

    a = run(n=8, steps=3, law="j2_linear", verbose=False, solver="krylov")
    b = run(n=8, steps=3, law="j2_linear", verbose=False, solver="krylov", layout=layout, device_gradient=True)
    for sa, sb in zip(a["history"], b["history"]):
        assert sa["iters"] == sb["iters"] <= 8
        assert abs(sa["sxx"] - sb["sxx"]) < 1e-9 * abs(sa["sxx"]) and abs(sa["p"] - sb["p"]) < 1e-12
    h = b["history"][-1]
    expect = (b["sig0"] + b["H"] * h["exx"]) / (1 + b["H"] / b["E"])
    assert abs(h["sxx"] - expect) < 1e-8 * expect


@pytest.mark.parametrize("preconditioner,n", [("jacobi", 6), ("mg", 16)])
def test_device_resident_loop_reaches_the_closed_form(preconditioner, n):
    """The consumer that keeps displacement, stress and tangent coefficients on the GPU (examples/device_fem.py:
    matrix-free residual and tangent operator through dxm_mesh_internal_force_device / _tangent_apply_device, CG
    preconditioned by the operator's diagonal or by the matrix-free multigrid V-cycle): same closed-form answer and
    the same Newton behaviour as the host loop."""
    from uniaxial_tension_3d_device import run

    out = run(n=n, steps=4, preconditioner=preconditioner, verbose=False)
    assert out["rel_err"] < 1e-9 and out["sxx_spread"] < 1e-6 and out["other_components_max"] < 1e-6
    assert abs(out["p_mean"] - out["p_closed_form"]) < 1e-10
    for step in out["history"]:
        assert step["iters"] <= 4 and step["norms"][-1] < 1e-7      # linear predictor: at most a few corrections
    if preconditioner == "mg":
        assert out["levels"] == 2 and out["cg_iterations"] / max(out["newton_iterations"] - 4, 1) < 40


@pytest.mark.parametrize("device_gradient", [False, True])
def test_two_materials_on_disjoint_cells_in_the_newton_loop(device_gradient):
    """examples/two_materials_3d.py (the set-up of demos/multimaterials/multimaterials.py:253-257): two maps over disjoint
    cell subsets deliver into rows of the same stress / tangent fields.  With the same law on both sets the loop must
    reproduce the homogeneous closed form of the single-map run; with a stiff elastic inclusion it must converge
    quadratically and leave the inclusion elastic and more loaded than the matrix."""
    import two_materials_3d as tm

    same = tm.run(n=6, steps=4, exx_max=2e-2, same=True, device_gradient=device_gradient, verbose=False)
    assert same["delivered_into_rows"] == [True, True]
    h = same["history"][-1]
    expect = (same["sig0"] + same["H"] * h["exx"]) / (1 + same["H"] / same["E"])
    assert abs(h["sxx_matrix"] - expect) < 1e-9 * expect and abs(h["sxx_inclusion"] - expect) < 1e-9 * expect
    two = tm.run(n=6, steps=4, exx_max=1e-2, same=False, device_gradient=device_gradient, verbose=False)
    for step in two["history"]:
        n_ = step["norms"]
        assert n_[-1] < 1e-7 and len(n_) <= 8
        if len(n_) >= 3 and n_[-2] < 1e-2 * n_[0]:
            assert n_[-1] < 1e-3 * n_[-2]          # the consistent tangent from both maps: quadratic tail
    assert two["history"][-1]["sxx_inclusion"] > 1.5 * two["history"][-1]["sxx_matrix"] and two["history"][-1]["p_max"] > 0


# ---- BASELINE.json configs[4] at the sizes it names (the stand-in FE loops of examples/; dolfinx itself is not in this image) ----------
def _closed_form(out_E, sig0, H, exx):
    """Homogeneous uniaxial stress: elastic E eps below yield, then sigma_xx = (sig0 + H eps) / (1 + H / E)."""
    return min(out_E * exx, (sig0 + H * exx) / (1 + H / out_E))


def _check_host_loop(out, max_iters=3):
    for step in out["history"]:
        expect = _closed_form(out["E"], out["sig0"], out["H"], step["exx"])
        assert abs(step["sxx"] - expect) < 1e-9 * expect, step
        assert step["sxx_spread"] < 1e-6 and 1 <= step["iters"] <= max_iters and step["norms"][-1] < 1e-8, step
        assert abs(step["p"] - (step["exx"] - expect / out["E"])) < 1e-11
    return sum(step["iters"] for step in out["history"])


def test_cfg5_host_assembly_loop_at_32_cubed():
    """`tests/uniaxial_tension.py:11-118` in 3-D with HOST assembly (examples/hex_fem.py: block-CSR tangent from the nine tangent
    coefficients, multigrid-CG) and the GPU constitutive update with the strain evaluated on the device: 32^3 hexahedra =
    262 144 Gauss points, 8 load steps; closed form to 1e-9 at every step, at most 3 Newton iterations per step."""
    from uniaxial_tension_3d import run

    out = run(n=32, steps=8, law="j2_linear", verbose=False, solver="krylov", layout="coef", device_gradient=True)
    assert out["points"] == 32 ** 3 * 8 and out["ndof"] == 3 * 33 ** 3
    total = _check_host_loop(out)
    assert 8 <= total <= 20


@pytest.mark.skipif(os.environ.get("DXM_TEST_CFG5_HOST_64") != "1", reason="opt-in (DXM_TEST_CFG5_HOST_64=1): 64^3 host assembly takes ~3 minutes")
def test_cfg5_host_assembly_loop_at_64_cubed():
    from uniaxial_tension_3d import run

    out = run(n=64, steps=8, law="j2_linear", verbose=False, solver="krylov", layout="coef", device_gradient=True)
    assert out["points"] == 64 ** 3 * 8
    # (the first, elastic increment is linear: the loop's predictor lifts the boundary increment through the tangent of the state
    # the map was last updated at and the law is first evaluated at that solution.  Rounds 4-5 evaluated it at "u with only the
    # boundary nodes moved" -- 16 % strain in the first layer of cells at 64^3, a plastic return there, five Newton iterations to
    # shake the layer off; the Krylov solver's tolerance, suspected in round 5, was not the cause)
    _check_host_loop(out, max_iters=3)
    assert out["history"][0]["iters"] <= 2


def test_cfg5_device_resident_loop_at_its_stated_size_200_cubed():
    """BASELINE.json configs[4] names a 200^3 hex mesh with quad_degree 2: 8e6 cells, 6.4e7 Gauss points, 2.44e7 dofs.  The
    device-resident stand-in loop (examples/device_fem.py: constitutive update by libdxmat.so from the displacement vector,
    matrix-free residual / tangent operator of examples/libdxmfem.so, multigrid-preconditioned CG) runs it in ~10 s and ~31 GB of
    HBM: 8 load steps to eps_xx = 2 %, closed form to 1e-9, every Gauss point plastic from step 2 on, at most 3 Newton iterations
    per step (the linear predictor is exact for the homogeneous solution)."""
    import torch

    from uniaxial_tension_3d_device import run

    free_b, _total = torch.cuda.mem_get_info(0)
    if free_b < 40e9:
        pytest.skip(f"needs ~31 GB of HBM, {free_b / 1e9:.0f} GB free")
    out = run(n=200, steps=8, preconditioner="mg", verbose=False)
    assert out["points"] == 64_000_000 and out["ndof"] == 3 * 201 ** 3 and out["levels"] >= 3
    assert out["rel_err"] < 1e-9 and out["sxx_spread"] < 1e-4 and out["other_components_max"] < 1e-4
    E, sig0, H = 70e3, 250.0, 5e3
    for step in out["history"]:
        expect = _closed_form(E, sig0, H, step["exx"])
        assert abs(step["sxx"] - expect) < 1e-9 * expect, step
        assert 1 <= step["iters"] <= 3 and step["norms"][-1] < 1e-7, step
    assert 8 <= out["newton_iterations"] <= 16
    assert out["constitutive_share_of_iteration"] < 0.05       # the update is not what such a loop waits for
    assert 20 < out["hbm_GiB_allocated_peak"] < 60
