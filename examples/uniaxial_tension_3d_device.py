#!/usr/bin/env python3
"""BASELINE config 5 with a device-resident consumer: 3-D uniaxial tension of the unit cube (the 3-D analogue of
``tests/uniaxial_tension.py:11-118``) where displacement, stress and tangent never leave the GPU
(``examples/device_fem.py``).  Checks itself against the closed form sigma_xx = (sig0 + H eps) / (1 + H / E).

    python examples/uniaxial_tension_3d_device.py --n 64  --steps 8
    python examples/uniaxial_tension_3d_device.py --n 200 --steps 8 --preconditioner mg        # 6.4e7 Gauss points
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))


def run(n=16, steps=8, exx_max=2e-2, preconditioner="mg", coarsest=8, verbose=True, cg_rtol=1e-10):
    import torch

    import dolfinx_materials_amd.materials as jm
    from device_fem import DeviceProblem
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    E, nu, sig0, H = 70e3, 0.3, 250.0, 5e3
    dev = torch.device("cuda", 0)
    material = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=nu), jm.LinearHardening(sig0, H)),
                           tangent_layout="coef")
    t0 = time.perf_counter()
    prob = DeviceProblem(n, material, dev, preconditioner=preconditioner, coarsest=coarsest)
    setup_s = time.perf_counter() - t0
    hist = []
    t_all = time.perf_counter()
    for k in range(1, steps + 1):
        exx = exx_max * k / steps
        norms = prob.solve_step(exx, cg_rtol=cg_rtol, log=print if verbose else None)
        sxx = prob.flux[:, 0]
        hist.append(dict(exx=exx, sxx=float(sxx.mean()), sxx_spread=float(sxx.max() - sxx.min()),
                         other_components_max=float(prob.flux[:, 1:].abs().max()), iters=len(norms), norms=norms))
        if verbose:
            print(f"step {k:2d} exx={exx:.4f} <sxx>={hist[-1]['sxx']:.6f} newton={len(norms)} |r|={norms[-1]:.2e}", flush=True)
    torch.cuda.synchronize()
    t = dict(prob.timers, total=time.perf_counter() - t_all, mesh_and_state_setup=setup_s)
    p = material.get_final_state_dict()["p"] if n <= 64 else None
    expect = (sig0 + H * exx_max) / (1 + H / E)
    its = max(t["newton_iterations"], 1)
    out = {"n": n, "points": prob.npoints, "ndof": prob.ndof, "preconditioner": preconditioner, "levels": len(prob.mg.levels) if prob.mg else 1,
           "sxx": hist[-1]["sxx"], "closed_form": expect, "rel_err": abs(hist[-1]["sxx"] - expect) / expect,
           "sxx_spread": hist[-1]["sxx_spread"], "other_components_max": hist[-1]["other_components_max"],
           "p_mean": float(p.mean()) if p is not None else None, "p_closed_form": exx_max - expect / E,
           "newton_iterations": its, "cg_iterations": t["cg_iterations"], "fine_operator_applications": t["operator_applications"],
           "seconds_per_newton_iteration": {k: round(t[k] / its, 5) for k in ("constitutive", "residual", "preconditioner_setup", "solve")},
           "constitutive_share_of_iteration": round(t["constitutive"] / max(t["constitutive"] + t["residual"] + t["preconditioner_setup"] + t["solve"], 1e-30), 5),
           "timers": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()},
           "hbm_GiB_allocated_peak": round(torch.cuda.max_memory_allocated() / 2**30, 2), "history": hist}
    material.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--exx", type=float, default=2e-2)
    ap.add_argument("--preconditioner", default="mg", choices=["mg", "jacobi"])
    ap.add_argument("--coarsest", type=int, default=8, help="smallest cells-per-edge of the multigrid hierarchy")
    ap.add_argument("--cg-rtol", type=float, default=1e-10)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    out = run(a.n, a.steps, a.exx, a.preconditioner, a.coarsest, verbose=not a.quiet, cg_rtol=a.cg_rtol)
    hist = out.pop("history")
    out["newton_per_step"] = [h["iters"] for h in hist]
    print(json.dumps(out))
