#!/usr/bin/env python3
"""3-D uniaxial tension of the unit cube with J2 plasticity: the 3-D analogue of the reference
driver ``tests/uniaxial_tension.py:11-118`` (BASELINE.json configs[4]) with the stand-in host FE
loop of ``examples/hex_fem.py`` (dolfinx is not available) and the GPU constitutive update.

    python examples/uniaxial_tension_3d.py [--n 16] [--steps 10] [--law j2_linear|fefp] [--layout full|sym|coef|pack4]
    python examples/uniaxial_tension_3d.py --n 64 --steps 8 --layout coef --device-gradient     # the config-5 stand-in (8 load steps: the
                                                                                                  # bare Newton of hex_fem.py has no line search for bigger jumps)

Symmetry planes x=0, y=0, z=0 are clamped in their normal direction and u_x is imposed on x=1, so
the solution is the homogeneous uniaxial stress state sigma_xx = R(p): a known answer the run
checks itself against.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

from hex_fem import HexMesh, newton_solve  # noqa: E402


def run(n=8, steps=10, law="j2_linear", exx_max=2e-2, verbose=True, device_gradient=False, layout="full", solver="auto"):
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from dolfinx_materials_amd.field_map import QuadratureFieldMap

    E, nu, sig0, H = 70e3, 0.3, 250.0, 5e3
    el = jm.LinearElasticIsotropic(E=E, nu=nu)
    mesh = HexMesh(n)
    u = np.zeros(mesh.ndof)
    if law == "j2_linear":
        # layout: what the host assembly consumes -- the (N,6,6) block, its 21-entry upper triangle, the nine coefficients of
        # Ct = c1 1x1 + c2 I + c3 n x n, or (c1, c2, c3, w) with n = dev(stress) w (hex_fem.HexMesh.element_matrices)
        material = JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.LinearHardening(sig0, H)), tangent_layout=layout)
        gname, fname, B = "strain", "stress", mesh.B_eps
        evaluator = lambda cells: mesh.strain(u, cells)  # noqa: E731
    else:
        material = JAXMaterial(jm.FeFpJ2Plasticity(el, jm.VoceHardening(sig0, 2 * sig0, 100.0)))
        gname, fname, B = "F", "PK1", mesh.B_grad
        evaluator = lambda cells: mesh.deformation_gradient(u, cells)  # noqa: E731
    qmap = QuadratureFieldMap(mesh.num_cells, mesh.nqp, material)
    qmap.register_gradient(gname, evaluator)
    if device_gradient:  # strain / F evaluated on the GPU from u: only u is uploaded per update
        from dolfinx_materials_amd.gradient import Hex8Mesh

        qmap.register_device_gradient(Hex8Mesh(mesh.coords, mesh.conn), lambda: u)
    # first call at u = 0, as the reference demos do (plane_elastoplasticity.py:146-149,
    # finite_strain_elastoplasticity.py:181-184): QuadratureMap.initialize_state records the
    # *current* gradient as the initial one, which must be F = I for the FeFp law
    qmap.update()

    x0, x1 = mesh.nodes_on(0, 0.0), mesh.nodes_on(0, 1.0)
    y0, z0 = mesh.nodes_on(1, 0.0), mesh.nodes_on(2, 0.0)
    bc_dofs = np.concatenate([3 * x0, 3 * x1, 3 * y0 + 1, 3 * z0 + 2])
    timers, hist = {}, []
    t_all = time.perf_counter()
    for k in range(1, steps + 1):
        exx = exx_max * k / steps
        bc_vals = np.concatenate([np.zeros(len(x0)), np.full(len(x1), exx), np.zeros(len(y0)), np.zeros(len(z0))])
        norms = newton_solve(mesh, qmap, u, bc_dofs, bc_vals, B, fname, timers=timers, solver=solver, log=print if verbose else None)
        flux = qmap.fluxes[fname].x.array.reshape(-1, qmap.fluxes[fname].dim)
        p = qmap.internal_state_variables["p"].x.array
        hist.append(dict(exx=exx, sxx=float(flux[:, 0].mean()), sxx_spread=float(np.ptp(flux[:, 0])), p=float(p.mean()), iters=len(norms), norms=norms))
        if verbose:
            print(f"step {k:2d} exx={exx:.4f} <flux_xx>={hist[-1]['sxx']:.4f} p={hist[-1]['p']:.5f} newton={len(norms)} |r|={norms[-1]:.2e}")
    timers["total"] = time.perf_counter() - t_all
    return dict(n=n, points=mesh.num_cells * 8, ndof=mesh.ndof, law=law, layout=layout, history=hist, timers=timers, E=E, nu=nu, sig0=sig0, H=H)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--law", default="j2_linear")
    ap.add_argument("--device-gradient", action="store_true")
    ap.add_argument("--layout", default="full", choices=["full", "sym", "coef", "pack4"])
    ap.add_argument("--solver", default="auto", choices=["auto", "direct", "krylov"])
    ap.add_argument("--exx", type=float, default=2e-2)
    a = ap.parse_args()
    out = run(a.n, a.steps, a.law, exx_max=a.exx, device_gradient=a.device_gradient, layout=a.layout, solver=a.solver)
    h = out["history"][-1]
    if a.law == "j2_linear":
        # homogeneous uniaxial stress: sigma_xx = sig0 + H p and eps_xx = sigma_xx / E + p
        expect = (out["sig0"] + out["H"] * h["exx"]) / (1 + out["H"] / out["E"])
        t, its = out["timers"], max(out["timers"].get("newton_iterations", 1), 1)
        print(json.dumps({"n": out["n"], "points": out["points"], "ndof": out["ndof"], "layout": out["layout"], "sxx": h["sxx"],
                          "closed_form": expect, "rel_err": abs(h["sxx"] - expect) / expect, "sxx_spread": h["sxx_spread"],
                          "newton_iterations": its,
                          "seconds_per_newton_iteration": {k: round(t.get(k, 0.0) / its, 4) for k in ("constitutive", "assembly", "solve")},
                          "timers": t}))
