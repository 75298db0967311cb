#!/usr/bin/env python3
"""Two materials on one mesh: a stiff elastic inclusion in a J2 matrix under tension -- the set-up of the reference's
``demos/multimaterials/multimaterials.py:253-257`` (one ``QuadratureMap`` per material, each over its own cells) with the stand-in
host FE loop of ``examples/hex_fem.py``.

    python examples/two_materials_3d.py [--n 16] [--steps 6] [--device-gradient] [--same]

Each map covers a SUBSET of the cells, so nothing can be bound as a whole: the engine delivers every point's stress and
tangent block into its row of the fields over all cells (``HIPMaterial.integrate_rows``), and both maps write into the SAME
flux / tangent fields here (disjoint rows), which is what the assembly reads.  ``--same`` gives both cell sets the matrix
material: the run must then reproduce the single-map solution of ``uniaxial_tension_3d.py`` (homogeneous uniaxial stress,
closed form) -- the check the script applies to itself.
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


class Maps:
    """What ``newton_solve`` needs of a ``QuadratureMap``, for several maps that share their flux / tangent fields."""

    def __init__(self, maps):
        self.maps = maps
        self.material, self.fluxes, self.jacobian_flatten = maps[0].material, maps[0].fluxes, maps[0].jacobian_flatten

    def update(self):
        for q in self.maps:
            q.update()

    def advance(self):
        for q in self.maps:
            q.advance()


def run(n=8, steps=6, exx_max=1e-2, same=False, device_gradient=False, verbose=True, solver="auto"):
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.field_map import QuadratureFieldMap
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    E, nu, sig0, H = 70e3, 0.3, 250.0, 5e3
    mesh = HexMesh(n)
    u = np.zeros(mesh.ndof)
    centres = mesh.coords[mesh.conn].mean(axis=1)
    inside = np.all(np.abs(centres - 0.5) < 0.25, axis=1)     # the inclusion: the central cube of half the edge length
    cells = {"inclusion": np.flatnonzero(inside).astype(np.int32), "matrix": np.flatnonzero(~inside).astype(np.int32)}
    matrix_law = lambda: jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=nu), jm.LinearHardening(sig0, H))  # noqa: E731
    laws = {"matrix": matrix_law(), "inclusion": matrix_law() if same else jm.ElasticBehavior(jm.LinearElasticIsotropic(E=5 * E, nu=nu))}
    maps = []
    for name in ("matrix", "inclusion"):
        q = QuadratureFieldMap(mesh.num_cells, mesh.nqp, JAXMaterial(laws[name]), cells=cells[name])
        if maps:   # one stress field and one tangent field over all cells: the maps own disjoint rows of them
            q.fluxes, q.jacobian_flatten = maps[0].fluxes, maps[0].jacobian_flatten
        q.register_gradient("strain", lambda c: mesh.strain(u, c))
        if device_gradient:
            from dolfinx_materials_amd.gradient import Hex8Mesh

            q.register_device_gradient(Hex8Mesh(mesh.coords, mesh.conn[cells[name]]), lambda: u)
        maps.append(q)
    both = Maps(maps)
    both.update()
    x0, x1 = mesh.nodes_on(0, 0.0), mesh.nodes_on(0, 1.0)
    y0, z0 = mesh.nodes_on(1, 0.0), mesh.nodes_on(2, 0.0)
    bc_dofs = np.concatenate([3 * x0, 3 * x1, 3 * y0 + 1, 3 * z0 + 2])
    timers, hist = {}, []
    t_all = time.perf_counter()
    for k in range(1, steps + 1):
        exx = exx_max * k / steps
        bc_vals = np.concatenate([np.zeros(len(x0)), np.full(len(x1), exx), np.zeros(len(y0)), np.zeros(len(z0))])
        norms = newton_solve(mesh, both, u, bc_dofs, bc_vals, mesh.B_eps, "stress", timers=timers, solver=solver, log=print if verbose else None)
        sxx = both.fluxes["stress"].values[:, 0].reshape(mesh.num_cells, -1)
        p = maps[0].internal_state_variables["p"].values.reshape(mesh.num_cells, -1)
        hist.append(dict(exx=exx, sxx_matrix=float(sxx[cells["matrix"]].mean()), sxx_inclusion=float(sxx[cells["inclusion"]].mean()),
                         p_max=float(p.max()), iters=len(norms), norms=norms))
        if verbose:
            print(f"step {k:2d} exx={exx:.4f} <sxx> matrix {hist[-1]['sxx_matrix']:.3f} inclusion {hist[-1]['sxx_inclusion']:.3f} "
                  f"p_max={hist[-1]['p_max']:.5f} newton={len(norms)} |r|={norms[-1]:.2e}")
    timers["total"] = time.perf_counter() - t_all
    rows = [bool(q._accel_plan().row_outputs) for q in maps]
    for q in maps:
        q.close()
        q.material.close()
    return dict(n=n, points=mesh.num_cells * 8, points_per_map={k: int(len(v) * 8) for k, v in cells.items()}, history=hist, timers=timers,
                delivered_into_rows=rows, E=E, nu=nu, sig0=sig0, H=H, same=same)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--exx", type=float, default=1e-2)
    ap.add_argument("--same", action="store_true", help="both cell sets get the matrix material: must reproduce the homogeneous closed form")
    ap.add_argument("--device-gradient", action="store_true")
    ap.add_argument("--solver", default="auto", choices=["auto", "direct", "krylov"])
    a = ap.parse_args()
    out = run(a.n, a.steps, exx_max=a.exx, same=a.same, device_gradient=a.device_gradient, solver=a.solver)
    h, t = out["history"][-1], out["timers"]
    its = max(t.get("newton_iterations", 1), 1)
    rec = {"n": out["n"], "points_per_map": out["points_per_map"], "delivered_into_rows": out["delivered_into_rows"],
           "sxx_matrix": h["sxx_matrix"], "sxx_inclusion": h["sxx_inclusion"], "p_max": h["p_max"], "newton_iterations": its,
           "seconds_per_newton_iteration": {k: round(t.get(k, 0.0) / its, 4) for k in ("constitutive", "assembly", "solve")}}
    if a.same:
        expect = (out["sig0"] + out["H"] * h["exx"]) / (1 + out["H"] / out["E"])
        rec.update(closed_form=expect, rel_err=abs(h["sxx_matrix"] - expect) / expect)
    print(json.dumps(rec))
