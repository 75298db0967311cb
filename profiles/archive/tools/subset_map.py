"""One `QuadratureFieldMap.update()` of a map over a SUBSET of the cells (every other cell of 2 * ncell hexahedra x 8 points):
the accelerated mixin with the engine delivering into the rows itself (`integrate_rows`), with its threaded row scatter after
`integrate`, and with numpy's fancy assignment, per update and per advance.

    python tools/subset_map.py [--points 5000000] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import dolfinx_materials_amd.materials as jm  # noqa: E402
from dolfinx_materials_amd.field_map import QuadratureFieldMap  # noqa: E402
from dolfinx_materials_amd.jaxmat import JAXMaterial  # noqa: E402

E, NU, SIG0, H = 70e3, 0.3, 350.0, 1e3


def run(npts, reps, mode, contiguous=False):
    nqp = 8
    ncell = npts // nqp
    cells = np.arange(ncell) if contiguous else np.arange(0, 2 * ncell, 2)
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0, H)))
    if mode == "numpy":   # what the mixin does with a material that offers no row moves: numpy's fancy indexing
        m.scatter_rows = m.gather_rows = None
    q = QuadratureFieldMap(2 * ncell, nqp, m, cells=cells)
    if mode != "engine":   # "threads": integrate + dxm_host_scatter_rows; "numpy": integrate + fancy assignment
        q._accel_plan().row_outputs = False
    rng = np.random.default_rng(0)
    strain = rng.standard_normal((len(cells) * nqp, 6)) * 2e-3
    q.register_gradient("strain", lambda c: strain.reshape(len(cells), -1))
    q.update()
    q.advance()
    # the compiled expression at zero cost (as bench.py's update_cadence does): the next strain is already where
    # Expression.eval(mesh, cells, values=...) would write it, the map's persistent page-locked gradient rows
    buf = q._accel_plan().grad_buffers["strain"]
    buf[...] = strain * 1.5

    class Ready:
        def eval(self, mesh, cells, values=None):
            assert values is not None and values.ctypes.data == buf.ctypes.data
            return values

    q.gradients["strain"].expression = Ready()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        q.update()
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    q.advance()
    t_adv = time.perf_counter() - t0
    chk = float(q.jacobian_flatten.x.array[::1001].sum() + q.fluxes["stress"].x.array[::101].sum())
    rec = {"points_in_map": len(cells) * nqp, "points_in_fields": 2 * ncell * nqp, "contiguous_cells": contiguous, "rows_moved_by": mode,
           "ms_per_update": round(float(np.median(ts)) * 1e3, 2), "ms_per_advance": round(t_adv * 1e3, 2), "check": chk}
    q.close()
    m.close()
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=5_000_000)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    for mode in ("engine", "threads", "numpy"):
        print(json.dumps(run(a.points, a.reps, mode)), flush=True)
