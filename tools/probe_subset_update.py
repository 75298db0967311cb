#!/usr/bin/env python3
"""Measurement (round 6): where one ``update()`` of a map over a SUBSET of the cells spends its time (cProfile of the bench's
`accelerated_update_subset_of_cells` leg: 5e6 points of every other cell in fields over 1e7 points)."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(npts=5_000_000, mode_isv=True, layout="full"):
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.field_map import QuadratureFieldMap
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, j2_history

    h = j2_history(npts)
    ncell = npts // 8
    cells = np.arange(0, 2 * ncell, 2)
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)), tangent_layout=layout)
    q = QuadratureFieldMap(2 * ncell, 8, m, cells=cells)
    q.isv_every_update = mode_isv
    strain = h[0][:ncell * 8]
    q.register_gradient("strain", lambda c, strain=strain: strain.reshape(len(c), -1))
    q.update()
    q.advance()
    buf = q._accel_plan().grad_buffers["strain"]
    buf[...] = h[1][:ncell * 8]

    class Ready:
        def eval(self, mesh, cells, values=None):
            return values

    q.gradients["strain"].expression = Ready()
    q.update()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        q.update()
        ts.append(time.perf_counter() - t0)
    print(f"layout={layout} isv_every_update={mode_isv!r}: ms per update {[round(t * 1e3, 1) for t in ts]}", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        q.update()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
    print(s.getvalue()[:4000], flush=True)
    m.set_option("verbose", 1)
    q.update()
    m.set_option("verbose", 0)
    q.close()
    m.close()


if __name__ == "__main__":
    for layout in (sys.argv[1:] or ["full"]):
        main(mode_isv=True, layout=layout)
        main(mode_isv="lazy", layout=layout)
