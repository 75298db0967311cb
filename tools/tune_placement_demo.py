#!/usr/bin/env python3
"""What dxm_tune_placement buys on this box: K fresh J2 handles at 1e7 points, kernel time before and
after tuning with the real gradient / flux / tangent buffers."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, j2_history

    law = sys.argv[1] if len(sys.argv) > 1 else "j2_linear"
    dev = torch.device("cuda:0")
    n = 10_000_000
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    hist = j2_history(n)[1:3]
    from helpers import to_device

    g = [to_device(h) for h in hist]
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def tm(m, reps=20):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record()
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            e1.record()
        torch.cuda.synchronize()
        return round(float(np.median([a.elapsed_time(b) for a, b in ev])), 4)

    for k in range(4):
        m = JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN)))
        m.set_data_manager(n)
        m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        for _ in range(3):
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        before = tm(m)
        import time
        t0 = time.perf_counter()
        info = m.tune_placement(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), max_candidates=int(os.environ.get("TUNE_K", "24")))
        wall = time.perf_counter() - t0
        after = tm(m)
        print(json.dumps({"handle": k, "median_ms_before": before, "median_ms_after": after, "tune": {a: round(b, 4) for a, b in info.items()},
                          "tune_wall_s": round(wall, 3)}), flush=True)


if __name__ == "__main__":
    main()
