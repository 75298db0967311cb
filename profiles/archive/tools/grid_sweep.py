#!/usr/bin/env python3
"""Kernel time against the grid size (option "blocks_per_cu"; the default is the number of RESIDENT
workgroups per CU, i.e. a persistent grid), every handle with its state placement tuned first.

    python tools/grid_sweep.py [j2_linear|j2_voce|elastic|fefp] [bpc ...]
"""
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
    from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, SIG0_F, SIGU_F, B_F, j2_history, fefp_path

    law = sys.argv[1] if len(sys.argv) > 1 else "j2_linear"
    variants = sys.argv[2:] or ["0", "8", "16", "32", "48", "64", "96", "128"]
    dev = torch.device("cuda:0")
    n = int(os.environ.get("POINTS", "10000000"))
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if law == "fefp":
        path = fefp_path(n)
        hist = [path[9], path[18]]
        mk = lambda: jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F))  # noqa: E731
    elif law == "elastic":
        hist = j2_history(n)[1:3]
        mk = lambda: jm.ElasticBehavior(el)  # noqa: E731
    elif law == "j2_voce":
        hist = j2_history(n, sig0=SIG0_V)[1:3]
        mk = lambda: jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V))  # noqa: E731
    else:
        hist = j2_history(n)[1:3]
        mk = lambda: jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))  # noqa: E731
    from helpers import to_device

    g = [to_device(h) for h in hist]
    ng = g[0].shape[1]
    flux = torch.empty((n, ng), dtype=torch.float64, device=dev)
    ct = torch.empty((n, ng * ng), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    if os.environ.get("TANGENT_SEARCH") == "1":   # first a tangent array where the kernel likes it (HIPMaterial.fastest_tangent_array)
        scout = JAXMaterial(mk())
        scout.set_data_manager(n)
        scout.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        scout.data_manager.update()
        ct, t_search, k_best = scout.fastest_tangent_array(lambda: torch.empty((n, ng * ng), dtype=torch.float64, device=dev), g[1].data_ptr(), flux.data_ptr())
        print(json.dumps({"tangent_array_search_ms": [round(t, 4) for t in t_search], "kept": k_best}), flush=True)
        scout.close()
        torch.cuda.empty_cache()
    mats = []
    for v in variants:
        m = JAXMaterial(mk())
        m.set_data_manager(n)
        if v != "0":
            m.set_option("blocks_per_cu", int(v))
        m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        info = m.tune_placement(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr())
        mats.append((v, m, info))
    times = [[] for _ in mats]
    for _ in range(6):
        for k, (_, m, _) in enumerate(mats):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
            for e0, e1 in ev:
                e0.record()
                m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
                e1.record()
            torch.cuda.synchronize()
            times[k] += [e0.elapsed_time(e1) for e0, e1 in ev]
    for k, (v, m, info) in enumerate(mats):
        print(json.dumps({"law": law, "points": n, "blocks_per_cu": "default" if v == "0" else int(v),
                          "median_ms": round(float(np.median(times[k])), 4), "min_ms": round(float(np.min(times[k])), 4),
                          "tune": {a: round(b, 4) for a, b in info.items()}}), flush=True)


if __name__ == "__main__":
    main()
