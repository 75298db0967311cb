#!/usr/bin/env python3
"""Per-law kernel throughput on one GPU (device-resident inputs/outputs, HIP events on the launch
stream).  Not the headline bench (that is bench.py); used to fill the per-law table of DESIGN.md.

    python tools/bench_laws.py [--points 10000000] [--reps 20] [--laws elastic j2_linear j2_voce fefp]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--laws", nargs="+", default=["elastic", "j2_linear", "j2_voce", "fefp"])
    a = ap.parse_args()
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, SIG0_F, SIGU_F, B_F, j2_history, fefp_path

    dev = torch.device("cuda:0")
    n = a.points
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    res = []
    for law in a.laws:
        if law == "elastic":
            beh, hist = jm.ElasticBehavior(el), j2_history(n)[1:3]
        elif law == "j2_linear":
            beh, hist = jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN)), j2_history(n)[1:3]
        elif law == "j2_voce":
            beh, hist = jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V)), j2_history(n, sig0=SIG0_V)[1:3]
        else:
            path = fefp_path(n)
            beh, hist = jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F)), [path[9], path[18]]
        m = JAXMaterial(beh)
        m.set_data_manager(n)
        ng, nf = m._info.n_grad, m._info.n_flux
        g = [torch.from_numpy(h).to(dev) for h in hist]
        del hist
        flux = torch.empty((n, nf), dtype=torch.float64, device=dev)
        ct = torch.empty((n, nf * ng), dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()  # s0 = state after the first increment
        for _ in range(3):
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        rc, stats = m.stats()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
        for e0, e1 in ev:
            e0.record()
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            e1.record()
        torch.cuda.synchronize()
        ms = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))
        ab = m.algorithmic_bytes_per_point
        r = {
            "law": law, "points": n, "kernel_ms": round(ms, 4), "Mpoints_per_s": round(n / ms / 1e3, 1),
            "algorithmic_bytes_per_point": ab, "GBs": round(ab * n / ms / 1e6, 1), "frac_of_8TBs": round(ab * n / ms / 1e6 / 8000, 4),
            "plastic_fraction": round(stats["n_plastic"] / n, 4), "max_local_iters": stats["max_local_iters"],
            "not_converged": stats["n_not_converged"], "rc": rc,
        }
        print(json.dumps(r), flush=True)
        res.append(r)
        del m, g, flux, ct
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
