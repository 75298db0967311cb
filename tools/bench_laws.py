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


def cpu_port(law, ns, budget=4.0):
    """Plain-C oracle (oracle/oracle_c.c) on the host cores: best of a small thread-count scan."""
    import time

    from oracle import oracle_c as oc
    from oracle import constitutive_np as onp
    from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, SIG0_F, SIGU_F, B_F, j2_history, fefp_path

    if law == "fefp":
        path = fefp_path(ns)
        st = onp.fefp_initial_state(ns)
        r0 = oc.fefp(path[9], st["cpinv"], st["p"], E, NU, SIG0_F, SIGU_F, B_F, nthreads=8)
        cp, p = r0["cpinv"].copy(), r0["p"].copy()
        fn = lambda nt: oc.fefp(path[18], cp, p, E, NU, SIG0_F, SIGU_F, B_F, nthreads=nt, out=r0)  # noqa: E731
    elif law == "elastic":
        eps = j2_history(ns)[2]
        fn = lambda nt: oc.elastic_iso(eps, E, NU, nthreads=nt)  # noqa: E731
    else:
        kind, s0, h1, h2 = (0, SIG0_LIN, H_LIN, 0.0) if law == "j2_linear" else (1, SIG0_V, SIGU_V, B_V)
        h = j2_history(ns, sig0=s0)
        r0 = oc.j2(h[1], np.zeros((ns, 6)), np.zeros(ns), E, NU, kind, s0, h1, h2, nthreads=8)
        ep, p = r0["epsp"].copy(), r0["p"].copy()
        fn = lambda nt: oc.j2(h[2], ep, p, E, NU, kind, s0, h1, h2, nthreads=nt, out=r0)  # noqa: E731
    ncpu = os.cpu_count() or 1
    best = (0.0, 1)
    for nt in sorted({t for t in (1, 8, 16, 32, 64) if t <= ncpu}):
        fn(nt)
        t0, calls = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget / 5 or calls < 2:
            fn(nt)
            calls += 1
        rate = ns * calls / (time.perf_counter() - t0) / 1e6
        if rate > best[0]:
            best = (rate, nt)
    return {"Mpoints_per_s": round(best[0], 2), "threads": best[1], "sample": ns, "kind": "port (oracle/oracle_c.c)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3, help="untimed launches before the timed ones (a fresh allocation needs 10-20 launches to reach its steady time)")
    ap.add_argument("--laws", nargs="+", default=["elastic", "j2_linear", "j2_voce", "fefp"])
    ap.add_argument("--cpu-sample", type=int, default=0, help="also time the plain-C oracle on this many points")
    ap.add_argument("--sym", action="store_true", help="symmetric-packed tangent (small-strain laws)")
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
        sym = a.sym and law != "fefp"
        m = JAXMaterial(beh, tangent_layout="sym" if sym else "full")
        m.set_data_manager(n)
        ng, nf = m._info.n_grad, m._info.n_flux
        from helpers import to_device

        g = [to_device(h) for h in hist]
        del hist
        flux = torch.empty((n, nf), dtype=torch.float64, device=dev)
        ct = torch.empty((n, nf * ng), dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()  # s0 = state after the first increment
        for _ in range(a.warmup):
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        rc, stats = m.stats()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
        for e0, e1 in ev:
            e0.record()
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            e1.record()
        torch.cuda.synchronize()
        ms = float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))
        ab = m.algorithmic_bytes_per_point - (15 * 8 if sym else 0)
        r = {
            "law": law + ("+sym21" if sym else ""), "points": n, "kernel_ms": round(ms, 4), "Mpoints_per_s": round(n / ms / 1e3, 1),
            "algorithmic_bytes_per_point": ab, "GBs": round(ab * n / ms / 1e6, 1), "frac_of_8TBs": round(ab * n / ms / 1e6 / 8000, 4),
            "plastic_fraction": round(stats["n_plastic"] / n, 4), "max_local_iters": stats["max_local_iters"],
            "not_converged": stats["n_not_converged"], "rc": rc,
        }
        if a.cpu_sample:
            r["cpu_port"] = cpu_port(law, a.cpu_sample)
        print(json.dumps(r), flush=True)
        res.append(r)
        del m, g, flux, ct
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
