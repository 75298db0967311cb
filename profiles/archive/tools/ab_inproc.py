#!/usr/bin/env python3
"""Same-process A/B of alternative builds of libdxmat.so: every build gets its own handle, all
handles share the gradient / flux / tangent buffers, and the timing rounds are interleaved, so box,
clock and buffer placement are common to all variants (separate processes on the same box differ
by ~5 %).

    python tools/ab_inproc.py --law j2_linear lib_a.so lib_b.so ...
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--law", default="j2_linear", choices=["elastic", "j2_linear", "j2_voce", "fefp"])
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--advance", action="store_true", help="advance after every update (the new state is read back by the next one)")
    ap.add_argument("--tune", type=int, default=0, help="let every handle search this many state placements first (best-vs-best comparison)")
    a = ap.parse_args()
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd import _lib
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, SIG0_F, SIGU_F, B_F, j2_history, fefp_path

    dev = torch.device("cuda:0")
    n = a.points
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if a.law == "elastic":
        mk, hist = (lambda: jm.ElasticBehavior(el)), j2_history(n)[1:3]
    elif a.law == "j2_linear":
        mk, hist = (lambda: jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))), j2_history(n)[1:3]
    elif a.law == "j2_voce":
        mk, hist = (lambda: jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V))), j2_history(n, sig0=SIG0_V)[1:3]
    else:
        path = fefp_path(n)
        mk, hist = (lambda: jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F))), [path[9], path[18]]
    from helpers import to_device

    g = [to_device(h) for h in hist]
    del hist
    st = torch.cuda.current_stream().cuda_stream
    _lib.load()  # torch first, then HIP (one runtime)
    mats = []
    flux = ct = None
    for path in a.libs:
        lib = _lib._bind(ctypes.CDLL(os.path.abspath(path)), strict=False)
        orig = _lib.load
        _lib.load = lambda lib=lib: lib
        try:
            m = JAXMaterial(mk())
        finally:
            _lib.load = orig
        m.set_data_manager(n)
        if flux is None:
            ng, nf = m._info.n_grad, m._info.n_flux
            flux = torch.empty((n, nf), dtype=torch.float64, device=dev)
            ct = torch.empty((n, nf * ng), dtype=torch.float64, device=dev)
        m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        for _ in range(3):
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        if a.tune:
            try:
                m.set_option("tune_max_skip_bytes", 16 * 2**30)
            except Exception:
                pass  # older builds search deep by default
            m.tune_info = m.tune_placement(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), max_candidates=a.tune)
        mats.append(m)
    times = [[] for _ in mats]
    for _ in range(a.rounds):
        for k, m in enumerate(mats):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
            for e0, e1 in ev:
                e0.record()
                m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
                e1.record()
                if a.advance:
                    m.data_manager.update()
            torch.cuda.synchronize()
            times[k] += [e0.elapsed_time(e1) for e0, e1 in ev]
    checks = []
    for m in mats:   # all builds must agree bit for bit (from the same state: skipped with --advance)
        if a.advance:
            checks.append(None)
            continue
        m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        torch.cuda.synchronize()
        checks.append((float(flux.sum()), float(ct.sum())))
    for k, m in enumerate(mats):
        ab = m.algorithmic_bytes_per_point
        ms = float(np.median(times[k]))
        print(json.dumps({"lib": os.path.basename(a.libs[k]), "law": a.law, "advance": a.advance, "points": n, "median_ms": round(ms, 4),
                          "min_ms": round(float(np.min(times[k])), 4), "frac_of_8TBs": round(ab * n / ms / 1e6 / 8000, 4),
                          "same_result": checks[k] == checks[0],
                          "s0": hex(m._lib.dxm_state_ptr(m._handle, 0, 0, 0) or 0), "s1": hex(m._lib.dxm_state_ptr(m._handle, 1, 0, 0) or 0),
                          "grad": hex(g[1].data_ptr()), "flux": hex(flux.data_ptr()), "ct": hex(ct.data_ptr())}), flush=True)


if __name__ == "__main__":
    main()
