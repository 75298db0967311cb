#!/usr/bin/env python3
"""FeFp return mapping, one thread per point against nine lanes per point (tools/fefp_subwave_probe.hip): both against the C
oracle on the same inputs (cfg 4's path, state of step 9, update to step 18), then timed with HIP events at 1e6 and 1e7 points,
next to the shipped kernel (return mapping + 9x9 tangent) on the same box.

    python tools/fefp_subwave_probe.py [--points 1000000]         # one JSON line per leg
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --kernel-trace -d out -- python3 tools/fefp_subwave_probe.py --pmc-only
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, nargs="*", default=[1_000_000, 10_000_000])
    ap.add_argument("--pmc-only", action="store_true", help="three launches of each kernel at 1e6 points, nothing else (for rocprofv3 --pmc)")
    a = ap.parse_args()
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_F, SIGU_F, B_F, fefp_path, to_device, to_host
    from oracle import constitutive_np as onp
    from oracle import oracle_c

    lib = C.CDLL(os.path.join(ROOT, "tools", "libfefpsubwave.so"))
    lib.rm_probe_launch.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
    lam, mu = onp.lame(E, NU)
    kappa = lam + 2.0 * mu / 3.0
    rtol = 1e-14
    prm = (C.c_double * 7)(mu, kappa, SIG0_F, SIGU_F, B_F, rtol * max(abs(SIG0_F), 2e-8 * mu), rtol)
    st = torch.cuda.current_stream().cuda_stream
    dev = torch.device("cuda:0")
    sizes = [1_000_000] if a.pmc_only else a.points
    for n in sizes:
        path = fefp_path(n)
        F0, F1 = path[9], path[18]
        ini = onp.fefp_initial_state(n)
        def oracle(F, cp, p):
            out = dict(P=np.empty((n, 9)), be_bar=np.empty((n, 6)), cpinv=np.empty((n, 6)), p=np.empty(n))
            return oracle_c.fefp(F, cp, p, E, NU, SIG0_F, SIGU_F, B_F, nthreads=16, tangent=False, out=out)

        r0 = oracle(F0, ini["cpinv"], ini["p"])
        ld = (n + 255) // 256 * 256 + 32
        s0 = np.zeros((13, ld))
        s0[0, :n] = r0["p"]
        s0[1:7, :n] = r0["be_bar"].T
        s0[7:13, :n] = r0["cpinv"].T
        d_F, d_s0 = to_device(F1), to_device(s0)
        d_s1 = torch.zeros((13, ld), dtype=torch.float64, device=dev)
        d_P = torch.zeros((n, 9), dtype=torch.float64, device=dev)
        ref = None if a.pmc_only else oracle(F1, r0["cpinv"], r0["p"])
        for which, name, blocks in ((0, "one_thread_per_point", 256 * 32), (1, "nine_lanes_per_point", 256 * 32)):
            def launch():
                rc = lib.rm_probe_launch(which, prm, 25, n, d_F.data_ptr(), d_s0.data_ptr(), d_s1.data_ptr(), ld, d_P.data_ptr(), blocks, st or None)
                assert rc == 0, rc
            d_s1.zero_()
            d_P.zero_()
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            if a.pmc_only:
                continue
            P, s1 = to_host(d_P), to_host(d_s1)
            err = {"P": float(np.abs(P - ref["P"]).max() / np.abs(ref["P"]).max()), "p": float(np.abs(s1[0, :n] - ref["p"]).max()),
                   "cpinv": float(np.abs(s1[7:13, :n].T - ref["cpinv"]).max()), "be_bar": float(np.abs(s1[1:7, :n].T - ref["be_bar"]).max())}
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for x, y in ev:
                x.record()
                launch()
                y.record()
            torch.cuda.synchronize()
            ms = float(np.median([x.elapsed_time(y) for x, y in ev]))
            moved = 8 * (9 + 7 + 9 + 13) * n   # F + (p, Cp^-1) in; PK1 + 13 state slots out
            print(json.dumps({"kernel": name, "points": n, "ms": round(ms, 4), "Mpoints_per_s": round(n / ms / 1e3, 1), "GBs_moved": round(moved / ms / 1e6, 1),
                              "max_err_vs_oracle": err, "plastic_fraction": ref["n_plastic"] / n}), flush=True)
        if a.pmc_only:
            continue
        # the shipped kernel (return mapping + tangent, LDS-transposed AoS traffic) on the same inputs, for scale
        m = JAXMaterial(jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(SIG0_F, SIGU_F, B_F)))
        m.set_data_manager(n)
        d_F0 = to_device(F0)
        flux = torch.empty((n, 9), dtype=torch.float64, device=dev)
        ct = torch.empty((n, 81), dtype=torch.float64, device=dev)
        m.integrate_device(d_F0.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        for _ in range(10):
            m.integrate_device(d_F.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for x, y in ev:
            x.record()
            m.integrate_device(d_F.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            y.record()
        torch.cuda.synchronize()
        ms = float(np.median([x.elapsed_time(y) for x, y in ev]))
        errP = float(np.abs(to_host(flux) - ref["P"]).max() / np.abs(ref["P"]).max())
        print(json.dumps({"kernel": "shipped fefp_kernel<1> (return mapping + 9x9 tangent)", "points": n, "ms": round(ms, 4), "Mpoints_per_s": round(n / ms / 1e3, 1),
                          "max_err_P_vs_oracle": errP}), flush=True)
        m.close()
        del d_F, d_s0, d_s1, d_P, flux, ct, d_F0
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
