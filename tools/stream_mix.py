#!/usr/bin/env python3
"""J2 kernel vs. a no-arithmetic streaming kernel with the same bytes per point (104 B read, 392 B
written), interleaved in one process on one box: how close is the constitutive kernel to the
ceiling of its own traffic mix?"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import bench
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    n = 10_000_000 // 64 * 64
    dev = torch.device("cuda:0")
    lib = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    lib.stream_mix_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    rbuf = torch.randn(n * 13, dtype=torch.float64, device=dev)
    wbuf = torch.empty(n * 49, dtype=torch.float64, device=dev)
    eps = [torch.from_numpy(h).to(dev) for h in bench.history(n, 1234)]
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=bench.E, nu=bench.NU), jm.LinearHardening(bench.SIG0, bench.H)))
    m.set_data_manager(n)
    for i in range(2):
        m.integrate_device(eps[i].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
    variants = {"j2_linear": lambda: m.integrate_device(eps[2].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)}
    for blocks in (1024, 2048, 4096):
        variants[f"stream_mix_{blocks}"] = (lambda b: (lambda: lib.stream_mix_launch(rbuf.data_ptr(), wbuf.data_ptr(), n, b, st or None)))(blocks)
    times = {k: [] for k in variants}
    for r in range(14):
        for k, fn in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[k].append(e0.elapsed_time(e1))
    base = None
    for k, t in times.items():
        med = float(np.median(t))
        gbs = 496 * n / med / 1e6
        if k == "j2_linear":
            base = gbs
        print(json.dumps({"kernel": k, "median_ms": round(med, 4), "GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / 8000, 4)}))
    best = max(496 * n / float(np.median(t)) / 1e6 for k, t in times.items() if k != "j2_linear")
    print(json.dumps({"j2_over_best_stream_mix": round(base / best, 4)}))


if __name__ == "__main__":
    main()
