#!/usr/bin/env python3
"""Kernel rate of the J2 law against the batch size (device path, HIP events): where the launch
floor ends and the HBM regime begins."""
import json
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, eps_yield

    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev).manual_seed(1)
    for n in (1_000, 10_000, 100_000, 1_000_000, 10_000_000, 30_000_000):
        d = torch.randn((n, 6), generator=gen, device=dev, dtype=torch.float64)
        d /= d.norm(dim=1, keepdim=True)
        eps = d * (torch.rand((n, 1), generator=gen, device=dev, dtype=torch.float64) * 4.0 * eps_yield(SIG0_LIN))
        m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)))
        m.set_data_manager(n)
        flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
        ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
        for _ in range(5):
            m.integrate_device(eps.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        reps = 200 if n <= 100_000 else 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            m.integrate_device(eps.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(json.dumps({"points": n, "us_per_update": round(us, 2), "Mpoints_per_s": round(n / us, 1),
                          "GBs": round(496 * n / us / 1e3, 1)}), flush=True)
        m.close()
        del eps, flux, ct, d
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
