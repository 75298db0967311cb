#!/usr/bin/env python3
"""Soak test: the same launch repeated many times must give bit-identical outputs (checks the
wave-private LDS hand-offs and the tile loop for rare races).  Prints one JSON line per law."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import bench
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n = 10_000_000 + 37   # ragged last tile on purpose
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    el = jm.LinearElasticIsotropic(E=bench.E, nu=bench.NU)
    gen = torch.Generator(device=dev).manual_seed(5)
    d = torch.randn((n, 6), generator=gen, device=dev, dtype=torch.float64)
    d /= d.norm(dim=1, keepdim=True)
    eps = d * (torch.rand((n, 1), generator=gen, device=dev, dtype=torch.float64) * 4.0 * 250.0 / (2 * bench.E / 2 / 1.3) * 0.8165)
    F = torch.randn((n, 9), generator=gen, device=dev, dtype=torch.float64) * 4e-3
    F[:, :3] += 1.0
    F[:, 0] += 2e-2
    cases = [
        ("j2_linear", jm.vonMisesIsotropicHardening(el, jm.LinearHardening(250.0, 5e3)), eps, 6),
        ("j2_voce", jm.vonMisesIsotropicHardening(el, jm.VoceHardening(350.0, 500.0, 1e3)), eps * 1.4, 6),
        ("fefp_j2_voce", jm.FeFpJ2Plasticity(el, jm.VoceHardening(500.0, 750.0, 1000.0)), F, 9),
    ]
    for name, beh, g, ng in cases:
        m = JAXMaterial(beh)
        m.set_data_manager(n)
        flux = torch.empty((n, ng), dtype=torch.float64, device=dev)
        ct = torch.empty((n, ng * ng), dtype=torch.float64, device=dev)
        isv = torch.empty((n, 7), dtype=torch.float64, device=dev)
        m.integrate_device((g * 0.5).data_ptr() if ng == 6 else ((g + torch.tensor([1.0, 1, 1, 0, 0, 0, 0, 0, 0], device=dev, dtype=torch.float64)) * 0.5).data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        ref = None
        bad = 0
        for r in range(reps):
            flux.zero_()
            ct.zero_()
            m.integrate_device(g.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            m.isv_device(1, isv.data_ptr(), st)
            chk = (float(flux.sum()), float(ct.sum()), float(isv.sum()), float(flux[-1].sum()), float(ct[-1].sum()))
            if ref is None:
                ref = chk
            elif chk != ref:
                bad += 1
        rc, stats = m.stats()
        print(json.dumps({"law": name, "points": n, "repetitions": reps, "mismatching_repetitions": bad,
                          "plastic_fraction": round(stats["n_plastic"] / n, 4), "nan": stats["n_nan"], "not_converged": stats["n_not_converged"]}), flush=True)
        m.close()
        del flux, ct, isv
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
