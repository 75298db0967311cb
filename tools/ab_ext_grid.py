#!/usr/bin/env python3
"""EXPERIMENT: persistent-grid size (DXM_BLOCKS_PER_CU) with every handle's state inside the SAME
pool (experiment build with the external-state hook), so that the placement mode is common."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DXM_LIB_PATH"] = os.path.join(ROOT, "dolfinx_materials_amd", "_jit", "ab", "libdxmat_ext.so")


def main():
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_F, SIGU_F, B_F, j2_history, fefp_path

    law = sys.argv[1] if len(sys.argv) > 1 else "j2_linear"
    dev = torch.device("cuda:0")
    n = 10_000_000
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if law == "fefp":
        path = fefp_path(n)
        hist, nslots = [path[9], path[18]], 13
        mk = lambda: jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F))  # noqa: E731
    else:
        hist, nslots = j2_history(n)[1:3], 7
        mk = lambda: jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))  # noqa: E731
    g = [torch.from_numpy(h).to(dev) for h in hist]
    ng = g[0].shape[1]
    flux = torch.empty((n, ng), dtype=torch.float64, device=dev)
    ct = torch.empty((n, ng * ng), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ld = (n + 255) // 256 * 256 + 32
    half = nslots * ld * 8
    variants = os.environ.get("BPC", "4 8 16 32 64 128 160").split() if law != "fefp" else os.environ.get("BPC", "2 4 8 16 32 64").split()
    pool = torch.zeros(len(variants) * (2 * half + (64 << 20)), dtype=torch.uint8, device=dev)
    mats = []
    for k, v in enumerate(variants):
        a0 = pool.data_ptr() + k * (2 * half + (64 << 20))
        a0 = (a0 + 255) // 256 * 256
        os.environ["DXM_STATE_EXTERNAL"] = hex(a0)
        os.environ["DXM_STATE_EXTERNAL_S1"] = hex(a0 + half)
        os.environ["DXM_BLOCKS_PER_CU"] = v
        m = JAXMaterial(mk())
        m.set_data_manager(n)
        m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        for _ in range(3):
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        mats.append(m)
    times = [[] for _ in mats]
    for _ in range(6):
        for k, m in enumerate(mats):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
            for e0, e1 in ev:
                e0.record()
                m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
                e1.record()
            torch.cuda.synchronize()
            times[k] += [e0.elapsed_time(e1) for e0, e1 in ev]
    for k, v in enumerate(variants):
        print(json.dumps({"law": law, "blocks_per_cu": int(v), "median_ms": round(float(np.median(times[k])), 4)}), flush=True)


if __name__ == "__main__":
    main()
