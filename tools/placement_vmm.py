#!/usr/bin/env python3
"""One time-boxed root-cause attempt on the placement bimodality: does a state buffer
assembled from separately created physical chunks (HIP VMM: hipMemCreate + hipMemMap, in order / permuted)
land in the fast mode of the J2 kernel deterministically, without searching?

Several fresh handles per configuration, all sharing the same gradient / flux / tangent arrays, timing rounds
interleaved in one process; compared with plain hipMalloc placement and with dxm_tune_placement."""
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

    n = 10_000_000
    dev = torch.device("cuda:0")
    h = j2_history(n)
    from helpers import to_device

    g1, g2 = to_device(h[1]), to_device(h[2])
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def make():
        m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)))
        m.set_data_manager(n)
        m.integrate_device(g1.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        return m

    configs = [("hipMalloc", None)] * 4
    for chunk in (2 << 20, 16 << 20, 256 << 20):
        configs += [(f"vmm_inorder_{chunk >> 20}MiB", (1, chunk, 0))] * 2
        configs += [(f"vmm_permuted_{chunk >> 20}MiB", (2, chunk, s)) for s in (1, 2, 3)]
    mats = []
    for name, place in configs:
        m = make()
        if place:
            m.place_state(*place)
        mats.append((name, m))

    def time_one(m, reps=8):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            m.integrate_device(g2.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in ev]))

    res = {i: [] for i in range(len(mats))}
    for rnd in range(4):
        for i, (name, m) in enumerate(mats):
            res[i].append(time_one(m))
    for i, (name, m) in enumerate(mats):
        print(json.dumps({"placement": name, "median_ms": round(float(np.median(res[i])), 4), "rounds_ms": [round(x, 4) for x in res[i]]}), flush=True)
    # reference: what the search finds on this box
    m = make()
    info = m.tune_placement(g2.data_ptr(), flux.data_ptr(), ct.data_ptr())
    print(json.dumps({"placement": "dxm_tune_placement", **{k: round(v, 4) if isinstance(v, float) else v for k, v in info.items()}}), flush=True)


if __name__ == "__main__":
    main()
