#!/usr/bin/env python3
"""EXPERIMENT: with the state placement tuned, how much does the placement of the CALLER's arrays
(gradient, flux, tangent) still matter?  Tries several allocations of each, one at a time."""
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

    law = sys.argv[1] if len(sys.argv) > 1 else "j2_linear"
    dev = torch.device("cuda:0")
    n = 10_000_000
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    hist = j2_history(n)[1:3]
    g0 = torch.from_numpy(hist[0]).to(dev)
    g1 = torch.from_numpy(hist[1]).to(dev)
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    beh = jm.ElasticBehavior(el) if law == "elastic" else jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))
    m = JAXMaterial(beh)
    m.set_data_manager(n)
    m.integrate_device(g0.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
    m.data_manager.update()

    def tm(g, f, c, reps=12):
        for _ in range(2):
            m.integrate_device(g.data_ptr(), f.data_ptr(), c.data_ptr(), st)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record()
            m.integrate_device(g.data_ptr(), f.data_ptr(), c.data_ptr(), st)
            e1.record()
        torch.cuda.synchronize()
        return round(float(np.median([a.elapsed_time(b) for a, b in ev])), 4)

    def tuned(g, f, c):
        info = m.tune_placement(g.data_ptr(), f.data_ptr(), c.data_ptr())
        return tm(g, f, c), info["candidates_tried"]

    base = tuned(g1, flux, ct)
    print(json.dumps({"law": law, "baseline_tuned_ms": base[0], "tried": base[1]}), flush=True)
    keep = []
    for k in range(8):
        c2 = torch.empty((n, 36), dtype=torch.float64, device=dev)
        keep.append(c2)
        t_plain = tm(g1, flux, c2)
        t_tuned, tried = tuned(g1, flux, c2)
        print(json.dumps({"vary": "tangent", "alloc": k, "addr": hex(c2.data_ptr()), "ms_state_as_is": t_plain, "ms_state_retuned": t_tuned, "tried": tried}), flush=True)
    del keep
    torch.cuda.empty_cache()
    keep = []
    for k in range(8):
        gg = g1.clone()
        keep.append(gg)
        t_plain = tm(gg, flux, ct)
        t_tuned, tried = tuned(gg, flux, ct)
        print(json.dumps({"vary": "gradient", "alloc": k, "addr": hex(gg.data_ptr()), "ms_state_as_is": t_plain, "ms_state_retuned": t_tuned, "tried": tried}), flush=True)
    del keep
    torch.cuda.empty_cache()
    keep = []
    for k in range(8):
        f2 = torch.empty((n, 6), dtype=torch.float64, device=dev)
        keep.append(f2)
        t_plain = tm(g1, f2, ct)
        t_tuned, tried = tuned(g1, f2, ct)
        print(json.dumps({"vary": "flux", "alloc": k, "addr": hex(f2.data_ptr()), "ms_state_as_is": t_plain, "ms_state_retuned": t_tuned, "tried": tried}), flush=True)


if __name__ == "__main__":
    main()
