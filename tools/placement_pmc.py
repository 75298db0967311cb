#!/usr/bin/env python3
"""EXPERIMENT: J2 launches on handles whose state lives in different allocations, in a fixed order,
to be run under `rocprofv3 --pmc ...` (per-dispatch counters) and bare (times).  Needs the
DXM_STATE_EXTERNAL experiment library.  Timed phase = the LAST 6 x H dispatches of the kernel."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DXM_LIB_PATH"] = os.path.join(ROOT, "dolfinx_materials_amd", "_jit", "ab", os.environ.get("EXT_LIB", "libdxmat_ext.so"))


def main():
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, j2_history

    dev = torch.device("cuda:0")
    n = 10_000_000
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    hist = j2_history(n)[1:3]
    g = [torch.from_numpy(h).to(dev) for h in hist]
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ld = (n + 255) // 256 * 256 + 32
    half = 7 * ld * 8
    pools = [torch.zeros(4 << 30, dtype=torch.uint8, device=dev) for _ in range(5)]
    specs = [("hipMalloc#0", None), ("hipMalloc#1", None)] + [(f"pool{i}", p.data_ptr()) for i, p in enumerate(pools)]
    mats = []
    for label, a0 in specs:
        os.environ.pop("DXM_STATE_EXTERNAL", None)
        os.environ.pop("DXM_STATE_EXTERNAL_S1", None)
        if a0 is not None:
            os.environ["DXM_STATE_EXTERNAL"] = hex(a0)
            os.environ["DXM_STATE_EXTERNAL_S1"] = hex(a0 + half)
        m = JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN)))
        m.set_data_manager(n)
        m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        for _ in range(3):
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        mats.append(m)
    torch.cuda.synchronize()
    out = []
    for k, m in enumerate(mats):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
        for e0, e1 in ev:
            e0.record()
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            e1.record()
        torch.cuda.synchronize()
        out.append({"handle": k, "state": specs[k][0], "median_ms": round(float(np.median([a.elapsed_time(b) for a, b in ev])), 4)})
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
