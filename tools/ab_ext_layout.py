#!/usr/bin/env python3
"""EXPERIMENT: SoA vs tile-blocked state layout (library copies built with the external-state hook,
`_jit/ab/libdxmat_ext.so` and `libdxmat_ext_tiled.so`), handles of both kinds placed inside the SAME
pools so that the placement mode is common."""
import ctypes
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
    from dolfinx_materials_amd import _lib
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
    _lib.load()
    # LIBS="name=lib.so[:blocks_per_cu],..."
    pairs = [x.split("=") for x in os.environ.get("LIBS", "soa=libdxmat_ext.so,tiled=libdxmat_ext_tiled.so").split(",")]
    bpc = {k: (f.split(":")[1] if ":" in f else None) for k, f in pairs}
    pairs = [(k, f.split(":")[0]) for k, f in pairs]
    libs = {k: _lib._bind(ctypes.CDLL(os.path.join(ROOT, "dolfinx_materials_amd", "_jit", "ab", f))) for k, f in pairs}
    kinds = [k for k, _ in pairs] * 2
    pools = [torch.zeros(len(kinds) * (1200 << 20) + (64 << 20), dtype=torch.uint8, device=dev) for _ in range(4)]
    mats = []
    for pi, p in enumerate(pools):
        for slot, kind in enumerate(kinds):
            a0 = p.data_ptr() + slot * (1200 << 20)
            os.environ["DXM_STATE_EXTERNAL"] = hex(a0)
            os.environ["DXM_STATE_EXTERNAL_S1"] = hex(a0 + half)
            os.environ.pop("DXM_BLOCKS_PER_CU", None)
            if bpc[kind]:
                os.environ["DXM_BLOCKS_PER_CU"] = bpc[kind]
            orig = _lib.load
            _lib.load = lambda lib=libs[kind]: lib
            try:
                m = JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN)))
            finally:
                _lib.load = orig
            m.set_data_manager(n)
            m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            m.data_manager.update()
            for _ in range(3):
                m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
            mats.append((pi, kind, m))
    times = [[] for _ in mats]
    for _ in range(6):
        for k, (_, _, m) in enumerate(mats):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
            for e0, e1 in ev:
                e0.record()
                m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
                e1.record()
            torch.cuda.synchronize()
            times[k] += [e0.elapsed_time(e1) for e0, e1 in ev]
    for k, (pi, kind, m) in enumerate(mats):
        rc, s = m.stats()
        print(json.dumps({"pool": pi, "layout": kind, "median_ms": round(float(np.median(times[k])), 4), "plastic": s["n_plastic"], "rc": rc}), flush=True)


if __name__ == "__main__":
    main()
