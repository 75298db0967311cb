#!/usr/bin/env python3
"""Interleaved A/B timing of launch variants of the J2 kernel in ONE process
(cdna_hip_programming.md section 5.4 rule 24).  Variants are selected through the environment
variables libdxmat reads at dxm_create time."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--law", default="j2_linear")
    ap.add_argument("--stagger", type=int, default=0, help="extra byte offset k*stagger of the k-th boundary array")
    ap.add_argument("--env", nargs="+", default=["DXM_BLOCKS_PER_CU=4", "DXM_BLOCKS_PER_CU=3", "DXM_BLOCKS_PER_CU=5", "DXM_LD_PAD=0", "DXM_LD_PAD=32"])
    a = ap.parse_args()
    import torch

    import bench
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    dev = torch.device("cuda:0")
    n = a.points
    def alloc(rows, cols, k):
        off = (k * a.stagger) // 8
        buf = torch.empty(rows * cols + off + 64, dtype=torch.float64, device=dev)
        return buf[off : off + rows * cols].view(rows, cols)

    eps = []
    for k, h in enumerate(bench.history(n, 1234)):
        e = alloc(n, 6, k)
        e.copy_(torch.from_numpy(h))
        eps.append(e)
    flux = alloc(n, 6, 4)
    ct = alloc(n, 36, 5)
    st = torch.cuda.current_stream().cuda_stream
    el = jm.LinearElasticIsotropic(E=bench.E, nu=bench.NU)
    hard = jm.LinearHardening(bench.SIG0, bench.H) if a.law == "j2_linear" else jm.VoceHardening(bench.SIG0, 2 * bench.SIG0, 1e3)

    variants = {}
    for spec in a.env:
        for kv in spec.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
        mats = []
        for k in (2, 3, 4):
            m = JAXMaterial(jm.vonMisesIsotropicHardening(el, hard))
            m.set_data_manager(n)
            for i in range(k - 1):
                m.integrate_device(eps[i].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
                m.data_manager.update()
            mats.append(m)
        variants[spec] = mats
        for kv in spec.split(","):
            os.environ.pop(kv.split("=")[0])
    times = {(s, j): [] for s in variants for j in range(3)}
    for r in range(a.rounds + 2):
        for s, mats in variants.items():
            for j in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                mats[j].integrate_device(eps[j + 1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times[(s, j)].append(e0.elapsed_time(e1))
    print(json.dumps({"eps": [hex(e.data_ptr()) for e in eps], "flux": hex(flux.data_ptr()), "ct": hex(ct.data_ptr())}))
    for s in variants:
        row = {"variant": s}
        tot = 0
        for j in range(3):
            t = np.array(times[(s, j)])
            row[f"inc{j+2}_med_ms"] = round(float(np.median(t)), 4)
            row[f"inc{j+2}_min_ms"] = round(float(t.min()), 4)
            tot += np.median(t)
        row["avg_med_ms"] = round(tot / 3, 4)
        row["GBs_avg"] = round(496 * n / (tot / 3) / 1e6, 1)
        row["s0_s1_addr"] = [[hex(m._lib.dxm_state_ptr(m._handle, 0, 0, 0)), hex(m._lib.dxm_state_ptr(m._handle, 1, 0, 0))] for m in variants[s]]
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
