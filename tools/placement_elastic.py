#!/usr/bin/env python3
"""EXPERIMENT: elastic kernel (3 streams, no resident state) with strain / stress / tangent carved from one
64 GiB allocation whose 8 GiB class map is measured first (copy rate against chunk 0): does the class of
each array decide the kernel time?"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GIB = 1 << 30


def main():
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, j2_history

    dev = torch.device("cuda:0")
    n = 10_000_000
    big = torch.zeros(64 * GIB, dtype=torch.uint8, device=dev)
    f64 = big.view(torch.float64)

    def tm(fn, reps=10):
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        return float(np.median([e0.elapsed_time(e1) for e0, e1 in ev]))

    ce = GIB // 8
    rates = [0] + [round(2 * GIB / tm(lambda k=k: f64[k * ce:(k + 1) * ce].copy_(f64[:ce]), 4) / 1e6) for k in range(1, 64)]
    thr = (max(rates[1:]) + min(rates[1:])) / 2
    cls = [1] + [1 if r > thr else 0 for r in rates[1:]]
    cls = [cls[0]] + [1 if cls[k - 1] + cls[k] + cls[k + 1] >= 2 else 0 for k in range(1, 63)] + [cls[63]]
    print(json.dumps({"class_map_1GiB_chunks": "".join("A" if c else "B" for c in cls)}), flush=True)

    def windows(c, need=4):
        out, k = [], 0
        while k + need <= 64:
            if all(x == c for x in cls[k:k + need]):
                out.append(k)
                k += need
            else:
                k += 1
        return out

    A, B = windows(1), windows(0)
    if len(A) < 3 or len(B) < 3:
        print(json.dumps({"error": "no clean class structure", "A": A, "B": B}))
        return
    m = JAXMaterial(jm.ElasticBehavior(jm.LinearElasticIsotropic(E=E, nu=NU)))
    m.set_data_manager(n)
    h = j2_history(n)[1]
    st = torch.cuda.current_stream().cuda_stream

    def arr(chunk, cols):   # array starting at the given 1 GiB chunk (<= 2.88 GB: stays inside a 4-chunk window)
        off = chunk * ce
        return f64[off: off + n * cols].view(n, cols)

    layouts = {
        "eps A0, sig A1, ct A2 (all class A)": (A[0], A[1], A[2]),
        "eps B0, sig B1, ct B2 (all class B)": (B[0], B[1], B[2]),
        "eps A0, sig A1, ct B0": (A[0], A[1], B[0]),
        "eps A0, sig B0, ct A1": (A[0], B[0], A[1]),
        "eps B0, sig A0, ct A1": (B[0], A[0], A[1]),
        "eps A0, sig B0, ct B1": (A[0], B[0], B[1]),
        "eps B0, sig B1, ct A0": (B[0], B[1], A[0]),
    }
    # H2: do lockstep equal-rate streams (strain and stress, 48 B/point each) alias?  Shift the stress array by
    # a byte offset relative to a fixed strain array (all in class A), tangent fixed.
    e, c = arr(A[0], 6), arr(A[2], 36)
    e.copy_(torch.from_numpy(h))
    base_s = A[1] * ce
    for dbytes in (0, 256, 4096, 65536, 1 << 20, (2 << 20) + 4096, (3 << 20) + 65536 + 256, 16 << 20, (100 << 20) + 4096 + 256, (333 << 20) + 8192):
        off = base_s + dbytes // 8
        sarr = f64[off: off + n * 6].view(n, 6)
        ms = tm(lambda: m.integrate_device(e.data_ptr(), sarr.data_ptr(), c.data_ptr(), st), 20)
        print(json.dumps({"shift_stress_by_bytes": dbytes, "ms": round(ms, 4)}), flush=True)
    # and the tangent array shifted, strain / stress fixed
    sfix = arr(A[1], 6)
    for dbytes in (0, 4096, 65536, (1 << 20) + 4096, (37 << 20) + 65536 + 256):
        off = A[2] * ce + dbytes // 8
        carr = f64[off: off + n * 36].view(n, 36)
        ms = tm(lambda: m.integrate_device(e.data_ptr(), sfix.data_ptr(), carr.data_ptr(), st), 20)
        print(json.dumps({"shift_tangent_by_bytes": dbytes, "ms": round(ms, 4)}), flush=True)
    for name, (ce_, cs_, cc_) in layouts.items():
        e, s, c = arr(ce_, 6), arr(cs_, 6), arr(cc_, 36)
        e.copy_(torch.from_numpy(h))
        ms = tm(lambda: m.integrate_device(e.data_ptr(), s.data_ptr(), c.data_ptr(), st), 20)
        print(json.dumps({"layout": name, "chunks": [ce_, cs_, cc_], "ms": round(ms, 4), "frac_of_8TBs": round(384 * n / ms / 1e6 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
