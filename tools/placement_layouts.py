#!/usr/bin/env python3
"""EXPERIMENT (needs the DXM_STATE_EXTERNAL experiment library): J2 kernel time for explicit
arrangements of gradient / flux / tangent / state over the 8 GiB-granular memory classes, all carved
from one 64 GiB allocation whose class map is measured first (copy rate against chunk 0)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DXM_LIB_PATH"] = os.path.join(ROOT, "dolfinx_materials_amd", "_jit", "ab", "libdxmat_ext.so")
GIB = 1 << 30


def main():
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, j2_history

    dev = torch.device("cuda:0")
    n = 10_000_000
    big = torch.zeros(64 * GIB, dtype=torch.uint8, device=dev)
    base = big.data_ptr()

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

    # class map in 1 GiB chunks: 1 = same class as chunk 0
    f64 = big.view(torch.float64)
    ce = GIB // 8
    rates = [0] + [round(2 * GIB / tm(lambda k=k: f64[k * ce:(k + 1) * ce].copy_(f64[:ce]), 4) / 1e6) for k in range(1, 64)]
    thr = (max(rates[1:]) + min(rates[1:])) / 2
    cls = [1] + [1 if r > thr else 0 for r in rates[1:]]
    print(json.dumps({"class_map_1GiB_chunks": "".join("A" if c else "B" for c in cls), "rates": rates}), flush=True)
    cls = [cls[0]] + [1 if cls[k - 1] + cls[k] + cls[k + 1] >= 2 else 0 for k in range(1, 63)] + [cls[63]]   # single blips

    def windows(c):
        out, k = [], 0
        while k + 7 <= 64:
            if all(x == c for x in cls[k:k + 7]):
                out.append(k)
                k += 8
            else:
                k += 1
        return out

    A, B = windows(1), windows(0)
    if len(A) < 2 or len(B) < 2:
        print(json.dumps({"error": "no clean class structure on this box", "A": A, "B": B}))
        return
    a0, a1, b0, b1 = A[0] * GIB, A[1] * GIB, B[0] * GIB, B[1] * GIB
    print(json.dumps({"A_windows_GiB": A, "B_windows_GiB": B}), flush=True)
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    hist = j2_history(n)[1:3]
    ld = (n + 255) // 256 * 256 + 32
    half = 7 * ld * 8
    MB = 1 << 20
    # sizes: eps0/eps1 480 MB each, flux 480 MB, ct 2880 MB, s0/s1 560 MB each -> offsets inside an 8 GiB run
    O = {"eps0": 0, "eps1": 512 * MB, "flux": 1024 * MB, "ct": 1536 * MB, "s0": 4608 * MB, "s1": 5248 * MB}   # ends at 5808 MB < 7 GiB

    def arr(off, cols):
        return f64[off // 8: off // 8 + n * cols].view(n, cols)

    st = torch.cuda.current_stream().cuda_stream
    layouts = {
        "all in A0": dict(eps=a0, flux=a0, ct=a0, s0=a0, s1=a0),
        "all in B0": dict(eps=b0, flux=b0, ct=b0, s0=b0, s1=b0),
        "state in B0, rest A0": dict(eps=a0, flux=a0, ct=a0, s0=b0, s1=b0),
        "state in A1 (same class, other region), rest A0": dict(eps=a0, flux=a0, ct=a0, s0=a1, s1=a1),
        "s1 in B0, rest A0": dict(eps=a0, flux=a0, ct=a0, s0=a0, s1=b0),
        "s0 in B0, rest A0": dict(eps=a0, flux=a0, ct=a0, s0=b0, s1=a0),
        "ct in B0, rest A0": dict(eps=a0, flux=a0, ct=b0, s0=a0, s1=a0),
        "ct+state in A0, eps+flux in B0": dict(eps=b0, flux=b0, ct=a0, s0=a0, s1=a0),
        "eps in B0, rest A0": dict(eps=b0, flux=a0, ct=a0, s0=a0, s1=a0),
        "flux in B0, rest A0": dict(eps=a0, flux=b0, ct=a0, s0=a0, s1=a0),
        "reads (eps,s0) in A0, writes (flux,ct,s1) in B0": dict(eps=a0, flux=b0, ct=b0, s0=a0, s1=b0),
        "reads in A0, writes in A1": dict(eps=a0, flux=a1, ct=a1, s0=a0, s1=a1),
        "every array in its own A/B alternately": dict(eps=a0, flux=b0, ct=a1, s0=b1, s1=a0),
    }
    for name, L in layouts.items():
        e0, e1 = arr(L["eps"] + O["eps0"], 6), arr(L["eps"] + O["eps1"], 6)
        e0.copy_(torch.from_numpy(hist[0]))
        e1.copy_(torch.from_numpy(hist[1]))
        flux, ct = arr(L["flux"] + O["flux"], 6), arr(L["ct"] + O["ct"], 36)
        # after the advance the handle READS what it wrote first (the S1 pointer) and WRITES the S0 pointer
        os.environ["DXM_STATE_EXTERNAL"] = hex(base + L["s1"] + O["s1"])
        os.environ["DXM_STATE_EXTERNAL_S1"] = hex(base + L["s0"] + O["s0"])
        m = JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN)))
        m.set_data_manager(n)
        m.integrate_device(e0.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        ms = tm(lambda: m.integrate_device(e1.data_ptr(), flux.data_ptr(), ct.data_ptr(), st), 20)
        rc, s = m.stats()
        assert rc == 0 and s["n_nan"] == 0 and s["n_plastic"] > 0
        rd = m._lib.dxm_state_ptr(m._handle, 0, 0, 0) - base
        print(json.dumps({"layout": name, "ms": round(ms, 4), "frac_of_8TBs": round(496 * n / ms / 1e6 / 8000, 4),
                          "reads_state_at_GiB": round(rd / GIB, 2)}), flush=True)
        m.close()


if __name__ == "__main__":
    main()
