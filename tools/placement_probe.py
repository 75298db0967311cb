#!/usr/bin/env python3
"""Does the kernel time depend on WHERE the resident state lives?  K handles of the same law and
the same library share the gradient / flux / tangent buffers; only their state allocations differ.
Prints the device addresses next to the interleaved median kernel time of every handle.

    python tools/placement_probe.py [--law j2_linear] [--handles 10] [--jitter-mib 0]
"""
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
    ap.add_argument("--law", default="j2_linear", choices=["j2_linear", "j2_voce"])
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--handles", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--jitter-mib", type=int, default=0, help="allocate a dummy buffer of k * this many MiB before handle k")
    ap.add_argument("--variants", nargs="*", default=[""],
                    help="environment settings applied round-robin at handle creation, e.g. '' DXM_LD_PAD=64 DXM_S1_SKEW=256")
    a = ap.parse_args()
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_V, SIGU_V, B_V, j2_history

    dev = torch.device("cuda:0")
    n = a.points
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if a.law == "j2_linear":
        mk, hist = (lambda: jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))), j2_history(n)[1:3]
    else:
        mk, hist = (lambda: jm.vonMisesIsotropicHardening(el, jm.VoceHardening(SIG0_V, SIGU_V, B_V))), j2_history(n, sig0=SIG0_V)[1:3]
    g = [torch.from_numpy(h).to(dev) for h in hist]
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    mats, dummies = [], []
    for k in range(a.handles):
        if a.jitter_mib and k:
            dummies.append(torch.empty(k * a.jitter_mib << 20, dtype=torch.uint8, device=dev))
        var = a.variants[k % len(a.variants)]
        kv = dict(x.split("=") for x in var.split(",") if x)
        os.environ.update(kv)
        m = JAXMaterial(mk())
        m.set_data_manager(n)
        for key in kv:
            del os.environ[key]
        m._variant = var
        m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        for _ in range(3):
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        mats.append(m)
    times = [[] for _ in mats]
    for _ in range(a.rounds):
        for k, m in enumerate(mats):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
            for e0, e1 in ev:
                e0.record()
                m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
                e1.record()
            torch.cuda.synchronize()
            times[k] += [e0.elapsed_time(e1) for e0, e1 in ev]
    # the J2-shaped streaming probe (tools/stream_mix.hip) on the SAME state memory, afterwards
    import ctypes as C
    sm = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    sm.stream_mix_j2_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    probe = []
    n64 = n // 64 * 64
    for m in mats:
        s0 = m._lib.dxm_state_ptr(m._handle, 0, 0, 0)
        s1 = m._lib.dxm_state_ptr(m._handle, 1, 0, 0)
        ldm = (m._lib.dxm_state_ptr(m._handle, 0, 1, 0) - s0) // 8
        fn = lambda: sm.stream_mix_j2_shape_launch(g[1].data_ptr(), s0, s1, ldm, flux.data_ptr(), ct.data_ptr(), n64, 1024, st or None)  # noqa: E731
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for e0, e1 in ev:
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        probe.append(round(float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])), 4))
    print(json.dumps({"eps": hex(g[1].data_ptr()), "flux": hex(flux.data_ptr()), "ct": hex(ct.data_ptr())}))
    for k, m in enumerate(mats):
        s0 = m._lib.dxm_state_ptr(m._handle, 0, 0, 0)
        s1 = m._lib.dxm_state_ptr(m._handle, 1, 0, 0)
        s0b = m._lib.dxm_state_ptr(m._handle, 0, 1, 0)
        print(json.dumps({"handle": k, "variant": m._variant, "s0": hex(s0), "s1": hex(s1), "slot_stride": s0b - s0,
                          "median_ms": round(float(np.median(times[k])), 4), "min_ms": round(float(np.min(times[k])), 4),
                          "probe_on_same_state_ms": probe[k]}), flush=True)


if __name__ == "__main__":
    main()
