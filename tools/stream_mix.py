#!/usr/bin/env python3
"""Each constitutive kernel vs. a no-arithmetic streaming kernel that moves the same bytes per point
(two perfectly linear 16 B-per-lane streams), interleaved in one process on one box: how close is
the kernel to the ceiling of its own traffic mix?"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import bench
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    n = 10_000_000 // 64 * 64
    dev = torch.device("cuda:0")
    lib = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    lib.stream_mix_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.stream_mix_pipelined_launch.argtypes = lib.stream_mix_launch.argtypes
    lib.stream_mix_nt_launch.argtypes = lib.stream_mix_launch.argtypes
    st = torch.cuda.current_stream().cuda_stream
    el = jm.LinearElasticIsotropic(E=bench.E, nu=bench.NU)
    gen = torch.Generator(device=dev).manual_seed(7)
    from helpers import to_device

    hist = [to_device(h) for h in bench.history(n, 1234)]
    Fg = torch.randn((n, 9), generator=gen, device=dev, dtype=torch.float64) * (0.2 * 2e-2)
    Fg[:, 0] += 1 + 2e-2
    Fg[:, 1] += 1 - 1e-2
    Fg[:, 2] += 1 - 1e-2
    F0 = 0.5 * (Fg + torch.tensor([1.0, 1, 1, 0, 0, 0, 0, 0, 0], device=dev, dtype=torch.float64))
    cases = [
        # name, behaviour, (first increment, timed increment), bytes read / written per point
        ("elastic", jm.ElasticBehavior(el), (hist[1], hist[2]), 48, 336),
        ("j2_linear", jm.vonMisesIsotropicHardening(el, jm.LinearHardening(bench.SIG0, bench.H)), (hist[1], hist[2]), 104, 392),
        ("fefp_j2_voce", jm.FeFpJ2Plasticity(el, jm.VoceHardening(500.0, 750.0, 1000.0)), (F0, Fg), 128, 824),
    ]
    for name, beh, (g0, g1), rb, wb in cases:
        m = JAXMaterial(beh)
        m.set_data_manager(n)
        ng, nf = m._info.n_grad, m._info.n_flux
        flux = torch.empty((n, nf), dtype=torch.float64, device=dev)
        ct = torch.empty((n, nf * ng), dtype=torch.float64, device=dev)
        rbuf = torch.randn(n * rb // 8, dtype=torch.float64, device=dev)
        wbuf = torch.empty(n * wb // 8, dtype=torch.float64, device=dev)
        m.integrate_device(g0.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        variants = {name: lambda: m.integrate_device(g1.data_ptr(), flux.data_ptr(), ct.data_ptr(), st)}
        if name == "fefp_j2_voce":  # probe at the FeFp kernel's occupancy (2 workgroups per CU)
            lib.stream_mix_capped_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
            for blocks in (512, 1024):
                variants[f"capped8waves_{blocks}"] = (lambda b: (lambda: lib.stream_mix_capped_launch(rbuf.data_ptr(), wbuf.data_ptr(), n, rb, wb, b, 70 * 1024, st or None)))(blocks)
        if name == "j2_linear":
            lib.stream_mix_j2_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
            ld = n + 32
            sa = torch.randn(7 * ld, dtype=torch.float64, device=dev)
            sb = torch.empty(7 * ld, dtype=torch.float64, device=dev)
            for blocks in (1024, 2048):
                variants[f"shape17_{blocks}"] = (lambda b: (lambda: lib.stream_mix_j2_shape_launch(g1.data_ptr(), sa.data_ptr(), sb.data_ptr(), ld, flux.data_ptr(), ct.data_ptr(), n, b, st or None)))(blocks)
        for blocks in (1024, 2048, 4096):
            variants[f"probe_{blocks}"] = (lambda b: (lambda: lib.stream_mix_launch(rbuf.data_ptr(), wbuf.data_ptr(), n, rb, wb, b, st or None)))(blocks)
            variants[f"probe_nt_{blocks}"] = (lambda b: (lambda: lib.stream_mix_nt_launch(rbuf.data_ptr(), wbuf.data_ptr(), n, rb, wb, b, st or None)))(blocks)
            variants[f"probe_pipelined_{blocks}"] = (lambda b: (lambda: lib.stream_mix_pipelined_launch(rbuf.data_ptr(), wbuf.data_ptr(), n, rb, wb, b, st or None)))(blocks)
        times = {k: [] for k in variants}
        for r in range(12):
            for k, fn in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if r >= 2:
                    times[k].append(e0.elapsed_time(e1))
        med = {k: float(np.median(t)) for k, t in times.items()}
        probe = min(v for k, v in med.items() if k.startswith("probe_"))
        plain = min(v for k, v in med.items() if k.startswith("probe_") and "pipelined" not in k and "_nt_" not in k)
        nts = min(v for k, v in med.items() if k.startswith("probe_nt_"))
        shape = [v for k, v in med.items() if k.startswith("shape17_")]
        piped = min(v for k, v in med.items() if "pipelined" in k)
        moved = (rb + wb) * n
        print(json.dumps({
            "law": name, "bytes_moved_per_point": rb + wb, "kernel_ms": round(med[name], 4), "probe_ms": round(probe, 4),
            "kernel_GBs_moved": round(moved / med[name] / 1e6, 1), "probe_GBs": round(moved / probe / 1e6, 1),
            "kernel_over_probe": round(probe / med[name], 4),
            "probe_plain_ms": round(plain, 4), "probe_pipelined_ms": round(piped, 4), "probe_nt_stores_ms": round(nts, 4),
            "probe_17_streams_ms": round(min(shape), 4) if shape else None,
            "probe_at_8_waves_per_cu_ms": round(min([v for k, v in med.items() if k.startswith("capped8waves_")]), 4) if any(k.startswith("capped8waves_") for k in med) else None,
        }), flush=True)
        m.close()
        del flux, ct, rbuf, wbuf
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
