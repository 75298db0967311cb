#!/usr/bin/env python3
"""The FeFp kernel against a no-arithmetic kernel with ITS memory shape, occupancy and compute gaps
(tools/stream_mix.hip::stream_mix_fefp_shape_kernel), interleaved in one process: is the kernel at the ceiling of its own
structure, and which change of structure would raise that ceiling?  Variants: residency (LDS cap), dependent-FMA gaps
standing for the per-point phase / the tangent evaluation of a round, points per round, loads issued ahead of the stores."""
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

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    n = 10_000_000 // 64 * 64
    dev = torch.device("cuda:0")
    lib = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    lib.stream_mix_fefp_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    gen = torch.Generator(device=dev).manual_seed(7)
    Fg = torch.randn((n, 9), generator=gen, device=dev, dtype=torch.float64) * (0.2 * 2e-2)
    Fg[:, 0] += 1 + 2e-2
    Fg[:, 1] += 1 - 1e-2
    Fg[:, 2] += 1 - 1e-2
    F0 = 0.5 * (Fg + torch.tensor([1.0, 1, 1, 0, 0, 0, 0, 0, 0], device=dev, dtype=torch.float64))
    m = JAXMaterial(jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=70e3, nu=0.3), jm.VoceHardening(500.0, 750.0, 1000.0)))
    m.set_data_manager(n)
    P = torch.empty((n, 9), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 81), dtype=torch.float64, device=dev)
    m.integrate_device(F0.data_ptr(), P.data_ptr(), ct.data_ptr(), st)
    m.data_manager.update()
    info = m.tune_placement(Fg.data_ptr(), P.data_ptr(), ct.data_ptr())
    ld = n + 32
    sa = torch.randn(7 * ld, dtype=torch.float64, device=dev)
    sb = torch.empty(13 * ld, dtype=torch.float64, device=dev)

    def shape(blocks, lds, pre, per_round, ppr, prefetch, state=0):
        return lambda: lib.stream_mix_fefp_shape_launch(Fg.data_ptr(), sa.data_ptr(), sb.data_ptr(), ld, P.data_ptr(), ct.data_ptr(), n,
                                                        blocks, lds, pre, per_round, ppr, prefetch, state, st or None)

    variants = {"fefp_kernel": lambda: m.integrate_device(Fg.data_ptr(), P.data_ptr(), ct.data_ptr(), st)}
    K70 = 70 * 1024
    for name, args in {
        "shape_uncapped": (2048, 0, 0, 0, 14, 0),
        "shape_8waves": (512, K70, 0, 0, 14, 0),
        "shape_8waves_prefetch": (512, K70, 0, 0, 14, 1),
        "shape_12waves": (768, 50 * 1024, 0, 0, 14, 0),
        "shape_16waves": (1024, 36 * 1024, 0, 0, 14, 0),
        # gaps: ~700 VALU of per-point phase + ~200 per round at 4 cycles each, as dependent FMAs of ~8 cycles
        "shape_8waves_gaps": (512, K70, 350, 100, 14, 0),
        "shape_8waves_gaps_prefetch": (512, K70, 350, 100, 14, 1),
        "shape_8waves_gaps_x2": (512, K70, 700, 200, 14, 0),
        "shape_8waves_gaps_half": (512, K70, 175, 50, 14, 0),
        "shape_8waves_gaps_ppr32": (512, K70, 350, 228, 32, 0),
        "shape_12waves_gaps": (768, 50 * 1024, 350, 100, 14, 0),
        "shape_12waves_gaps_x2": (768, 50 * 1024, 700, 200, 14, 0),
        "shape_16waves_gaps_x2": (1024, 36 * 1024, 700, 200, 14, 0),
        # the state moved tile-blocked as 16 B-per-lane accesses / not at all
        "blocked_state_uncapped": (2048, 0, 0, 0, 14, 0, 1),
        "blocked_state_8waves": (512, K70, 0, 0, 14, 0, 1),
        "blocked_state_8waves_gaps": (512, K70, 350, 100, 14, 0, 1),
        "blocked_state_8waves_gaps_prefetch": (512, K70, 350, 100, 14, 1, 1),
        # alignment of the tangent stores: 14 points per round start 112 / 96 / 80 / 64 B past a 128 B line, 16 points on a line
        "ppr16_8waves": (512, K70, 0, 0, 16, 0),
        "ppr16_8waves_gaps": (512, K70, 350, 125, 16, 0),
        "ppr16_uncapped": (2048, 0, 0, 0, 16, 0),
        "ppr14_aligned_8waves": (512, K70, 0, 0, 14, 0, 3),
        "ppr14_aligned_8waves_gaps": (512, K70, 350, 100, 14, 0, 3),
        "ppr14_aligned_uncapped": (2048, 0, 0, 0, 14, 0, 3),
        "ppr64_8waves": (512, K70, 0, 0, 64, 0),
        "no_state_ppr64_uncapped": (2048, 0, 0, 0, 64, 0, 2),
        "no_state_uncapped": (2048, 0, 0, 0, 14, 0, 2),
        "no_state_8waves": (512, K70, 0, 0, 14, 0, 2),
    }.items():
        variants[name] = shape(*args)
    times = {k: [] for k in variants}
    for r in range(10):
        for k, fn in variants.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[k].append(e0.elapsed_time(e1))
    out = {k: round(float(np.median(t)), 4) for k, t in times.items()}
    out["placement"] = info
    out["GBs_moved_kernel"] = round(952 * n / out["fefp_kernel"] / 1e6, 1)
    print(json.dumps(out), flush=True)
    m.close()


if __name__ == "__main__":
    main()
