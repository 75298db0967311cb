#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer form (what QuadratureMap.update() hands over: numpy
arrays in, numpy arrays out).  Reported in DESIGN.md and as `host_path` of the bench line; never
the headline `value`.

Modes:  r01      full 36-entry tangent and the ISVs downloaded in every call (round-1 behaviour:
                 option packed_transfer = 0, lazy_isv = False), 440 B/point over PCIe;
        packed   the 9 tangent coefficients moved and the 6x6 block rebuilt on the host, ISVs on
                 demand (default), 168 B/point;
        bound    packed + results delivered straight into caller-owned arrays (bind_outputs);
        coef     the caller takes the 9 coefficients themselves (tangent_layout="coef"): the transfer pipeline alone;
        pinned_in  bound + the strain array handed over in page-locked memory (what a caller that owns its gradient
                 buffer can do; QuadratureMap's gather makes a fresh pageable array per call)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


MIN_POINTS = None
PAGEABLE_DMA = False
FRESH_INPUT = False


LAW = "j2_linear"


def run(n, mode, reps, threads=None):
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, SIG0_F, SIGU_F, B_F, fefp_path, j2_history

    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    if LAW == "fefp":   # 9x9 tangent: 72 B up, 72 + 648 (r01) or 72 + 432 (packed: its 54 building blocks) down
        h = [None] + fefp_path(n, nsteps=4, eps=3e-2)[1:3]
        m = JAXMaterial(jm.FeFpJ2Plasticity(el, jm.VoceHardening(SIG0_F, SIGU_F, B_F)), lazy_isv=(mode != "r01"))
    else:
        h = j2_history(n)
        m = JAXMaterial(jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN)),
                        lazy_isv=(mode != "r01"), tangent_layout="coef" if mode == "coef" else "full")
    m.set_data_manager(n)
    if mode == "r01":
        m.set_option("packed_transfer", 0)
        m.set_option("max_chunks", 8)
    if threads:
        m.set_option("host_threads", threads)
    if MIN_POINTS is not None:
        m.set_option("packed_min_points", MIN_POINTS)
    if PAGEABLE_DMA:
        m.set_option("pageable_dma", 1)
    nf = 9 if LAW == "fefp" else 6
    if mode in ("bound", "pinned_in"):
        flux_fn, jac_fn = np.zeros(n * nf), np.zeros(n * nf * nf)
        m.bind_outputs(flux=flux_fn, tangent=jac_fn)
    m.integrate(h[1])
    m.data_manager.update()
    if mode == "pinned_in":
        from dolfinx_materials_amd._lib import PinnedArray

        pin = PinnedArray(h[2].shape)
        pin.array[:] = h[2]
        h[2] = pin.array
    m.integrate(h[2])
    ts = []
    for _ in range(reps):
        g = np.array(h[2]) if FRESH_INPUT else h[2]   # QuadratureMap.update builds a new gradient array per call
        t0 = time.perf_counter()
        m.integrate(g)
        ts.append(time.perf_counter() - t0)
        del g
    dt = float(np.median(ts))
    bpp = 48 + 48 + (56 + 288 if mode == "r01" else 72)
    if LAW == "fefp":
        bpp = 72 + 72 + (56 + 648 if mode == "r01" else 432)
    out = {"law": LAW, "fresh_input": FRESH_INPUT, "pageable_dma": PAGEABLE_DMA, "mode": mode, "points": n, "host_threads": threads or 8, "host_path_ms": round(dt * 1e3, 3),
           "min_ms": round(min(ts) * 1e3, 3), "Mpoints_per_s": round(n / dt / 1e6, 2), "pcie_bytes_per_point": bpp,
           "GBs_over_pcie": round(n * bpp / dt / 1e9, 2)}
    m.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, nargs="+", default=[100_000, 1_000_000, 10_000_000])
    ap.add_argument("--modes", nargs="+", default=["r01", "packed", "bound"])
    ap.add_argument("--threads", type=int, nargs="+", default=[0])
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--packed-min-points", type=int, default=None, help="option packed_min_points (library default 262144)")
    ap.add_argument("--law", default="j2_linear", choices=["j2_linear", "fefp"])
    ap.add_argument("--fresh-input", action="store_true", help="a newly allocated gradient array per call, as QuadratureMap.update hands over")
    ap.add_argument("--pageable-dma", action="store_true", help="option pageable_dma = 1: the runtime's pageable transfer path (faster, fragile)")
    a = ap.parse_args()
    global MIN_POINTS, LAW, PAGEABLE_DMA, FRESH_INPUT
    MIN_POINTS, LAW, PAGEABLE_DMA, FRESH_INPUT = a.packed_min_points, a.law, a.pageable_dma, a.fresh_input
    for n in a.points:
        for mode in a.modes:
            for t in a.threads:
                print(json.dumps(run(n, mode, a.reps, t or None)), flush=True)


if __name__ == "__main__":
    main()
