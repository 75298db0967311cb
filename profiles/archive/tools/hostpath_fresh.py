#!/usr/bin/env python3
"""Where the time goes when the strain array of the host-buffer form is NEW in every call (what QuadratureMap.update
hands over, quadrature_map.py:304-313) -- DESIGN.md section 7.  Variants of how the caller's array came to be, each with the
library's timeline (option tune_verbose) for one call and the median of `reps` calls:
  same        the same numpy array every call
  fresh       np.array(src) right before the call, freed after it (bench.py host_path.new_strain_array_every_call)
  fresh_keep  np.array(src) right before the call, all of them kept alive (no munmap between calls)
  fresh_early the new array is made BEFORE the previous call's results are consumed (one call of distance)
  hugepage    fresh + madvise(MADV_HUGEPAGE) on the new range before it is written
  unrelated_munmap  the same array every call, but a 480 MB scratch array is made and freed before each call
  fresh_sleep fresh + 100 ms of sleep between freeing the previous array / making the new one and the call
(--own-outputs: results into the material's own page-locked arrays instead of bound, registered caller arrays;
 NUMPY_MADVISE_HUGEPAGE=0 in the environment: numpy does not ask for transparent huge pages)
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
libc = ctypes.CDLL("libc.so.6", use_errno=True)


def new_array(src, variant):
    if variant == "hugepage":
        g = np.empty_like(src)
        a = g.ctypes.data
        lo = (a + (2 << 20) - 1) & ~((2 << 20) - 1)
        hi = (a + g.nbytes) & ~((2 << 20) - 1)
        if hi > lo:
            libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), 14)   # MADV_HUGEPAGE
        g[...] = src
        return g
    return np.array(src)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--variants", nargs="+", default=["same", "fresh", "fresh_keep", "hugepage", "same"])
    ap.add_argument("--option", nargs=2, action="append", default=[], metavar=("NAME", "VALUE"))
    ap.add_argument("--verbose-call", action="store_true")
    ap.add_argument("--own-outputs", action="store_true")
    ap.add_argument("--verbose-all", action="store_true", help="library timeline (stderr) for every call, with the call's wall time")
    a = ap.parse_args()
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, j2_history

    n = a.points
    h = j2_history(n)
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)))
    m.set_data_manager(n)
    for k, v in a.option:
        m.set_option(k, float(v))
    if not a.own_outputs:
        flux_fn, jac_fn = np.zeros(n * 6), np.zeros(n * 36)
        m.bind_outputs(flux=flux_fn, tangent=jac_fn)
    m.integrate(h[0])
    m.data_manager.update()
    m.integrate(h[1])
    keep = []
    for spec in a.variants:
        variant, _, opts = spec.partition(":")
        applied = {}
        for kv in filter(None, opts.split(",")):
            k_, v_ = kv.split("=")
            m.set_option(k_, float(v_))
            applied[k_] = float(v_)
        ts, tmake = [], []
        for r in range(a.reps + (1 if a.verbose_call else 0)):
            t0 = time.perf_counter()
            if variant == "unrelated_munmap":
                tmp = np.array(h[1])
                del tmp
            if variant == "pinned":   # the caller's array is page-locked (dxm_host_alloc): DMA upload, no staging
                if "pin" not in globals():
                    from dolfinx_materials_amd._lib import PinnedArray
                    globals()["pin"] = PinnedArray(h[1].shape)
                    pin.array[...] = h[1]
                g = pin.array
            else:
                g = h[1] if variant in ("same", "unrelated_munmap") else new_array(h[1], variant)
            tmake.append(time.perf_counter() - t0)
            if variant == "fresh_sleep":
                time.sleep(0.1)
            verbose = (a.verbose_call and r == a.reps) or a.verbose_all
            if verbose:
                m.set_option("tune_verbose", 1)
            t0 = time.perf_counter()
            m.integrate(g)
            dt = time.perf_counter() - t0
            if verbose:
                m.set_option("tune_verbose", 0)
                print(f"[{variant} rep {r}] call took {dt * 1e3:.2f} ms", file=sys.stderr, flush=True)
            if not verbose or a.verbose_all:
                ts.append(dt)
            if variant == "fresh_keep":
                keep.append(g)
            del g
        print(json.dumps({"variant": variant, "variant_options": applied, "own_outputs": a.own_outputs, "thp_env": os.environ.get("NUMPY_MADVISE_HUGEPAGE"), "points": n, "options": dict(a.option), "ms_median": round(float(np.median(ts)) * 1e3, 2),
                          "ms_min": round(min(ts) * 1e3, 2), "ms_all": [round(t * 1e3, 1) for t in ts],
                          "ms_to_make_the_array": round(float(np.median(tmake)) * 1e3, 1)}), flush=True)
        del keep[:]
    m.close()


if __name__ == "__main__":
    main()
