#!/usr/bin/env python3
"""Why was `host_path.value` (the SAME strain array every call) slower than `new_strain_array_every_call` in the r03 driver line?

Legs: option register_input forced to 0 (always staged), 2 (always page-locked for the call) and 1 (adaptive), each with the same
array every call and with a new array every call, bound outputs, 30 calls each; per leg the per-call times (ms), the median
after the handle's probing calls (calls 1-5), what each call did (`last_upload`), and what page-locking THAT array costs
(`dxm_host_register` timed directly, with the fraction of the array that sits on transparent huge pages from
/proc/self/smaps).  One JSON line per leg."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def thp_fraction(a):
    """Fraction of the array's bytes backed by anonymous huge pages (AnonHugePages of the mapping that holds it)."""
    addr, end = a.ctypes.data, a.ctypes.data + a.nbytes
    try:
        cur = None
        size = huge = 0
        for line in open("/proc/self/smaps"):
            if "-" in line.split(" ")[0] and line[0] in "0123456789abcdef":
                lo, hi = (int(x, 16) for x in line.split(" ")[0].split("-"))
                cur = (lo, hi) if (lo < end and hi > addr) else None
            elif cur and line.startswith("Size:"):
                size += int(line.split()[1])
            elif cur and line.startswith("AnonHugePages:"):
                huge += int(line.split()[1])
        return round(huge / size, 3) if size else None
    except Exception:
        return None


def numa_nodes(a):
    """Pages of the array per NUMA node, from /proc/self/numa_maps (the mapping that holds it): {"N0": pages, ...}."""
    addr = a.ctypes.data
    try:
        best = None
        for line in open("/proc/self/numa_maps"):
            parts = line.split()
            lo = int(parts[0], 16)
            if lo <= addr and (best is None or lo > best[0]):
                best = (lo, parts)
        out = {}
        for tok in best[1][1:]:
            if tok[0] == "N" and "=" in tok:
                k, v = tok.split("=")
                out[k] = int(v)
            elif tok.startswith("kernelpagesize_kB="):
                out["pagesize_kB"] = int(tok.split("=")[1])
        return out
    except Exception as exc:
        return {"error": repr(exc)}


def main():
    import bench
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd import _lib
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    calls = 30
    lib = _lib.load()
    h = bench.history(n, 1234)

    def lock_ms(a):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            rc = lib.dxm_host_register(a.ctypes.data, a.nbytes)
            ts.append((time.perf_counter() - t0) * 1e3)
            if rc == 0:
                t0 = time.perf_counter()
                lib.dxm_host_unregister(a.ctypes.data)
                ts[-1] = (ts[-1], (time.perf_counter() - t0) * 1e3)
        return [(round(x[0], 2), round(x[1], 2)) if isinstance(x, tuple) else round(x, 2) for x in ts]

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import box_telemetry as bt

    gpu_node = (bt.condensed(bt.snapshot(tools=False)) or {}).get("numa_node")
    print(json.dumps({"gpu_numa_node": gpu_node, "cpu_now": os.sched_getcpu() if hasattr(os, "sched_getcpu") else None,
                      "affinity_cpus": len(os.sched_getaffinity(0))}), flush=True)
    print(json.dumps({"array": "h[1] (allocated once, early)", "thp_fraction": thp_fraction(h[1]), "numa": numa_nodes(h[1]), "register_unregister_ms": lock_ms(h[1])}), flush=True)
    fresh = np.array(h[1])
    print(json.dumps({"array": "np.array(h[1]) (allocated now)", "thp_fraction": thp_fraction(fresh), "numa": numa_nodes(fresh), "register_unregister_ms": lock_ms(fresh)}), flush=True)
    del fresh

    recopied = np.array(h[1])   # the same content in pages first touched NOW: "same array every call", placed like a fresh one
    for mode in (0, 2, 1):
        for new_array in (False, True, "recopied", "rewritten"):
            m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=bench.E, nu=bench.NU), jm.LinearHardening(bench.SIG0, bench.H)))
            m.set_data_manager(n)
            m.set_option("register_input", mode)
            flux_fn, jac_fn = np.zeros(n * 6), np.zeros(n * 36)
            m.bind_outputs(flux=flux_fn, tangent=jac_fn)
            m.integrate(h[0])
            m.data_manager.update()
            ts, ups = [], []
            for k in range(calls):
                g = recopied if new_array in ("recopied", "rewritten") else (np.array(h[1]) if new_array else h[1])
                if new_array == "rewritten":   # what Expression.eval(values=...) does to a bound gradient Function before every update
                    g[...] = h[1]
                t0 = time.perf_counter()
                m.integrate(g)
                ts.append((time.perf_counter() - t0) * 1e3)
                ups.append({"dma (page-locked for the call)": "R", "staged through the ring": "S"}.get(m.last_upload, "?"))
                del g
            m.close()
            print(json.dumps({"register_input": mode, "new_array_every_call": new_array, "numa_of_last_array": None, "median_ms_after_call_5": round(float(np.median(ts[5:])), 2),
                              "min_ms": round(min(ts), 2), "per_call_ms": [round(t, 1) for t in ts], "upload_per_call": "".join(ups)}), flush=True)


if __name__ == "__main__":
    main()
