"""Where the caller's arrays sit relative to the GPU (2-socket hosts): NUMA node of a numpy array's pages against that of the
library's own page-locked memory, and what the host-buffer call costs for each combination of {array allocated on node k} x
{page-locked for the call and uploaded by DMA | staged through the ring by the worker threads | already page-locked (hipHostMalloc)}.

    python tools/numa_probe.py [--points 10000000] [--reps 5]
"""
import argparse
import ctypes as C
import glob
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

libc = C.CDLL(None, use_errno=True)
SYS_move_pages = 279


def node_of(arr, samples=16):
    """NUMA node of `samples` pages spread over the array (move_pages with no target nodes = query)."""
    base, n = arr.ctypes.data, arr.nbytes
    addrs = [(base + (n * k) // samples) & ~4095 for k in range(samples)]
    pages = (C.c_void_p * samples)(*addrs)
    status = (C.c_int * samples)()
    rc = libc.syscall(SYS_move_pages, 0, C.c_ulong(samples), pages, None, status, 0)
    if rc != 0:
        return f"move_pages failed (errno {C.get_errno()})"
    vals = sorted(set(status))
    return vals[0] if len(vals) == 1 else vals


def cpus_of_node(k):
    txt = open(f"/sys/devices/system/node/node{k}/cpulist").read().strip()
    out = set()
    for part in txt.split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd._lib import PinnedArray
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, j2_history

    nodes = sorted(int(p.rsplit("node", 1)[1]) for p in glob.glob("/sys/devices/system/node/node[0-9]*"))
    allowed = os.sched_getaffinity(0)
    gpu_nodes = {p: open(p).read().strip() for p in glob.glob("/sys/class/drm/card*/device/numa_node")}
    print(json.dumps({"nodes": nodes, "allowed_cpus": len(allowed), "allowed_per_node": {k: len(allowed & cpus_of_node(k)) for k in nodes},
                      "gpu_numa_node_sysfs": gpu_nodes, "cpu_now": os.sched_getcpu() if hasattr(os, "sched_getcpu") else None}), flush=True)
    n = a.points
    h = j2_history(n)
    pin = PinnedArray(h[1].shape)
    pin.array[...] = h[1]
    print(json.dumps({"library_page_locked_memory_on_node": node_of(pin.array)}), flush=True)
    for out_node in nodes:
        # results into arrays allocated (first touched) on node `out_node`, then bound
        os.sched_setaffinity(0, allowed & cpus_of_node(out_node) or allowed)
        flux_fn, jac_fn = np.zeros(n * 6), np.zeros(n * 36)
        flux_fn[:] = 1.0
        jac_fn[:] = 1.0
        os.sched_setaffinity(0, allowed)
        m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)))
        m.set_data_manager(n)
        m.bind_outputs(flux=flux_fn, tangent=jac_fn)
        m.integrate(h[0])
        m.data_manager.update()
        m.integrate(h[1])
        for in_node in nodes:
            os.sched_setaffinity(0, allowed & cpus_of_node(in_node) or allowed)
            g = np.array(h[1])
            os.sched_setaffinity(0, allowed)
            for mode, src in (("registered_dma", g), ("staged_by_workers", g), ("already_page_locked", pin.array)):
                m.set_option("register_input", 0 if mode == "staged_by_workers" else 1)
                m.integrate(src)
                ts = []
                for _ in range(a.reps):
                    t0 = time.perf_counter()
                    m.integrate(src)
                    ts.append(time.perf_counter() - t0)
                print(json.dumps({"outputs_on_node": node_of(jac_fn), "strain_on_node": node_of(src), "upload": mode, "last_upload": m.last_upload,
                                  "ms_median": round(float(np.median(ts)) * 1e3, 2), "ms_min": round(min(ts) * 1e3, 2)}), flush=True)
            del g
        m.close()
        del flux_fn, jac_fn


if __name__ == "__main__":
    main()
