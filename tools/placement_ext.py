#!/usr/bin/env python3
"""EXPERIMENT (needs a library built with the DXM_STATE_EXTERNAL hook, not the shipped one):
real J2 kernel with the state placed at chosen offsets of a few big pool allocations."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DXM_LIB_PATH"] = os.path.join(ROOT, "dolfinx_materials_amd", "_jit", "ab", os.environ.get("EXT_LIB", "libdxmat_ext.so"))


def main():
    import torch

    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, j2_history

    dev = torch.device("cuda:0")
    n = 10_000_000
    el = jm.LinearElasticIsotropic(E=E, nu=NU)
    mk = lambda: jm.vonMisesIsotropicHardening(el, jm.LinearHardening(SIG0_LIN, H_LIN))  # noqa: E731
    hist = j2_history(n)[1:3]
    g = [torch.from_numpy(h).to(dev) for h in hist]
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ld = (n + 255) // 256 * 256 + 32
    half = 7 * ld * 8
    npools = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    pool_bytes = 4 << 30
    pools = [torch.zeros(pool_bytes, dtype=torch.uint8, device=dev) for _ in range(npools)]
    specs = []   # (label, s0 address, s1 address)
    for k in range(2):
        specs.append((f"hipMalloc#{k}", None, None))
    for pi, p in enumerate(pools):
        b = p.data_ptr()
        for off in (0, 1200 << 20, 2400 << 20):
            specs.append((f"pool{pi}+{off >> 20}MiB", b + off, b + off + half))
    for pi in range(npools - 1):
        specs.append((f"s0:pool{pi}+0 s1:pool{pi + 1}+3000MiB", pools[pi].data_ptr(), pools[pi + 1].data_ptr() + (3000 << 20)))
    # plain streaming inside each pool: is the region itself fast or slow?
    import ctypes as C
    sm = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    sm.stream_mix_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]

    def tm(fn, reps=10):
        for _ in range(3):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for e0, e1 in ev:
            e0.record()
            fn()
            e1.record()
        torch.cuda.synchronize()
        return round(float(np.median([e0.elapsed_time(e1) for e0, e1 in ev])), 4)

    for pi, p in enumerate(pools):
        b = p.data_ptr()
        v = p.view(torch.float64)
        half_el = v.numel() // 2
        r = {"pool": pi, "base": hex(b)}
        r["copy_2GiB_to_2GiB_ms"] = tm(lambda: v[half_el:].copy_(v[:half_el]))
        np_ = 4_000_000 // 64 * 64
        r["probe_104r_392w_4e6pts_ms"] = tm(lambda: sm.stream_mix_launch(b, b + (1 << 30), np_, 104, 392, 2048, st or None))   # reads 0.42 GB at b, writes 1.57 GB at b + 1 GiB
        nq = (3 << 30) // 256 // 64 * 64                      # 3 GiB of the 4 GiB pool
        r["probe_read_only_3GiB_ms"] = tm(lambda: sm.stream_mix_launch(b, b, nq, 256, 0, 2048, st or None))
        r["probe_write_only_3GiB_ms"] = tm(lambda: sm.stream_mix_launch(b, b, nq, 0, 256, 2048, st or None))
        print(json.dumps(r), flush=True)
    mats = []
    for label, a0, a1 in specs:
        os.environ.pop("DXM_STATE_EXTERNAL", None)
        os.environ.pop("DXM_STATE_EXTERNAL_S1", None)
        if a0 is not None:
            os.environ["DXM_STATE_EXTERNAL"] = hex(a0)
            os.environ["DXM_STATE_EXTERNAL_S1"] = hex(a1)
        m = JAXMaterial(mk())
        m.set_data_manager(n)
        m.integrate_device(g[0].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
        for _ in range(3):
            m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        rc, s = m.stats()
        assert rc == 0 and s["n_nan"] == 0
        mats.append(m)
    sm.stream_mix_j2_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    sm.stream_mix_state_only_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    d_eps = torch.randn((n, 6), dtype=torch.float64, device=dev)
    d_sig = torch.empty((n, 6), dtype=torch.float64, device=dev)
    d_ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    n64 = n // 64 * 64
    times = [[] for _ in mats]
    for _ in range(5):
        for k, m in enumerate(mats):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
            for e0, e1 in ev:
                e0.record()
                m.integrate_device(g[1].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
                e1.record()
            torch.cuda.synchronize()
            times[k] += [e0.elapsed_time(e1) for e0, e1 in ev]
    for k, m in enumerate(mats):
        p0 = m._lib.dxm_state_ptr(m._handle, 0, 0, 0)
        p1 = m._lib.dxm_state_ptr(m._handle, 1, 0, 0)
        r = {"state": specs[k][0], "s0": hex(p0), "s1": hex(p1), "median_ms": round(float(np.median(times[k])), 4)}
        r["shape_probe_real_buffers_ms"] = tm(lambda: sm.stream_mix_j2_shape_launch(g[1].data_ptr(), p0, p1, ld, flux.data_ptr(), ct.data_ptr(), n64, 1024, st or None))
        r["shape_probe_dummy_buffers_ms"] = tm(lambda: sm.stream_mix_j2_shape_launch(d_eps.data_ptr(), p0, p1, ld, d_sig.data_ptr(), d_ct.data_ptr(), n64, 1024, st or None))
        r["state_only_probe_ms"] = tm(lambda: sm.stream_mix_state_only_launch(p0, p1, ld, n64, 1024, st or None))
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
