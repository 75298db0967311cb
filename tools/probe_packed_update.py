#!/usr/bin/env python3
"""Measurement (round 6): where the time of one accelerated ``update()`` goes once no tangent block is rebuilt on the host.

  * QuadratureFieldMap at N points: layout in {full, pack4} x isv_every_update in {True, "lazy"}: ms per update;
  * the raw device-to-host rate of this box for the bytes of one pack4 update (136 B/point) into (a) memory from
    ``dxm_host_alloc`` (hipHostMalloc), (b) numpy memory page-locked in place (``dxm_host_register``: what a bound Function
    is), as one transfer and as 64 x 4 chunked transfers on two streams, alone and with the 48 B/point upload running.
Prints one JSON line per figure."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


SPLIT_CHUNKS = (24, 16)


def update_legs(n, reps=11):
    """``integrate`` on the bound Function memory of an accelerated map (what ``update()`` spends its time in), for layout x ISV mode x
    stream scheme; the two stream schemes alternate call by call on ONE handle."""
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.field_map import QuadratureFieldMap
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_LIN, H_LIN, j2_history

    nqp = 8
    ncell = n // nqp
    npts = ncell * nqp
    h = j2_history(npts)
    for layout in ("full", "pack4"):
        for mode in (True, "lazy"):
            m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=E, nu=NU), jm.LinearHardening(SIG0_LIN, H_LIN)), tangent_layout=layout)
            q = QuadratureFieldMap(ncell, nqp, m)
            q.isv_every_update = mode
            now = {"k": 0}
            q.register_gradient("strain", lambda c: h[now["k"]].reshape(ncell, nqp * 6)[c])
            q.update()
            q.advance()
            now["k"] = 1
            q.update()
            g = q.gradients["strain"].function.x.array.reshape(npts, 6)
            variants = [("alternating", {"split_streams": 0})] + [(f"split_{k}", {"split_streams": 1, "max_chunks": k}) for k in SPLIT_CHUNKS]
            base = {"max_chunks": 64, "split_streams": 1}
            ts = {name: [] for name, _ in variants}
            for r in range(reps + 2):
                for name, opts in variants:
                    for k, v in {**base, **opts}.items():
                        m.set_option(k, v)
                    t0 = time.perf_counter()
                    m.integrate(g)
                    if r >= 2:
                        ts[name].append(time.perf_counter() - t0)
            for k, v in base.items():
                m.set_option(k, v)
            print(json.dumps({"leg": "integrate", "layout": layout, "isv_every_update": mode, "points": npts,
                              "ms_min_med": {name: [round(min(v) * 1e3, 2), round(float(np.median(v)) * 1e3, 2)] for name, v in ts.items()}}), flush=True)
            if layout == "pack4" and mode is True:
                m.set_option("verbose", 1)
                m.integrate(g)
                m.set_option("verbose", 0)
            q.close()
            m.close()


def fefp_legs(n, reps=7):
    """The finite-strain law in the host-buffer form with bound (page-locked) arrays: 72 B/point up, 72 + 432 down (PK1 + the 54
    building blocks of the 9x9 tangent, rebuilt by the workers), split against alternating streams."""
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial
    from helpers import E, NU, SIG0_F, SIGU_F, B_F, fefp_path

    path = fefp_path(n, nsteps=4, eps=3e-2)
    m = JAXMaterial(jm.FeFpJ2Plasticity(jm.LinearElasticIsotropic(E=E, nu=NU), jm.VoceHardening(SIG0_F, SIGU_F, B_F)))
    m.set_data_manager(n)
    flux_fn, jac_fn, grad_fn = np.zeros(n * 9), np.zeros(n * 81), np.zeros(n * 9)
    m.bind_outputs(flux=flux_fn, tangent=jac_fn)
    m.bind_inputs(gradient=grad_fn)
    g = grad_fn.reshape(n, 9)
    g[...] = path[1]
    m.integrate(g)
    m.data_manager.update()
    g[...] = path[2]
    variants = [("alternating", {"split_streams": 0}), ("split", {"split_streams": 1}), ("split_16", {"split_streams": 1, "max_chunks": 16})]
    base = {"max_chunks": 64, "split_streams": 1}
    ts = {name: [] for name, _ in variants}
    for r in range(reps + 2):
        for name, opts in variants:
            for k, v in {**base, **opts}.items():
                m.set_option(k, v)
            t0 = time.perf_counter()
            m.integrate(g)
            if r >= 2:
                ts[name].append(time.perf_counter() - t0)
    print(json.dumps({"leg": "integrate_fefp", "points": n, "ms_min_med": {name: [round(min(v) * 1e3, 2), round(float(np.median(v)) * 1e3, 2)] for name, v in ts.items()}}), flush=True)
    m.close()


def raw_rates(n):
    import torch

    from dolfinx_materials_amd import _lib

    lib = _lib.load()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    D2H, H2D = 2, 1
    dev = torch.device("cuda:0")
    nbytes = 136 * n
    up = 48 * n
    d = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
    d_up = torch.empty(up // 8, dtype=torch.float64, device=dev)
    s = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
    own = _lib.PinnedArray((nbytes // 8,))
    own_up = _lib.PinnedArray((up // 8,))
    own_up.array[...] = 1.0
    reg = np.zeros(nbytes // 8)
    assert lib.dxm_host_register(reg.ctypes.data, reg.nbytes) == 0
    for name, host in (("hipHostMalloc", own.array), ("registered_numpy", reg)):
        for chunks in (1, 64):
            for with_upload in (False, True):
                ts = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    if with_upload:
                        hip.hipMemcpyAsync(d_up.data_ptr(), own_up.array.ctypes.data, up, H2D, s[2].cuda_stream)
                    step = (nbytes // chunks) // 8 * 8
                    for c in range(chunks):
                        off = c * step
                        cnt = step if c < chunks - 1 else nbytes - off
                        # four transfers per chunk like the pipeline (flux, two state fields, coefficients): 48 + 8 + 48 + 32 of 136
                        o = off
                        for part in (48, 8, 48, 32):
                            sz = cnt * part // 136 // 8 * 8 if part != 32 else off + cnt - o
                            hip.hipMemcpyAsync(host.ctypes.data + o, d.data_ptr() + o, sz, D2H, s[c & 1].cuda_stream)
                            o += sz
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                print(json.dumps({"leg": "raw_d2h", "host_memory": name, "chunks": chunks, "upload_running": with_upload, "bytes": nbytes,
                                  "ms_min_med": [round(min(ts) * 1e3, 2), round(float(np.median(ts)) * 1e3, 2)], "GBs_best": round(nbytes / min(ts) / 1e9, 1)}), flush=True)
    # the pipeline's own pattern, piece by piece, on raw HIP calls: two streams, 64 chunks; per chunk [H2D of its 48 B/point] ->
    # [a small kernel: hipMemsetAsync] -> [its four downloads], all on the chunk's stream
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    scratch = torch.empty(1 << 20, dtype=torch.float64, device=dev)
    host = reg
    for pattern in ("d2h_only", "h2d_then_d2h", "h2d_kernel_d2h", "kernel_d2h"):
        for chunks in (64, 32):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step = (nbytes // chunks) // 8 * 8
                ustep = (up // chunks) // 8 * 8
                for c in range(chunks):
                    st = s[c & 1].cuda_stream
                    off = c * step
                    cnt = step if c < chunks - 1 else nbytes - off
                    if pattern.startswith("h2d"):
                        hip.hipMemcpyAsync(d_up.data_ptr() + c * ustep, own_up.array.ctypes.data + c * ustep, ustep, H2D, st)
                    if "kernel" in pattern:
                        hip.hipMemsetAsync(scratch.data_ptr(), 0, 1 << 16, st)
                    o = off
                    for part in (48, 8, 48, 32):
                        sz = cnt * part // 136 // 8 * 8 if part != 32 else off + cnt - o
                        hip.hipMemcpyAsync(host.ctypes.data + o, d.data_ptr() + o, sz, D2H, st)
                        o += sz
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print(json.dumps({"leg": "raw_pattern", "pattern": pattern, "chunks": chunks, "bytes_down": nbytes, "ms_min_med": [round(min(ts) * 1e3, 2), round(float(np.median(ts)) * 1e3, 2)],
                              "GBs_down_best": round(nbytes / min(ts) / 1e9, 1)}), flush=True)
    # uploads (+ a small kernel + an event per chunk) on their own stream, the downloads of chunk c issued on two other streams
    # (a) behind a device-side hipStreamWaitEvent, (b) by the host once hipEventSynchronize(event c) returns
    hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    events = []
    for _ in range(64):
        e = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0   # hipEventDisableTiming
        events.append(e)
    for gate in ("device_wait", "host_wait"):
        for chunks in (64, 32):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                step = (nbytes // chunks) // 8 * 8
                ustep = (up // chunks) // 8 * 8
                su = s[2].cuda_stream
                for c in range(chunks):
                    hip.hipMemcpyAsync(d_up.data_ptr() + c * ustep, own_up.array.ctypes.data + c * ustep, ustep, H2D, su)
                    hip.hipMemsetAsync(scratch.data_ptr(), 0, 1 << 16, su)
                    hip.hipEventRecord(events[c], su)
                for c in range(chunks):
                    st = s[c & 1].cuda_stream
                    if gate == "device_wait":
                        hip.hipStreamWaitEvent(st, events[c], 0)
                    else:
                        hip.hipEventSynchronize(events[c])
                    off = c * step
                    cnt = step if c < chunks - 1 else nbytes - off
                    o = off
                    for part in (48, 8, 48, 32):
                        sz = cnt * part // 136 // 8 * 8 if part != 32 else off + cnt - o
                        hip.hipMemcpyAsync(host.ctypes.data + o, d.data_ptr() + o, sz, D2H, st)
                        o += sz
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            print(json.dumps({"leg": "raw_pattern", "pattern": "uploads_on_their_own_stream_" + gate, "chunks": chunks, "bytes_down": nbytes,
                              "ms_min_med": [round(min(ts) * 1e3, 2), round(float(np.median(ts)) * 1e3, 2)], "GBs_down_best": round(nbytes / min(ts) / 1e9, 1)}), flush=True)
    lib.dxm_host_unregister(reg.ctypes.data)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    if "--no-raw" not in sys.argv:
        raw_rates(n)
    if "--sweep" in sys.argv:
        SPLIT_CHUNKS = (64, 4, 8, 12, 16, 24)   # 64: the library's own cap for the batch size decides
    if "--fefp" in sys.argv:
        fefp_legs(n)
    elif "--raw-only" not in sys.argv:
        update_legs(n)
