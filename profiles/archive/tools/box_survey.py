#!/usr/bin/env python3
"""One record of what THIS box is: telemetry (tools/box_telemetry.py) + the bandwidth probes + the J2 kernel, each run as
a continuous stream of launches with the sysfs sampler beside it, then the driver's cadence (idle, 5 + 20 launches).

    python tools/box_survey.py [--seconds 1.2] [--out gpurun_out/box_survey.jsonl]

Per leg: per-launch HIP-event times (first five, median of the first tenth, median of the last half, minimum), the sampler's
min / median / max of sclk, mclk, fclk, power, busy, and -- for the stamped probe -- the shader clock under that load
(d s_memtime / d s_memrealtime x 100 MHz).  No placement search anywhere: every array is where its first allocation put it.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import box_telemetry as bt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.0, help="length of each continuous leg")
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--out", default="", help="default: gpurun_out/box_survey_<label>.json (one file per call: gpurun merges files back, it does not append)")
    ap.add_argument("--label", default="")
    args = ap.parse_args()

    snap0 = bt.snapshot(tools=True)   # before this process touches the GPU
    import torch

    import bench
    import dolfinx_materials_amd.materials as jm
    from dolfinx_materials_amd.jaxmat import JAXMaterial

    n = args.points // 64 * 64
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    st = torch.cuda.current_stream().cuda_stream
    lib = C.CDLL(os.path.join(ROOT, "tools", "libstreammix.so"))
    sig7 = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.stream_mix_nt_launch.argtypes = sig7
    lib.stream_mix_launch.argtypes = sig7
    lib.stream_mix_clock_launch.argtypes = sig7[:6] + [C.c_void_p, C.c_void_p]
    lib.stream_mix_j2_shape_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]

    hist = [bench.to_dev(h, dev) for h in bench.history(n, 1234)]
    flux = torch.empty((n, 6), dtype=torch.float64, device=dev)
    ct = torch.empty((n, 36), dtype=torch.float64, device=dev)
    m = JAXMaterial(jm.vonMisesIsotropicHardening(jm.LinearElasticIsotropic(E=bench.E, nu=bench.NU), jm.LinearHardening(bench.SIG0, bench.H)))
    m.set_data_manager(n)
    for k in range(2):
        m.integrate_device(hist[k].data_ptr(), flux.data_ptr(), ct.data_ptr(), st)
        m.data_manager.update()
    rb, wb = 104, 392
    rbuf = torch.randn(n * rb // 8, dtype=torch.float64, device=dev)
    wbuf = torch.empty(n * wb // 8, dtype=torch.float64, device=dev)
    ld = n + 32
    sa = torch.randn(7 * ld, dtype=torch.float64, device=dev)
    sb = torch.empty(7 * ld, dtype=torch.float64, device=dev)
    ca = torch.empty(1 << 27, dtype=torch.float64, device=dev).normal_()
    cb = torch.empty_like(ca)
    nblk = 2048
    big = torch.randn(n * (rb + wb) // 8, dtype=torch.float64, device=dev)
    stamps = torch.zeros(4 * nblk, dtype=torch.int64, device=dev)
    moved = (rb + wb) * n

    legs = {
        "j2_kernel": (lambda: m.integrate_device(hist[2].data_ptr(), flux.data_ptr(), ct.data_ptr(), st), moved),
        "stream_nt": (lambda: lib.stream_mix_nt_launch(rbuf.data_ptr(), wbuf.data_ptr(), n, rb, wb, nblk, st or None), moved),
        "stream_plain": (lambda: lib.stream_mix_launch(rbuf.data_ptr(), wbuf.data_ptr(), n, rb, wb, nblk, st or None), moved),
        "stream_j2_shape": (lambda: lib.stream_mix_j2_shape_launch(hist[2].data_ptr(), sa.data_ptr(), sb.data_ptr(), ld, flux.data_ptr(), ct.data_ptr(), n, nblk, st or None), moved),
        "stream_stamped": (lambda: lib.stream_mix_clock_launch(rbuf.data_ptr(), wbuf.data_ptr(), n, rb, wb, nblk, stamps.data_ptr(), st or None), moved),
        "torch_copy_1GiB": (lambda: cb.copy_(ca), 2 * ca.numel() * 8),
        # one direction only, same total bytes per launch: does the box differ in reads, in writes, or in the mix?
        "read_only": (lambda: lib.stream_mix_launch(big.data_ptr(), wbuf.data_ptr(), n, rb + wb, 0, nblk, st or None), moved),
        "write_only_nt": (lambda: lib.stream_mix_nt_launch(rbuf.data_ptr(), big.data_ptr(), n, 0, rb + wb, nblk, st or None), moved),
        "write_only_plain": (lambda: lib.stream_mix_launch(rbuf.data_ptr(), big.data_ptr(), n, 0, rb + wb, nblk, st or None), moved),
    }

    def continuous(fn, seconds, max_launches=4000):
        """Back-to-back launches for `seconds`, one event pair per launch, no host work in between."""
        ev = []
        t0 = time.perf_counter()
        batch = 50
        while time.perf_counter() - t0 < seconds and len(ev) < max_launches:
            for _ in range(batch):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                ev.append((a, b))
            # keep the queue shallow enough that wall time tracks device time (the queue never drains)
            ev[-batch // 2][1].synchronize()
        torch.cuda.synchronize()
        return np.array([a.elapsed_time(b) for a, b in ev])

    def leg_record(name, fn, nbytes, seconds):
        torch.cuda.synchronize()
        m0 = bt.metrics()
        with bt.Sampler(period_s=0.005) as s:
            ts = continuous(fn, seconds)
        m1 = bt.metrics()
        k = len(ts)
        steady = float(np.median(ts[k // 2:]))
        rec = {"launches": k, "first5_ms": [round(float(x), 4) for x in ts[:5]], "first_tenth_median_ms": round(float(np.median(ts[: max(1, k // 10)])), 4),
               "steady_ms": round(steady, 4), "min_ms": round(float(ts.min()), 4), "p95_ms": round(float(np.percentile(ts, 95)), 4),
               "GBs": round(nbytes / steady / 1e6, 1), "sampler": s.summary(), "firmware": bt.metrics_delta(m0, m1)}
        if name == "stream_stamped":
            hs = stamps.cpu().numpy().reshape(-1, 4).astype(np.float64)
            dc, dr = hs[:, 2] - hs[:, 0], hs[:, 3] - hs[:, 1]
            ok = dr > 0
            if ok.any():
                mhz = dc[ok] / dr[ok] * 100.0
                rec["shader_clock_under_load_mhz"] = [round(float(np.percentile(mhz, q))) for q in (5, 50, 95)]
                rec["workgroup_lifetime_us_median"] = round(float(np.median(dr[ok])) / 100.0, 1)
        return rec

    try:
        pr = torch.cuda.get_device_properties(0)
        hip_pci = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        hip_pci = None
    out = {"label": args.label, "hip_pci_bus_id": hip_pci, "time": time.strftime("%Y-%m-%dT%H:%M:%S"), "points": n, "box": bt.condensed(snap0), "legs": {}}
    # a settling leg first so that every later leg starts from a busy chip
    continuous(legs["stream_nt"][0], 0.5)
    for name, (fn, nbytes) in legs.items():
        out["legs"][name] = leg_record(name, fn, nbytes, args.seconds)
    # once more at the end: did the box drift while we measured?
    out["legs"]["j2_kernel_again"] = leg_record("j2_kernel", legs["j2_kernel"][0], moved, args.seconds)

    # the driver's cadence: host idle, then 5 warm-up + 20 timed launches, each with its own event pair
    cad = {}
    fn = legs["j2_kernel"][0]
    for idle in (0.0, 0.05, 0.5, 2.0, 5.0):
        continuous(fn, 0.3)
        before = bt.fast_read(bt.my_card()) if bt.my_card() else {}
        time.sleep(idle)
        after_idle = bt.fast_read(bt.my_card()) if bt.my_card() else {}
        ev = []
        for _ in range(25):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        ts = np.array([a.elapsed_time(b) for a, b in ev])
        cad[f"idle_{idle}s"] = {"warmup5_ms": [round(float(x), 4) for x in ts[:5]], "timed20_mean_ms": round(float(ts[5:].mean()), 4),
                                "timed20_median_ms": round(float(np.median(ts[5:])), 4), "timed20_max_ms": round(float(ts[5:].max()), 4),
                                "clocks_busy": {k: before.get(k) for k in ("sclk", "mclk", "fclk", "power_w")},
                                "clocks_after_idle": {k: after_idle.get(k) for k in ("sclk", "mclk", "fclk", "power_w")}}
    out["driver_cadence"] = cad
    out["j2_steady_frac_of_peak"] = round(moved / out["legs"]["j2_kernel"]["steady_ms"] / 1e6 / 8000.0, 4)
    out["j2_over_stream_nt"] = round(out["legs"]["stream_nt"]["steady_ms"] / out["legs"]["j2_kernel"]["steady_ms"], 4)
    out["j2_over_stream_j2_shape"] = round(out["legs"]["stream_j2_shape"]["steady_ms"] / out["legs"]["j2_kernel"]["steady_ms"], 4)
    snap1 = bt.snapshot(tools=False)
    out["box_after"] = {k: v for k, v in bt.condensed(snap1).items() if k in ("sclk", "mclk", "fclk", "power_w", "temp_c", "hbm_temp_c", "vram_used", "gpu_busy", "mem_busy")}
    out["metrics_end"] = bt.metrics()
    out["snapshot_full"] = snap0
    m.close()
    args.out = args.out or os.path.join(ROOT, "gpurun_out", f"box_survey_{args.label or int(time.time())}.json")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "a") as f:
        f.write(json.dumps(out) + "\n")
    brief = {k: out[k] for k in ("box", "j2_steady_frac_of_peak", "j2_over_stream_nt", "j2_over_stream_j2_shape")}
    brief["legs"] = {k: {kk: v[kk] for kk in ("steady_ms", "GBs", "first5_ms", "sampler", "shader_clock_under_load_mhz") if kk in v} for k, v in out["legs"].items()}
    brief["driver_cadence"] = cad
    print(json.dumps(brief, indent=1))


if __name__ == "__main__":
    main()
