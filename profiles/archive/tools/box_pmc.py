#!/usr/bin/env python3
"""Behind the L2 on THIS lease: hardware counters of the headline kernel (first-allocation placement, `bench.py --pmc-child`:
six launches at 1e7 points) that could tell a slow lease from a fast one -- address translation (UTCL1 hits / misses, UTCL2
credit stalls), the L2's DRAM interface (credit stalls, queue levels, write stalls, tag stalls) and request latencies.  Counters
only besides --kernel-trace, one rocprofv3 pass per group, the program directly after `--`; run BEFORE anything else touches the GPU.

    python tools/box_pmc.py [--label box20] [--out gpurun_out/box_pmc_<label>.json]
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import box_telemetry as bt  # noqa: E402

GROUPS = [
    ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"],
    ["TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_TAG_STALL_sum"],
    ["TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_RDREQ_sum"],
    ["TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"],
    ["TCC_BUSY_sum", "TCC_CYCLE_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
    ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"],
]


def one_pass(counters, kernel="small_strain_kernel<1"):
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="dxm_boxpmc_", dir="/tmp")
    try:
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", "--points", "10000000", "--law", "j2_linear"]
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        vals, dur = {}, []
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel in row["Kernel_Name"]:
                    vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for f in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel in row["Kernel_Name"]:
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
        # the last launches are the steady ones (the child runs 2 set-up increments, then 6 launches of increment 3)
        out = {k: sum(v[-6:]) / len(v[-6:]) for k, v in vals.items()}
        out["kernel_ms_in_this_pass"] = round(sum(dur[-6:]) / max(1, len(dur[-6:])), 4) if dur else None
        if r.returncode != 0 or not vals:
            out["error"] = f"rc {r.returncode}: {r.stderr[-200:]}"
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--label", default="")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    box = bt.condensed(bt.snapshot(tools=False))
    rec = {"label": a.label, "time": time.strftime("%Y-%m-%dT%H:%M:%S"), "box": {k: box.get(k) for k in ("unique_id", "pci", "vbios", "vram_used", "vram_of_kfd_processes_on_my_gpu")},
           "passes": []}
    for g in GROUPS:
        rec["passes"].append(one_pass(g))
    flat = {}
    for p in rec["passes"]:
        flat.update({k: v for k, v in p.items() if k not in ("kernel_ms_in_this_pass", "error")})
    ms = [p["kernel_ms_in_this_pass"] for p in rec["passes"] if p.get("kernel_ms_in_this_pass")]
    rec["kernel_ms_mean_over_passes"] = round(sum(ms) / len(ms), 4) if ms else None
    d = {}
    if flat.get("TCP_UTCL1_REQUEST_sum"):
        d["utcl1_miss_per_request"] = flat.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0) / flat["TCP_UTCL1_REQUEST_sum"]
    if flat.get("TCP_TCC_READ_REQ_sum"):
        d["read_latency_cycles_per_request"] = flat.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / flat["TCP_TCC_READ_REQ_sum"]
    if flat.get("TCP_TCC_WRITE_REQ_sum"):
        d["write_latency_cycles_per_request"] = flat.get("TCP_TCC_WRITE_REQ_LATENCY_sum", 0) / flat["TCP_TCC_WRITE_REQ_sum"]
    if flat.get("TCC_EA0_WRREQ_sum"):
        d["ea_write_queue_level_per_request"] = flat.get("TCC_EA0_WRREQ_LEVEL_sum", 0) / flat["TCC_EA0_WRREQ_sum"]
        d["ea_write_dram_credit_stall_per_request"] = flat.get("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", 0) / flat["TCC_EA0_WRREQ_sum"]
    if flat.get("TCC_EA0_RDREQ_sum"):
        d["ea_read_queue_level_per_request"] = flat.get("TCC_EA0_RDREQ_LEVEL_sum", 0) / flat["TCC_EA0_RDREQ_sum"]
        d["ea_read_dram_credit_stall_per_request"] = flat.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", 0) / flat["TCC_EA0_RDREQ_sum"]
    if flat.get("TCC_CYCLE_sum"):
        d["tcc_busy_fraction"] = flat.get("TCC_BUSY_sum", 0) / flat["TCC_CYCLE_sum"]
    rec["derived"] = {k: round(v, 4) for k, v in d.items()}
    rec["counters"] = flat
    out = a.out or os.path.join(ROOT, "gpurun_out", f"box_pmc_{a.label or int(time.time())}.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, "w"))
    print(json.dumps({k: rec[k] for k in ("label", "box", "kernel_ms_mean_over_passes", "derived")}, indent=1))
    print(json.dumps(flat))


if __name__ == "__main__":
    main()
