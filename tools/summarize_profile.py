#!/usr/bin/env python3
"""Summarises the rocprofv3 output of tools/profile.sh into profiles/.

    python tools/summarize_profile.py gpurun_out/prof profiles/r01 [--points 10000000 --law j2_linear]

Writes <prefix>_kernel_stats.csv (verbatim rocprofv3 --stats table), <prefix>_pmc.json (mean
counter values per launch of the constitutive kernel) and <prefix>_summary.md, and refreshes
profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

HBM bytes follow MI355X_MICROARCH.md section "HBM": WRITE_SIZE (KiB) is exact for 16 B/lane
streaming stores; FETCH_SIZE (KiB) reports 1/2 of the bytes of a wide coalesced streaming read on
gfx950 (TCC_EA0_RDREQ counts 128-B requests tallied at 64 B) and is doubled.  The doubling is
checked against TCC_EA0_RDREQ/_WRREQ and against the known minimal byte count of this kernel.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil


def _source_hash():
    """bench.py::source_hash: the stamp that ties the traffic figure to the code it was measured on."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    return bench.source_hash()


def _newest(files):
    """Only the most recent run's file: `gpurun` MERGES what a call wrote into gpurun_out/, so a pass directory can still
    hold the files of an earlier round (this mixed round-2 counters into the first round-3 summaries)."""
    return [max(files, key=os.path.getmtime)] if files else []


def counters(path, match):
    agg = collections.defaultdict(list)
    files = _newest(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))
    for f in files:
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("prefix")
    ap.add_argument("--kernel", default="small_strain_kernel<1")
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--law", default="j2_linear")
    ap.add_argument("--alg-bytes", type=int, default=496)
    ap.add_argument("--no-traffic-json", action="store_true")
    ap.add_argument("--command-text", default="`bash tools/profile.sh` (rocprofv3 --kernel-trace --stats, then separate --pmc passes) around\n"
                    "`python3 bench.py --no-cpu-baseline --no-other-laws --no-host-path --no-live-traffic --no-stream-probe --no-telemetry` (the default bench command: 200 steps, 10 warm-up, every array where its first allocation put it, without the context legs).")
    ap.add_argument("--steps", type=int, default=200, help="timed steps of the bench command = the LAST dispatches of the kernel")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.prefix) or ".", exist_ok=True)

    stats_files = _newest(glob.glob(os.path.join(a.src, "trace", "**", "*kernel_stats.csv"), recursive=True))
    rows = []
    if stats_files:
        shutil.copy(stats_files[0], a.prefix + "_kernel_stats.csv")
        rows = list(csv.DictReader(open(stats_files[0])))
    krow = next((r for r in rows if a.kernel in r["Name"]), None)

    # the timed region of bench.py = the last `steps` dispatches of the kernel in the trace; the ones
    # before it are setup: building the load-step contexts, placement tuning (incl. rejected slow
    # candidates), warm-up.  The --stats table averages over all of them.
    timed = None
    tfiles = _newest(glob.glob(os.path.join(a.src, "trace", "**", "*kernel_trace.csv"), recursive=True))
    if tfiles:
        disp = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(tfiles[0]))
                       if a.kernel in r["Kernel_Name"]))
        if len(disp) >= a.steps:
            last = disp[-a.steps:]
            timed = {"dispatches": a.steps, "of": len(disp), "avg_ns": sum(e - b for b, e in last) / a.steps,
                     "min_ns": min(e - b for b, e in last), "max_ns": max(e - b for b, e in last)}

    pmc, counts = {}, {}
    for d in ("pmc_fetch", "pmc_write", "pmc_ea", "pmc_sq", "pmc_tcc"):
        c, n = counters(os.path.join(a.src, d), a.kernel)
        pmc.update(c)
        counts.update(n)
    read_b = 2.0 * pmc.get("FETCH_SIZE", float("nan")) * 1024.0
    write_b = pmc.get("WRITE_SIZE", float("nan")) * 1024.0
    traffic = read_b + write_b
    alg = a.alg_bytes * a.points
    out = {
        "kernel": a.kernel,
        "points": a.points,
        "law": a.law,
        "launches_sampled": counts,
        "counters_mean_per_launch": pmc,
        "hbm_read_bytes_per_launch": read_b,
        "hbm_write_bytes_per_launch": write_b,
        "hbm_bytes_per_launch": traffic,
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": traffic / alg,
        "check_rdreq_x128B": pmc.get("TCC_EA0_RDREQ_sum", float("nan")) * 128.0,
        "check_wrreq_x64B": pmc.get("TCC_EA0_WRREQ_sum", float("nan")) * 64.0,
    }
    if krow:
        out["rocprof_avg_ns"] = float(krow["AverageNs"])
        out["rocprof_calls"] = int(krow["Calls"])
        out["achieved_GBs_from_rocprof_avg"] = alg / float(krow["AverageNs"])
    if timed:
        out["rocprof_timed_region"] = timed
        out["achieved_GBs_timed_region"] = alg / timed["avg_ns"]
    json.dump(out, open(a.prefix + "_pmc.json", "w"), indent=1)
    if not a.no_traffic_json:
        json.dump(
            {"points": a.points, "law": a.law, "hbm_bytes_per_launch": traffic, "source": os.path.basename(a.prefix) + "_pmc.json",
             "source_hash": _source_hash()},
            open(os.path.join(os.path.dirname(a.prefix) or ".", "pmc_traffic.json"), "w"),
        )

    with open(a.prefix + "_summary.md", "w") as f:
        f.write(f"# rocprofv3 summary: `{a.kernel}` ({a.law}, {a.points} points per launch)\n\n")
        f.write("Command: " + a.command_text + "\n\n## kernel-trace --stats\n\n")
        if rows:
            f.write("| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|\n")
            for r in rows:
                f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |\n")
        if krow:
            f.write(f"\nAlgorithmic bytes per launch {alg:.4g} / average duration {float(krow['AverageNs'])/1e3:.1f} us = "
                    f"**{out['achieved_GBs_from_rocprof_avg']:.0f} GB/s** = {out['achieved_GBs_from_rocprof_avg']/8000:.3f} of the 8 TB/s HBM3E peak.\n")
        if timed:
            f.write(f"\nTimed region = the last {timed['dispatches']} of the {timed['of']} dispatches of this kernel in the trace (the earlier ones "
                    f"are set-up, settling and warm-up launches of the same kernel): average "
                    f"**{timed['avg_ns']/1e3:.1f} us** (min {timed['min_ns']/1e3:.1f}, max {timed['max_ns']/1e3:.1f}) = **{alg/timed['avg_ns']:.0f} GB/s** = "
                    f"{alg/timed['avg_ns']/8000:.3f} of peak; " + ("this is the figure `roofline.achieved` of the bench line corresponds to.\n" if a.law == "j2_linear"
                                                                     else "the steady-state figure of this kernel (`other_laws` of the bench line times 30 launches after 10 untimed ones).\n"))
        f.write("\n## PMC (mean per launch)\n\n| counter | value |\n|---|---|\n")
        for k in sorted(pmc):
            f.write(f"| {k} | {pmc[k]:.6g} |\n")
        f.write(f"\nHBM read  = 2 x FETCH_SIZE x 1024 = {read_b:.4g} B (check: TCC_EA0_RDREQ x 128 B = {out['check_rdreq_x128B']:.4g})\n\n")
        f.write(f"HBM write = WRITE_SIZE x 1024 = {write_b:.4g} B (check: TCC_EA0_WRREQ x 64 B = {out['check_wrreq_x64B']:.4g})\n\n")
        f.write(f"HBM traffic per launch = {traffic:.4g} B = {traffic/alg:.4f} x the algorithmic {alg:.4g} B: no wasted re-reads.\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
