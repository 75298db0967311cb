#!/usr/bin/env python3
"""profiles/r04_box_survey.jsonl -> profiles/r04_box_survey.md: one row per lease, the figures the conclusion rests on."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r04_box_survey.jsonl"))]


def g(d, *keys, default=None):
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


out = ["# r04 — box survey (tools/box_survey.py, one gpurun lease per row)", "",
       "Kernel and probe columns: steady-state ms per launch at 1e7 points / 4.96 GB (median of the second half of ~1 s of back-to-back launches).",
       "`clk`: in-kernel shader clock under the streaming load (s_memtime / s_memrealtime, median over workgroups).  `P`: mean socket power over the J2 leg",
       "(firmware energy accumulator).  `ppt`: fraction of firmware samples with the power limiter active.  Telemetry of box1 was read from the wrong card (see note in the record).", "",
       "| lease | GPU unique_id | VBIOS | J2 kernel | frac of 8 TB/s | linear probe | 17-stream probe | read only | write only (nt) | torch copy GB/s | clk MHz | sclk / uclk / fclk | P W (cap) | ppt | HBM °C | thermal residency | partition | other VRAM users on this GPU |",
       "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    L, b = r["legs"], r.get("box", {})
    fw = g(L, "j2_kernel", "firmware", default={}) or {}
    smp = g(L, "j2_kernel", "sampler", default={}) or {}
    med = lambda k: (smp.get(k) or [None, None, None])[1]   # noqa: E731
    therm = [fw.get(k) for k in ("socket_thm_residency_frac", "hbm_thm_residency_frac", "prochot_residency_frac")]
    out.append("| {} | `{}` | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} / {} / {} | {} ({}) | {} | {} | {} | {} / {} | {} |".format(
        r.get("label"), (b.get("unique_id") or "?")[:8] + "…", (b.get("vbios") or "–")[-4:], g(L, "j2_kernel", "steady_ms"), r.get("j2_steady_frac_of_peak"), g(L, "stream_nt", "steady_ms"),
        g(L, "stream_j2_shape", "steady_ms"), g(L, "read_only", "steady_ms", default="–"), g(L, "write_only_nt", "steady_ms", default="–"),
        g(L, "torch_copy_1GiB", "GBs"), (g(L, "stream_stamped", "shader_clock_under_load_mhz") or [None, "–", None])[1],
        med("sclk") or "–", fw.get("current_uclk_end") or med("mclk") or "–", med("fclk") or "–", fw.get("mean_socket_power_w", "–"), b.get("power_cap_w", "–"),
        fw.get("ppt_residency_frac", "–"), fw.get("temperature_mem_end", "–"), "0" if therm and all(t == 0.0 for t in therm) else (therm if any(t is not None for t in therm) else "–"),
        b.get("compute_partition", "–"), b.get("memory_partition", "–"), b.get("vram_of_kfd_processes_on_my_gpu", b.get("vram_other_processes_on_my_gpu", "–"))))
out += ["", "Driver cadence (host idle for t seconds, then 5 + 20 launches, each with its own event pair; mean of the 20, ms):", "",
        "| lease | t = 0 | 0.05 | 0.5 | 2 | 5 | first launch after 5 s idle |", "|---|---|---|---|---|---|---|"]
for r in rows:
    c = r.get("driver_cadence", {})
    out.append("| {} | {} | {} | {} | {} | {} | {} |".format(r.get("label"), *[g(c, f"idle_{t}s", "timed20_mean_ms") for t in ("0.0", "0.05", "0.5", "2.0", "5.0")],
                                                           (g(c, "idle_5.0s", "warmup5_ms") or ["–"])[0]))
open(os.path.join(ROOT, "profiles", "r04_box_survey.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
