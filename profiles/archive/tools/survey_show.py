#!/usr/bin/env python3
"""Condensed view of box-survey records (tools/box_survey.py): one block per box."""
import json
import sys

for path in sys.argv[1:] or ["gpurun_out/box_survey.jsonl"]:
    for line in open(path):
        r = json.loads(line)
        b = r["box"]
        print(f"== {r.get('label')} {r.get('time')} uid {b.get('unique_id')} pci {b.get('pci')} (hip {r.get('hip_pci_bus_id')}) {b.get('compute_partition')}/{b.get('memory_partition')} "
              f"cap {b.get('power_cap_w')} W vram_used {b.get('vram_used')} others_on_gpu {b.get('vram_of_kfd_processes_on_my_gpu', b.get('vram_other_processes_on_my_gpu'))} "
              f"node busy GPUs besides mine {b.get('node_gpus_busy_besides_mine')} load {b.get('host_loadavg')}")
        for k, v in r["legs"].items():
            s = v["sampler"]
            print(f"  {k:18s} n={v['launches']:5d} first5 {v['first5_ms']} steady {v['steady_ms']:.4f} min {v['min_ms']:.4f} {v['GBs']:7.1f} GB/s "
                  f"clk {v.get('shader_clock_under_load_mhz')} sclk {s.get('sclk')} mclk {s.get('mclk')} fclk {s.get('fclk')} P {s.get('power_w')} "
                  f"busy {s.get('gpu_busy')} mem_busy {s.get('mem_busy')}")
            f = v.get("firmware") or {}
            if f:
                print("      firmware: " + " ".join(f"{k}={f[k]}" for k in f if f[k] is not None and k != "firmware_samples"))
        print(f"  j2 frac {r['j2_steady_frac_of_peak']}  j2/stream_nt {r['j2_over_stream_nt']}  j2/shape17 {r['j2_over_stream_j2_shape']}")
        for k, v in r["driver_cadence"].items():
            print(f"  {k:12s} warm5 {v['warmup5_ms']} timed20 mean {v['timed20_mean_ms']} max {v['timed20_max_ms']} after-idle {v['clocks_after_idle']}")
