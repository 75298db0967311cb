#!/usr/bin/env python3
"""Summarise gpurun_out/ppmc (tools/placement_pmc.sh): counters per handle next to the bare times."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ppmc"
bare = json.loads(open(os.path.join(src, "bare.log")).read().strip().splitlines()[-1])
H = len(bare)
rows = {h["handle"]: {"state": h["state"], "bare_ms": h["median_ms"]} for h in bare}
for d in sorted(glob.glob(os.path.join(src, "g*"))):
    if not os.path.isdir(d):
        continue
    per = collections.defaultdict(dict)   # counter -> dispatch id -> value
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "small_strain" in r["Kernel_Name"]:
                per[r["Counter_Name"]][int(r["Dispatch_Id"])] = per[r["Counter_Name"]].get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    for name, byid in per.items():
        ids = sorted(byid)[-6 * H:]
        for h in range(H):
            vals = [byid[i] for i in ids[6 * h: 6 * h + 6]]
            rows[h][name] = sum(vals) / len(vals)
names = sorted({k for r in rows.values() for k in r if k not in ("state", "bare_ms")})
print("counter".ljust(44) + "".join(f"{rows[h]['state'][:11]:>13}" for h in range(H)))
print("bare_ms".ljust(44) + "".join(f"{rows[h]['bare_ms']:13.4f}" for h in range(H)))
for nme in names:
    print(nme.ljust(44) + "".join(f"{rows[h].get(nme, float('nan')):13.4g}" for h in range(H)))
