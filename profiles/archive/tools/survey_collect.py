#!/usr/bin/env python3
"""gpurun_out/box_survey_*.json (one per gpurun call, tools/box_survey.py) -> profiles/r04_box_survey.jsonl: one line per box,
without the raw tool dumps.  Records of the two exploratory calls that only kept their printed summary (survey1.log: telemetry
still read from the wrong card of the node; survey2.log) are added from those logs, marked as such."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles", "r04_box_survey.jsonl")
recs = []
for path, label, note in ((os.path.join(ROOT, "gpurun_out", "survey1.log"), "box1", "first call: sysfs telemetry read from card0 of the node, which was NOT the leased GPU (amd-smi: BDF 0000:0d:00.0, "
                           "serial da4a9087d5da397c); timings are of the leased GPU"),
                          (os.path.join(ROOT, "gpurun_out", "survey2.log"), "box2", "second call: printed summary only")):
    if os.path.exists(path):
        txt = open(path).read()
        i = txt.find("{")
        try:
            r = json.loads(txt[i:])
            r["label"], r["note"] = label, note
            if label == "box1":
                r["box"] = {"unique_id": "da4a9087d5da397c", "pci": "0000:0d:00.0", "telemetry": "not of this GPU"}
                for leg in r["legs"].values():
                    leg.pop("sampler", None)
                for c in r.get("driver_cadence", {}).values():
                    c.pop("clocks_busy", None), c.pop("clocks_after_idle", None)
            recs.append(r)
        except Exception as exc:
            print("skipping", path, exc, file=sys.stderr)
def order(p):
    m = re.search(r"box(\d+)", p)
    return int(m.group(1)) if m else 0
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "box_survey_*.json")), key=order):
    for line in open(path):
        r = json.loads(line)
        snap = r.pop("snapshot_full", {})
        r["static"] = {"sysfs": snap.get("sysfs"), "hwmon": snap.get("hwmon"), "ras": snap.get("ras"), "kfd": snap.get("kfd"),
                       "node_gpus": snap.get("node_gpus"), "host": snap.get("host"), "sclk_levels": snap.get("sclk_levels"),
                       "mclk_levels": snap.get("mclk_levels"), "fclk_levels": snap.get("fclk_levels")}
        recs.append(r)
with open(out, "w") as f:
    for r in recs:
        f.write(json.dumps(r) + "\n")
print(f"{len(recs)} records -> {out}")
