#!/bin/bash
# After `bash tools/profile_all.sh` on a GPU box: the four per-kernel summaries of ONE lease + that lease's box block into profiles/.
#     bash tools/summarize_all.sh r05        ->  profiles/r05_{j2linear,elastic,j2voce,fefp}_{summary.md,pmc.json,kernel_stats.csv}, profiles/r05_profile_box.json
set -eu
TAG=${1:?round tag, e.g. r05}
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/gpurun_out/prof_all
LAWCMD="\`bash tools/profile_all.sh\` (one lease for all four kernels: rocprofv3 --kernel-trace --stats, then separate --pmc passes) around \`python3 tools/bench_laws.py --laws LAW --points 10000000 --reps 60 --warmup 20\` (device-resident arrays where their first allocation put them; 1 + 20 untimed launches, then 60 timed ones)."
python3 $R/tools/summarize_profile.py $SRC/j2_linear $R/profiles/${TAG}_j2linear --kernel "small_strain_kernel<1" --law j2_linear --alg-bytes 496 --steps 200 \
  --command-text "\`bash tools/profile_all.sh\` (one lease for all four kernels: rocprofv3 --kernel-trace --stats, then separate --pmc passes) around \`python3 bench.py --no-cpu-baseline --no-other-laws --no-host-path --no-live-traffic --no-stream-probe --no-telemetry\` (the default bench command: 200 steps, 10 warm-up, every array where its first allocation put it, without the context legs)." > /dev/null
python3 $R/tools/summarize_profile.py $SRC/elastic $R/profiles/${TAG}_elastic --kernel "small_strain_kernel<0" --law elastic --alg-bytes 384 --steps 60 --no-traffic-json --command-text "${LAWCMD//LAW/elastic}" > /dev/null
python3 $R/tools/summarize_profile.py $SRC/j2_voce $R/profiles/${TAG}_j2voce --kernel "small_strain_kernel<2" --law j2_voce --alg-bytes 496 --steps 60 --no-traffic-json --command-text "${LAWCMD//LAW/j2_voce}" > /dev/null
python3 $R/tools/summarize_profile.py $SRC/fefp $R/profiles/${TAG}_fefp --kernel "fefp_kernel<1" --law fefp --alg-bytes 976 --steps 60 --no-traffic-json --command-text "${LAWCMD//LAW/fefp}" > /dev/null
python3 - "$SRC" "$R/profiles/${TAG}_profile_box.json" <<'PY'
import json, sys
src, dst = sys.argv[1:3]
box = json.load(open(src + "/box.json"))
line = None
for ln in open(src + "/bench.json"):
    if ln.startswith("{"):
        line = json.loads(ln)
keep = {"box_snapshot_before_the_profiles": box}
if line:
    r = line["roofline"]
    keep["unprofiled_bench_line_of_this_lease"] = {"value": line["value"], "ms_per_step": line["ms_per_step"], "kernel_ms": r["kernel_ms"], "frac": r["frac"],
                                                   "frac_of_stream_probe": r.get("frac_of_stream_probe"), "traffic_over_algorithmic": r.get("traffic_over_algorithmic"),
                                                   "other_laws": line.get("other_laws"), "box": line.get("box")}
json.dump(keep, open(dst, "w"), indent=1)
PY
grep -h "of peak; this is\|of the 8 TB/s" $R/profiles/${TAG}_*_summary.md
