#!/bin/bash
# tools/pmc_loop.sh FIRST LAST: one gpurun lease per label pmc<k>: behind-the-L2 counters of the headline kernel, then a short survey
cd "$(dirname "$0")/.." || exit 1
for k in $(seq "$1" "$2"); do
  timeout 1500 tools/gpu.sh --timeout 600 -- "python tools/box_pmc.py --label pmc$k > gpurun_out/box_pmc_$k.log 2>&1; tail -c 1500 gpurun_out/box_pmc_$k.log" > /tmp/pmc_call_$k.txt 2>&1
  echo "pmc$k rc $?"
done
