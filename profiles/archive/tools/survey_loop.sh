#!/bin/bash
# tools/survey_loop.sh FIRST LAST: one gpurun call (= one fresh box) per label box<k>, sequentially
cd "$(dirname "$0")/.." || exit 1
for k in $(seq "$1" "$2"); do
  timeout 1500 tools/gpu.sh --timeout 420 -- "python tools/box_survey.py --label box$k > gpurun_out/survey$k.log 2>&1; tail -c 200 gpurun_out/survey$k.log" > /tmp/survey_call_$k.txt 2>&1
  echo "box$k rc $?"
done
