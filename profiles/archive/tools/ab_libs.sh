#!/bin/bash
# A/B of alternative builds of libdxmat.so on ONE box (boxes differ by > 10 %): runs
# tools/bench_laws.py with DXM_LIB_PATH pointing at each library in turn, twice, interleaved.
#   tools/ab_libs.sh out.jsonl "laws..." lib1.so lib2.so ...
out=$1; laws=$2; shift 2
: > "$out"
for rep in 1 2; do
  for lib in "$@"; do
    DXM_LIB_PATH=$lib timeout 300 python tools/bench_laws.py --laws $laws --reps 30 2>&1 | while read -r line; do
      echo "{\"lib\": \"$(basename $lib)\", \"rep\": $rep, \"r\": $line}" >> "$out"
    done
  done
done
