#!/bin/bash
# ONE script, ONE lease: the rocprofv3 evidence of all four kernels (J2 linear through bench.py, the headline command; elastic,
# J2 Voce and FeFp through tools/bench_laws.py) plus the lease's own box block, so that DESIGN.md section 3's table can quote
# four summaries that share a box.  Run through gpurun from the repo root:
#     gpurun --timeout 1500 -- 'bash tools/profile_all.sh'
# then here:  bash tools/summarize_all.sh r05
# Counters are collected in their own passes (never together with sys/hip/hsa tracing); the program itself follows `--`.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TOP=$R/gpurun_out/prof_all
rm -rf $TOP; mkdir -p $TOP
cd /tmp && export TMPDIR=/tmp
# the box, from its own side, before any profiler is around (sysfs + SMI tools; tools/box_telemetry.py refuses under a profiler)
python3 $R/tools/box_telemetry.py > $TOP/box.json 2> $TOP/box.err
# an unprofiled bench line of the same lease (the figures the summaries are compared with)
python3 $R/bench.py --no-cpu-baseline --no-host-path > $TOP/bench.json 2> $TOP/bench.err

OUT=$TOP/j2_linear; mkdir -p $OUT
ARGS="--no-cpu-baseline --no-other-laws --no-host-path --no-live-traffic --no-stream-probe --no-telemetry"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/pmc_ea -- python3 $R/bench.py $ARGS > $OUT/pmc_ea.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > $OUT/pmc_sq.log 2>&1

for LAW in elastic j2_voce fefp; do
  OUT=$TOP/$LAW; mkdir -p $OUT
  ARGS="--laws $LAW --points 10000000 --reps 60 --warmup 20"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_laws.py $ARGS > $OUT/trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/bench_laws.py $ARGS > $OUT/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/tools/bench_laws.py $ARGS > $OUT/pmc_write.log 2>&1
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 $R/tools/bench_laws.py $ARGS > $OUT/pmc_tcc.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/bench_laws.py $ARGS > $OUT/pmc_sq.log 2>&1
done
# keep what travels back small: the per-dispatch traces are needed (timed region), the rest of rocprofv3's output is not
find $TOP -name "*agent_info.csv" -delete
du -sh $TOP; tail -n 2 $TOP/*/trace.log
