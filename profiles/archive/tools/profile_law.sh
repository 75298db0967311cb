#!/bin/bash
# rocprofv3 evidence for one law's kernel through tools/bench_laws.py (run via gpurun from the repo root).
# usage: bash tools/profile_law.sh fefp [points]
set -u
LAW=${1:-fefp}
PTS=${2:-10000000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$LAW
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--laws $LAW --points $PTS --reps 8 ${DXM_PROFILE_EXTRA:-}"   # DXM_PROFILE_EXTRA=--tangent-search: with the placement searches of the bench
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_laws.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/bench_laws.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/tools/bench_laws.py $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/tools/bench_laws.py $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_ea -- python3 $R/tools/bench_laws.py $ARGS > $OUT/pmc_ea.log 2>&1
tail -n 3 $OUT/*.log
cd /tmp
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc -- python3 $R/tools/bench_laws.py $ARGS > $OUT/pmc_tcc.log 2>&1
