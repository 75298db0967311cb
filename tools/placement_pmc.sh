#!/bin/bash
# EXPERIMENT: per-dispatch counters of tools/placement_pmc.py, one rocprofv3 --pmc pass per group.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ppmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 python3 $R/tools/placement_pmc.py > $OUT/bare.log 2>&1
i=0
for grp in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum" \
           "TCC_NC_REQ_sum TCC_UC_REQ_sum TCC_CC_REQ_sum TCC_RW_REQ_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_UTCL2_BUSY" \
           "TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -- python3 $R/tools/placement_pmc.py > $OUT/g$i.log 2>&1
done
du -sh $OUT
