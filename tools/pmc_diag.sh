#!/bin/bash
# Diagnostic counter passes on the finest-level fused Jacobi sweep (run ON the GPU box):
#   bash tools/pmc_diag.sh "3,0,1,16" tagA ; bash tools/pmc_diag.sh "3,0,0,1" tagB
# One rocprofv3 pass per counter group (no trace domains besides --kernel-trace), results under
# gpurun_out/diag/<tag>/<group>/ ; tools/pmc_diag_summary.py prints the Jacobi kernel's rows.
R=$(cd "$(dirname "$0")/.." && pwd)
CFG=$1; TAG=$2
export TMPDIR=/tmp SPARSH_PMC_CFG=$CFG
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_DEP_WAIT" \
           "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TA_TA_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/diag/$TAG/g$i -- python3 $R/tools/pmc_traffic.py --run > $R/gpurun_out/diag_${TAG}_g$i.log 2>&1 || echo "group $i ($grp) failed"
done
