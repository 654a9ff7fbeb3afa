#!/bin/bash
# FETCH_SIZE + basic SQ counters of the finest-level Jacobi sweep under a given kernel config
# (run ON the GPU box):  bash tools/pmc_fetch_cfg.sh "3,0,1,16" tag
R=$(cd "$(dirname "$0")/.." && pwd)
CFG=$1; TAG=$2
export TMPDIR=/tmp SPARSH_PMC_CFG=$CFG
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/diag/$TAG/g1 -- python3 $R/tools/pmc_traffic.py --run > $R/gpurun_out/diag_${TAG}_g1.log 2>&1 || echo "fetch pass failed"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/diag/$TAG/g2 -- python3 $R/tools/pmc_traffic.py --run > $R/gpurun_out/diag_${TAG}_g2.log 2>&1 || echo "sq pass failed"
