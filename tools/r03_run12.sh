#!/bin/bash
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_final
mkdir -p $O
cd $R
timeout -k 10 1700 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_parity.py -m gpu -q -k "not full_size_properties and not extended_hierarchy" > $O/gpu_tests_rest.log 2>&1; echo "rest tests rc=$?"; tail -12 $O/gpu_tests_rest.log
# can two ranks share one GPU under RCCL here?
timeout -k 10 180 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/micro/rccl_same_gpu_probe.py > $O/rccl_same_gpu_probe.log 2>&1; echo "rccl same-gpu probe rc=$?"; tail -8 $O/rccl_same_gpu_probe.log
