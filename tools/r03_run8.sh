#!/bin/bash
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_nd
mkdir -p $O
cd $R
python tools/fem_tail_probe.py > $O/fem_tail_probe.txt 2>&1; cat $O/fem_tail_probe.txt
timeout -k 10 1700 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_coarse.py > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/gpu_tests.log
