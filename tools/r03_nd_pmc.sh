#!/bin/bash
# HBM traffic of the nested-dissection coarse solve (FETCH_SIZE / WRITE_SIZE passes) + kernel trace, 100^3 / 2D 1000^2 / FEM
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_nd
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in 100 2d fem; do
  rm -rf /tmp/ndpmc_f /tmp/ndpmc_w /tmp/ndtr
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/ndpmc_f -- python3 $R/tools/nd_trace.py --run --case $c > $O/pmc_f_$c.log 2>&1 || { echo "fetch pass $c failed"; continue; }
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/ndpmc_w -- python3 $R/tools/nd_trace.py --run --case $c > $O/pmc_w_$c.log 2>&1 || { echo "write pass $c failed"; continue; }
  python3 $R/tools/nd_trace.py --summarize-pmc /tmp/ndpmc_f /tmp/ndpmc_w > $O/nd_solve_pmc_$c.json; cat $O/nd_solve_pmc_$c.json
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ndtr -- python3 $R/tools/nd_trace.py --run --case $c > $O/trace_$c.log 2>&1 && python3 $R/tools/nd_trace.py --summarize /tmp/ndtr > $O/nd_solve_trace_$c.txt && cat $O/nd_solve_trace_$c.txt
done
