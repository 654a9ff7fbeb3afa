#!/bin/bash
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_nd
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_coarse.py -x -q > $O/tests_coarse.log 2>&1; echo "coarse tests rc=$?"; tail -4 $O/tests_coarse.log
export TMPDIR=/tmp
cd /tmp
for c in 100 2d fem; do
  rm -rf /tmp/ndtr_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ndtr_$c -- python3 $R/tools/nd_trace.py --run --case $c > $O/trace_$c.log 2>&1 || { echo "trace $c failed"; tail -5 $O/trace_$c.log; continue; }
  python3 $R/tools/nd_trace.py --summarize /tmp/ndtr_$c > $O/nd_solve_trace_$c.txt
  cat $O/nd_solve_trace_$c.txt
done
cd $R
# 216^3: where the extended hierarchy stops (round 2: <= 4000 rows, 13 levels; now: <= coarse_limit, nested dissection on 39 366 rows)
for u in 4000 0; do
  SPARSH_EXTEND_UNTIL=$u python bench.py --no-cpu --no-pmc --no-families > $O/bench216_extend_$u.json 2> $O/bench216_extend_$u.err; echo "bench extend_until=$u rc=$?"
  python - <<PY
import json
d=json.loads(open("$O/bench216_extend_$u.json").read().strip().splitlines()[-1])
c=d["config"]
print("extend_until=$u", d["value"], "it/s", d["ms_per_step"], "ms; levels", len(c["levels"]), c["levels"][-3:], "coarsest", c["coarsest_level"]["form"], "full solve", c["full_solve_to_1e-8"], "setup", c["setup_seconds_host"])
PY
done
