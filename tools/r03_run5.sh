#!/bin/bash
# top-merge sweep of the nested-dissection solver, per-launch traces, then the whole GPU suite and the default bench
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_nd
mkdir -p $O
cd $R
for f in nd,64,384,0 nd,64,384,1024 nd,64,384,2048 nd,64,384,4096 nd,64,192,2048; do
  SPARSH_COARSE_FORM=$f timeout -k 10 300 python tools/config_bench.py C3D_poisson3d_100 C2D_poisson2d_1000 CU_fem_unstructured_525825 > $O/configs_$f.json 2> $O/configs_$f.err; echo "config bench $f rc=$?"
  python - <<PY
import json
d=json.load(open("$O/configs_$f.json"))
for k,v in d.items():
    c=v["coarsest"]
    if c["form"]!="nested_dissection": continue
    print("$f", k, "coarse_us", v["coarse_solve_us"], "setup_s", v["setup_seconds"], "MB %.1f"%(c["bytes"]/1e6), "levels", c["nd_levels"], "nodes", c["nd_nodes"], "maxpiv", c["nd_max_pivot"], " | ".join("%s %s"%(m, v[m]["rate"]) for m in ("amg","pcg","pbicg") if m in v and "rate" in v[m]))
PY
done
export TMPDIR=/tmp
cd /tmp
for c in 100 2d fem; do
  rm -rf /tmp/ndtr_$c
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ndtr_$c -- python3 $R/tools/nd_trace.py --run --case $c > $O/trace_$c.log 2>&1 || { echo "trace $c failed"; tail -5 $O/trace_$c.log; continue; }
  python3 $R/tools/nd_trace.py --summarize /tmp/ndtr_$c > $O/nd_solve_trace_$c.txt || head -3 $(find /tmp/ndtr_$c -name "*kernel_trace.csv" | head -1)
  cat $O/nd_solve_trace_$c.txt
done
cd $R
timeout -k 10 1500 python -m pytest tests -m gpu -x -q ${PYTEST_EXTRA} > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/gpu_tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"])[:600])
print(json.dumps(d["config"]["whole_iteration"]))
print(json.dumps(d["config"]["spmv_finest_level"]))
print(json.dumps(d["cpu_baseline"])[:700])
PY
