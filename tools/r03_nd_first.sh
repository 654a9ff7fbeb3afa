#!/bin/bash
# first GPU run of the nested-dissection coarse solver: its tests, then config bench nd vs bt, then the profiler launch-count micro test
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_nd
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_coarse.py -x -q > $O/tests_coarse.log 2>&1; echo "coarse tests rc=$?"; tail -15 $O/tests_coarse.log
for f in nd bt; do
  SPARSH_COARSE_FORM=$f timeout -k 10 300 python tools/config_bench.py C3D_poisson3d_100 C2D_poisson2d_1000 CU_fem_unstructured_525825 > $O/configs_$f.json 2> $O/configs_$f.err; echo "config bench $f rc=$?"
  grep -E "^\[" $O/configs_$f.err
done
python - <<PY
import json
for f in ("nd","bt"):
    try:
        d=json.load(open("$O/configs_%s.json"%f))
        for k,v in d.items():
            c=v["coarsest"]; print(f, k, "coarse_solve_us", v["coarse_solve_us"], "setup_s", v["setup_seconds"], c.get("form"), "MB", c["bytes"]/1e6, "nd_levels", c.get("nd_levels"), "nodes", c.get("nd_nodes"), "maxpiv", c.get("nd_max_pivot"))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp; export TMPDIR=/tmp
for n in 10000 40000; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/many_$n -- $R/tools/micro/many_launches $n 8192 > $O/many_$n.log 2>&1
  echo "rocprofv3 --pmc with $n trivial launches: exit code $?" | tee -a $O/many_launches_exit.txt
  tail -3 $O/many_$n.log
  rm -rf $O/many_$n
done
