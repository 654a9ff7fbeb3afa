#!/bin/bash
# nested-dissection coarse solver: tests, then a sweep of (leaf, merge_rows) on the three reference-policy configs
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_nd
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_coarse.py -x -q > $O/tests_coarse.log 2>&1; echo "coarse tests rc=$?"; tail -4 $O/tests_coarse.log
for f in ${SWEEP:-nd nd,32,192 nd,64,384 nd,64,768 nd,128,384 nd,128,1024 nd,32,96}; do
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
