#!/bin/bash
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_final
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_coarse.py -m gpu -q -k "level_policy or config1_full or config2_full or coarse" 2>&1 | tail -5
timeout -k 10 600 python tools/config_bench.py C3D_poisson3d_136 BIG_poisson2d > $O/configs_policy.json; echo "config rc=$?"
python - <<PY
import json
d=json.load(open("$O/configs_policy.json"))
for k,v in d.items():
    c=v["coarsest"]; print(k, len(v["levels"]), v["levels"][-1], "ext", c["extended"], "coarse_us", v["coarse_solve_us"], "setup", v["setup_seconds"], c["form"], "MB %.0f"%(c["bytes"]/1e6), {m:(v[m].get("count"),v[m].get("seconds"),v[m].get("rate")) for m in ("amg","pcg") if m in v})
PY
