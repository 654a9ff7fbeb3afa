#!/bin/bash
# A/B of the coarse-factor prefetch (same box, interleaved), then PMC traffic of the ND solve
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_nd
mkdir -p $O
cd $R
for rep in 1 2; do
for pf in 0 1; do
  SPARSH_COARSE_PREFETCH=$pf python tools/config_bench.py C3D_poisson3d_100 C2D_poisson2d_1000 CU_fem_unstructured_525825 > $O/pf_cfg_${pf}_$rep.json 2>/dev/null
  SPARSH_COARSE_PREFETCH=$pf python bench.py --no-cpu --no-pmc --no-families > $O/pf_bench_${pf}_$rep.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("$O/pf_cfg_${pf}_$rep.json"))
b=json.loads(open("$O/pf_bench_${pf}_$rep.json").read().strip().splitlines()[-1])
print("prefetch=$pf rep=$rep | 216^3 %.1f it/s |"%b["value"], " | ".join("%s %s"%(k.split("_")[1], " ".join("%s %.0f"%(m, v[m]["rate"]) for m in ("amg","pcg","pbicg") if m in v)) for k,v in d.items()))
PY
done
done
bash tools/r03_nd_pmc.sh
