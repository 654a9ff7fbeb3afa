#!/bin/bash
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/r03_nd
mkdir -p $O
cd $R
for t in 0 2048; do
SPARSH_ND_TIMING=1 python - <<PY
import sparsh_amg_amd as sa, numpy as np, time
from sparsh_amg_amd import problems
for name, gen in (("100^3", lambda: problems.poisson3d(100)), ("2d1000", lambda: problems.poisson2d(1000))):
    rp, ci, v = gen()
    for rep in range(2):
        A = sa.sp_matrix_mg(rp, ci, v).set_coarse_form("nd", 0, -1, $t)
        t0 = time.time(); A.setup(sa.default_params(print_setup=0, print_solve=0)); t1 = time.time()
        print(name, "top", $t, "rep", rep, "setup wall %.3f s, setup_seconds %.3f" % (t1 - t0, A.setup_seconds), flush=True)
        A.close()
PY
done
timeout -k 10 1700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/gpu_tests.log
