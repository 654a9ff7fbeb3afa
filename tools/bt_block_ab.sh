#!/bin/bash
# A/B of the block size of the block-tridiagonal coarse solver (wide-band coarsest levels: 100^3, FEM stand-in)
for blk in 0 1280 1536 2048; do
  echo "== SPARSH_COARSE_BLOCK=$blk"
  SPARSH_COARSE_BLOCK=$blk python tools/config_bench.py C3D_poisson3d_100 CU_fem_unstructured_525825 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items():
    if 'extended' in k or 'beck' in k: continue
    print(k, v['coarsest']['block'], v['coarsest']['nblocks'], v['coarsest']['bytes']>>20, 'MB', v['coarse_solve_us'], 'us', {m:v[m]['rate'] for m in ('amg','pcg','pbicg') if m in v}, 'setup', v['setup_seconds'])"
done
