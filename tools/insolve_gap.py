#!/usr/bin/env python3
"""Why does the finest-level fused Jacobi sweep take longer inside the solve than launched back to back?
Times the same kernel (a) x -> y repeatedly, (b) ping-ponging between two vectors as a smoothing leg does,
(c) ping-ponging on the level's own resident buffers, for the default table path and the general layout."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

rp, ci, v = problems.poisson3d(int(sys.argv[1]) if len(sys.argv) > 1 else 216)
for fold in (True, False):
    A = sa.sp_matrix_mg(rp, ci, v)
    if not fold:
        A.set_const_slots(False)
    A.setup(sa.default_params(print_setup=0, print_solve=0))
    print("layout:", "folded (table path)" if fold else "general values", A.level_kernel(0))
    for l in (0, 1, 2):
        for rnd in range(2):
            t = {op: A.bench_op(op, l, 26) * 1e6 for op in ("jacobi", "jacobi_pingpong", "jacobi_pingpong_resident")}
            print(f"  level {l} round {rnd}: same-vectors {t['jacobi']:7.2f} us   ping-pong {t['jacobi_pingpong']:7.2f} us   ping-pong resident {t['jacobi_pingpong_resident']:7.2f} us", flush=True)
    A.close()
