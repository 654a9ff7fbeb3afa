#!/usr/bin/env python3
"""A/B of the 16-bit column index form on Beck's multi-entry P / R at 216^3 (the operators the default policy compresses)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
rp, ci, v = problems.poisson3d(n)
N = len(rp) - 1
for mode in (0, 1, 0, 1):
    A = sa.sp_matrix_mg(rp, ci, v).set_index_compression(mode).setup(sa.default_params(print_setup=0, print_solve=0, coarsening=1))
    out = [f"idx16 mode {mode}:"]
    for l in (0, 1):
        i = A.level_info(l)
        for op in ("restrict", "prolong", "jacobi"):
            out.append(f"L{l} {op} {A.bench_op(op, l, 20) * 1e6:.1f}us")
    bd, xd = A.dev_alloc(8 * N), A.dev_alloc(8 * N)
    A.h2d(bd, np.ones(N))
    best = 0
    for rep in range(2):
        A.h2d(xd, np.zeros(N))
        h, it, sec, rc = A.solve_dev("pcg", bd, xd)
        best = max(best, it / sec)
    out.append(f"pcg {best:.1f} it/s")
    print(" ".join(out), flush=True)
    A.close()
