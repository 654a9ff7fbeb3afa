#!/usr/bin/env python3
"""A/B: zero-guess sweep of the V-cycle written by the cg_update kernel vs its own launch (216^3 AMG-PCG, one process)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
rp, ci, v = problems.poisson3d(n)
N = len(rp) - 1
A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0))
bd, xd = A.dev_alloc(8 * N), A.dev_alloc(8 * N)
A.h2d(bd, np.ones(N))
for rnd in range(4):
    for fz in (0, 1, 2):  # 2: fused, with x / p / Ap / d streamed non-temporally
        A.set_fused_zero_sweep(fz)
        best = 0.0
        for rep in range(2):
            A.h2d(xd, np.zeros(N))
            h, it, sec, rc = A.solve_dev("pcg", bd, xd)
            best = max(best, it / sec)
        print(f"r{rnd} fused_zero={fz}: pcg {it} it, {best:.1f} it/s", flush=True)
