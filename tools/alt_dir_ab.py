#!/usr/bin/env python3
"""A/B: consecutive sweeps walking the level in alternating directions vs always ascending (216^3: back-to-back sweeps
and the whole AMG-PCG iteration; one process, interleaved rounds)."""
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
for rnd in range(3):
    for alt in (0, 1):
        A.set_alternate_sweeps(bool(alt))
        t = [A.bench_op("jacobi_pingpong", l, 40) * 1e6 for l in (0, 1, 2)]
        A.h2d(xd, np.zeros(N))
        h, it, sec, rc = A.solve_dev("pcg", bd, xd)
        print(f"r{rnd} alternate={alt}: ping-pong sweeps L0 {t[0]:.1f} L1 {t[1]:.1f} L2 {t[2]:.1f} us; pcg {it} it {it / sec:.1f} it/s", flush=True)
