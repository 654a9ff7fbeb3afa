#!/usr/bin/env python3
"""A/B: consecutive sweeps walking the level in alternating directions (mode 2) vs always ascending (mode 0), 216^3, for the
table path, the general sliced-diagonal layout and the CSR-stream family; one process, interleaved rounds."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
rp, ci, v = problems.poisson3d(n)
N = len(rp) - 1
for name, fold, cfg in (("table", True, None), ("general_sdia", False, None), ("csr_stream", True, (0, 3, -1, -1))):
    A = sa.sp_matrix_mg(rp, ci, v)
    if not fold:
        A.set_const_slots(False)
    if cfg:
        A.set_kernel_config(*cfg)
    A.setup(sa.default_params(print_setup=0, print_solve=0))
    bd, xd = A.dev_alloc(8 * N), A.dev_alloc(8 * N)
    A.h2d(bd, np.ones(N))
    for rnd in range(2):
        for alt in (0, 2):
            A.set_alternate_sweeps(alt)
            t = [A.bench_op("jacobi_pingpong", l, 30) * 1e6 for l in (0, 1)]
            best = 0.0
            for rep in range(2):
                A.h2d(xd, np.zeros(N))
                h, it, sec, rc = A.solve_dev("pcg", bd, xd)
                best = max(best, it / sec)
            print(f"{name} r{rnd} alternate={alt}: ping-pong sweeps L0 {t[0]:.1f} L1 {t[1]:.1f} us; pcg {best:.1f} it/s", flush=True)
    A.close()
