#!/usr/bin/env python3
"""Run N coarse-level direct solves of one configuration (for rocprofv3 --kernel-trace --stats: which kernel of the
block-tridiagonal solve takes what).  Usage: python tools/coarse_profile.py [2d|3d|fem] [reps] [block]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

which = sys.argv[1] if len(sys.argv) > 1 else "2d"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rp, ci, v = {"2d": lambda: problems.poisson2d(1000), "3d": lambda: problems.poisson3d(100), "fem": problems.fem_unstructured}[which]()
A = sa.sp_matrix_mg(rp, ci, v)
if block:
    A.set_coarse_block(block)
A.setup(sa.default_params(print_setup=0, print_solve=0))
print(A.coarse_info(), flush=True)
sec = A.bench_op("coarse", A.nlevels - 1, reps)
print(f"coarse solve: {sec * 1e6:.1f} us", flush=True)
A.close()
