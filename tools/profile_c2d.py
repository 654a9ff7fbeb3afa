#!/usr/bin/env python3
"""BASELINE configs[1] under the profiler: 5-pt 2D Poisson 1000^2, AMG V(7,7) cycles to 1e-8.
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/profile_c2d.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

rp, ci, v = problems.poisson2d(1000)
n = len(rp) - 1
A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0))
bd, xd = A.dev_alloc(8 * n), A.dev_alloc(8 * n)
A.h2d(bd, np.ones(n))
for rep in range(3):
    A.h2d(xd, np.zeros(n))
    h, it, sec, rc = A.solve_dev("amg", bd, xd)
    print(f"AMG: {it} V-cycles in {sec:.4f} s = {it / sec:.1f} V-cycles/s, residual {h[-1]:.3e}", file=sys.stderr)
