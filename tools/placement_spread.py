#!/usr/bin/env python3
"""Run-to-run spread of the finest-level sweep and what the setup-time placement search does about it: 33 timed AMG-PCG
iterations with the profile events on, with the search (default) and without, in ONE process (two handles), then the same
sweep launched back to back.  Start it several times: within a process the figures are stable, between processes they are
not (profiles/r02_finest_sweep_placement_luck.txt)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

rp, ci, v = problems.poisson3d(216)
N = len(rp) - 1
for search in (0, 1, 0, 1):
    A = sa.sp_matrix_mg(rp, ci, v).set_placement_search(bool(search)).setup(sa.default_params(print_setup=0, print_solve=0, tol=0.0))
    bd, xd = A.dev_alloc(8 * N), A.dev_alloc(8 * N)
    A.h2d(bd, np.ones(N))
    out = []
    for rnd in range(2):
        A.h2d(xd, np.zeros(N))
        A.krylov_init_dev("pcg", bd, xd)
        A.krylov_step_dev(3)
        A.profile(True)
        A.sync()
        t = time.perf_counter()
        A.krylov_step_dev(30)
        A.sync()
        t = time.perf_counter() - t
        A.profile(False)
        p = A.profile_read()
        out.append(f"{30 / t:.1f} it/s, finest sweep in-solve {p['seconds'] / p['launches'] * 1e6:.1f} us")
    print(f"search={search} {A.placement_info()} setup {A.setup_seconds:.2f}s: " + " | ".join(out), flush=True)
    A.close()
