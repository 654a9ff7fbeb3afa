#!/usr/bin/env python3
"""AMG-PCG rate and coarse-solve time as a function of where the hierarchy stops (how many levels before
the direct solver takes over): the extended 13-level default with its dense 2468-row inverse against
shallower hierarchies whose coarsest level goes to the block-tridiagonal device solver.
Usage: python tools/depth_bench.py [--n 216] [--levels 7 8 9 10 13] [--steps 30]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=216)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--levels", type=int, nargs="*", default=[7, 8, 9, 10, 13])
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    rp, ci, v = problems.poisson3d(a.n) if a.dim == 3 else problems.poisson2d(a.n)
    n = len(rp) - 1
    b = np.ones(n)
    print(f"{'levels':>6s} {'coarsest':>9s} {'form':>6s} {'blocks':>7s} {'B':>6s} {'MB':>8s} {'setup s':>8s} {'coarse us':>10s} {'it/s':>8s} {'iters':>6s} {'solve ms':>9s}")
    for ml in a.levels:
        t0 = time.time()
        A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0, max_levels=ml, coarse_limit=(1 << 30) if ml < 13 else 40000))
        t_setup = time.time() - t0
        info = A.coarse_info()
        tc = A.bench_op("coarse", A.nlevels - 1, 20)
        bd = A.dev_alloc(8 * n)
        xd = A.dev_alloc(8 * n)
        A.h2d(bd, b)
        A.h2d(xd, np.zeros(n))
        A.set_stopping(0.0, 100000, 1 << 30)
        A.krylov_init_dev("pcg", bd, xd)
        A.krylov_step_dev(3)
        A.sync()
        t1 = time.perf_counter()
        A.krylov_step_dev(a.steps)
        A.sync()
        its = a.steps / (time.perf_counter() - t1)
        A.set_stopping(1e-8, 100000, 1)
        A.h2d(xd, np.zeros(n))
        h, it, sec, rc = A.solve_dev("pcg", bd, xd)
        print(f"{A.nlevels:6d} {info['rows']:9d} {'dense' if info['dense'] else 'bt':>6s} {info['nblocks']:7d} {info['block']:6d} {info['bytes'] / 1e6:8.1f} "
              f"{t_setup:8.2f} {tc * 1e6:10.1f} {its:8.1f} {it:6d} {sec * 1e3:9.2f}", flush=True)
        A.close()


if __name__ == "__main__":
    main()
