#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one MI355X (they are parity-test cases in
tests/test_gpu_configs.py; this prints their rates for DESIGN.md / profiles/):
  C2D  5-pt 2D Poisson 1000^2 (1 M rows): AMG V(7,7) cycles/s and AMG-PCG iterations/s
  C3D  7-pt 3D Poisson 216^3 (10 M rows): AMG V(7,7) cycles/s (bench.py reports the PCG rate)
  CU   unstructured P1-FEM M + dt K stand-in, 525 825 rows: AMG-PBiCGStab iterations/s
SPARSH_MTX=/path/to/file.mtx adds that MatrixMarket file as a further CU case.
All rates are solve-phase only (hierarchy resident, vectors in HBM), full solves to 1e-8.
Usage: python tools/config_bench.py > profiles/r02_configs.json"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems


def run(name, rp, ci, v, methods, rhs="ones", **params):
    n = len(rp) - 1
    out = {"rows": n, "nnz": int(rp[-1])}
    A = sa.sp_matrix_mg(rp, ci, v)
    if os.environ.get("SPARSH_COARSE_BLOCK"):  # A/B of the block-tridiagonal coarse solver's block size
        A.set_coarse_block(int(os.environ["SPARSH_COARSE_BLOCK"]))
    if os.environ.get("SPARSH_COARSE_FORM"):   # "nd[,leaf[,merge_rows]]" or "bt": A/B of the two device factorisations
        f = os.environ["SPARSH_COARSE_FORM"].split(",")
        A.set_coarse_form(f[0], int(f[1]) if len(f) > 1 else 0, int(f[2]) if len(f) > 2 else -1, int(f[3]) if len(f) > 3 else -1)
    A.setup(sa.default_params(print_setup=0, print_solve=0, **params))
    out["levels"] = [A.level_info(l)["nrow"] for l in range(A.nlevels)]
    out["level_kernels"] = [A.level_kernel(l) for l in range(A.nlevels - 1)]
    out["coarsest"] = A.coarse_info()
    out["coarse_solve_us"] = round(A.bench_op("coarse", A.nlevels - 1, 20) * 1e6, 1)
    out["setup_seconds"] = round(A.setup_seconds, 3)
    out["rhs"] = rhs
    # constant right-hand side for the Poisson cases (the reference's goldens use b = 1); the FEM stand-in gets a random
    # one: on M + dt K the constant vector is almost an eigenvector and BiCGStab breaks down on it (tests/test_gpu_configs.py)
    b = np.ones(n) if rhs == "ones" else np.random.default_rng(4).standard_normal(n) * 1e-3
    bd, xd = A.dev_alloc(8 * n), A.dev_alloc(8 * n)
    A.h2d(bd, b)
    for m in methods:
        best = None
        err = None
        for rep in range(3):
            A.h2d(xd, np.zeros(n))
            try:
                h, it, sec, rc = A.solve_dev(m, bd, xd)
            except sa.SparshError as e:  # e.g. BiCGStab breakdown (0/0), which the CPU oracle reproduces
                err = str(e)
                break
            if best is None or sec < best[1]:
                best = (it, sec, float(h[-1]), rc)
        if err is not None:
            out[m] = {"error": err}
            print(f"[{name}] {m}: {err}", file=sys.stderr, flush=True)
            continue
        it, sec, res, rc = best
        unit = "V-cycles/s" if m == "amg" else "iterations/s"
        out[m] = {"count": it, "seconds": round(sec, 5), "rate": round(it / sec, 1), "unit": unit, "final_residual": res, "rc": rc}
        print(f"[{name}] {m}: {it} in {sec:.4f}s = {it / sec:.1f} {unit}, residual {res:.3e}", file=sys.stderr, flush=True)
    A.close()
    return out


def main():
    res = {}
    only = [a for a in sys.argv[1:] if not a.startswith("-")]  # substrings of case names to run (default: all)

    def want(name):
        return not only or any(o in name for o in only)
    if want("C2D_poisson2d_1000"):
        res["C2D_poisson2d_1000"] = run("C2D", *problems.poisson2d(1000), ["amg", "pcg"])  # default: the reference's 6 levels + block-tridiagonal direct solve
    if want("C2D_poisson2d_1000_extended_hierarchy"):
        res["C2D_poisson2d_1000_extended_hierarchy"] = run("C2D ext", *problems.poisson2d(1000), ["amg", "pcg"], coarse_limit=8192, extend_until=4000)
    if want("C3D_poisson3d_100"):
        res["C3D_poisson3d_100"] = run("100^3", *problems.poisson3d(100), ["amg", "pcg"])
    if want("C3D_poisson3d_100_extended_hierarchy"):
        res["C3D_poisson3d_100_extended_hierarchy"] = run("100^3 ext", *problems.poisson3d(100), ["amg", "pcg"], coarse_limit=8192, extend_until=4000)
    # the reference's own 6-level policy one size further than the default coarse_limit allows: 136^3 leaves a
    # 78 608-row coarsest level; next to it the default (hierarchy extended until <= 40 000 rows)
    if want("C3D_poisson3d_136_reference_policy"):
        res["C3D_poisson3d_136_reference_policy"] = run("136^3 ref", *problems.poisson3d(136), ["amg", "pcg"], coarse_limit=100000)
    if want("C3D_poisson3d_136_default"):   # since the factor-size rule (coarse_factor_mb): the reference's 6 levels
        res["C3D_poisson3d_136_default"] = run("136^3", *problems.poisson3d(136), ["amg", "pcg"])
    if want("C3D_poisson3d_136_row_rule"):  # coarse_limit rows alone: extended to 39 304 rows
        res["C3D_poisson3d_136_row_rule"] = run("136^3 rows", *problems.poisson3d(136), ["amg", "pcg"], coarse_factor_mb=0)
    if want("C3D_poisson3d_216"):
        res["C3D_poisson3d_216"] = run("C3D", *problems.poisson3d(216), ["amg", "pcg"])
    if want("C3D_poisson3d_216_round2_hierarchy"):  # extended until <= limit_upper rows: 13 levels, dense 2468-row coarsest level
        res["C3D_poisson3d_216_round2_hierarchy"] = run("C3D 13 levels", *problems.poisson3d(216), ["amg", "pcg"], extend_until=4000)
    if want("C3D_poisson3d_216_8_levels"):          # stop one level earlier: 78 732-row coarsest level
        res["C3D_poisson3d_216_8_levels"] = run("C3D 8 levels", *problems.poisson3d(216), ["amg", "pcg"], coarse_limit=80000)
    if want("C3D_poisson3d_216_reference_policy"):  # the reference's own 6 levels: 314 928 rows to the direct solver (what PARDISO gets in the reference)
        res["C3D_poisson3d_216_reference_policy"] = run("C3D 6 levels", *problems.poisson3d(216), ["pcg"], coarse_limit=1 << 30)
    # nu = 6 sweeps: what the reference's GPU path effectively runs (smooth_iter without the +1 of the CPU path)
    if want("C3D_poisson3d_216_nu6"):
        res["C3D_poisson3d_216_nu6"] = run("C3D nu=6", *problems.poisson3d(216), ["amg", "pcg"], sweeps=6)
    # Beck's classical C/F interpolation instead of HEM aggregation (general multi-entry P and R, denser
    # coarse operators: CSR-stream / sliced-ELL kernels on the coarse levels)
    if want("C3D_poisson3d_100_beck"):
        res["C3D_poisson3d_100_beck"] = run("C3D 100^3 Beck", *problems.poisson3d(100), ["amg", "pcg"], coarsening=1)
    if want("C2D_poisson2d_1000_beck"):
        res["C2D_poisson2d_1000_beck"] = run("C2D Beck", *problems.poisson2d(1000), ["amg", "pcg"], coarsening=1)
    if want("C3D_poisson3d_216_beck"):
        res["C3D_poisson3d_216_beck"] = run("C3D 216^3 Beck", *problems.poisson3d(216), ["amg", "pcg"], coarsening=1)
    # larger inputs (only when asked for by name): robustness of the nested-dissection solver on big 2D / unstructured coarsest levels
    if only and want("BIG_poisson2d_3000_reference_policy"):   # 9 M rows, the reference's 6 levels: 281 250-row 2D coarsest level to the direct solver
        res["BIG_poisson2d_3000_reference_policy"] = run("2D 3000^2 ref", *problems.poisson2d(3000), ["pcg"], coarse_limit=1 << 30)
    if only and want("BIG_poisson2d_3000_default"):   # (= the reference policy since the factor-size rule)
        res["BIG_poisson2d_3000_default"] = run("2D 3000^2", *problems.poisson2d(3000), ["pcg"])
    if only and want("BIG_poisson2d_3000_row_rule"):
        res["BIG_poisson2d_3000_row_rule"] = run("2D 3000^2 rows", *problems.poisson2d(3000), ["pcg"], coarse_factor_mb=0)
    if only and want("BIG_fem_unstructured_2M"):
        res["BIG_fem_unstructured_2M"] = run("FEM 2M", *problems.fem_unstructured(2000000, seed=3), ["pcg"], rhs="random")
    if only and want("BIG_fem_unstructured_2M_reference_policy"):
        res["BIG_fem_unstructured_2M_reference_policy"] = run("FEM 2M ref", *problems.fem_unstructured(2000000, seed=3), ["pcg"], rhs="random", coarse_limit=1 << 30)
    mtx = os.environ.get("SPARSH_MTX")  # e.g. SuiteSparse parabolic_fem.mtx when it is on the box
    if mtx and os.path.exists(mtx):
        if want("CU_"):
            res["CU_" + os.path.basename(mtx)] = run("CU file", *problems.read_matrix_market(mtx), ["pbicg", "pcg"])
    if want("CU_fem_unstructured_525825"):
        res["CU_fem_unstructured_525825"] = run("CU", *problems.fem_unstructured(), ["pbicg", "pcg"], rhs="random")
    if want("CU_fem_unstructured_60000"):
        res["CU_fem_unstructured_60000"] = run("CU60k", *problems.fem_unstructured(60000, seed=7), ["pbicg", "pcg"], rhs="random")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
