#!/usr/bin/env python3
"""BASELINE configs[4] (irregular-nnz SpMV stress; P1-FEM stand-in for parabolic_fem, 525 825 rows): achieved
GB/s of SpMV / fused Jacobi sweep on the finest level for every kernel family the operator qualifies for, priced
with SURVEY section 8d's CSR model (12 nnz + 20 n / 12 nnz + 36 n), plus AMG-PCG and AMG-BiCGStab rates.
Usage: python tools/config4_spmv.py [--n 525825] [--ordering morton|random]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=525825)
    ap.add_argument("--ordering", default="morton")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rp, ci, v = problems.fem_unstructured(a.n, ordering=a.ordering)
    n, nnz = len(rp) - 1, int(rp[-1])
    lens = np.diff(rp)
    out = {"rows": n, "nnz": nnz, "row_length": {"min": int(lens.min()), "max": int(lens.max()), "mean": round(float(lens.mean()), 2)},
           "ordering": a.ordering, "families": {}}
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0))
    out["levels"] = [A.level_info(l)["nrow"] for l in range(A.nlevels)]
    out["coarsest"] = A.coarse_info()
    for kind, vec in ((0, 1), (0, 2), (0, 0), (1, 1), (2, 0)):
        A.set_kernel_config(kind=kind, vec=vec)
        name = A.level_kernel(0)
        key = f"kind{kind}_vec{vec}:{name}"
        if kind == 2 and name != "sell_kernel":
            out["families"][key] = "operator does not qualify for the sliced-ELL mirror (padding > 12.5 %)"
            continue
        ts = min(A.bench_op("spmv", 0, a.reps) for _ in range(3))
        tj = min(A.bench_op("jacobi", 0, a.reps) for _ in range(3))
        out["families"][key] = {"spmv_us": round(ts * 1e6, 2), "spmv_GBps": round((12 * nnz + 20 * n) / ts / 1e9, 1),
                                "spmv_frac_8TBps": round((12 * nnz + 20 * n) / ts / 8e12, 4),
                                "jacobi_us": round(tj * 1e6, 2), "jacobi_GBps": round((12 * nnz + 36 * n) / tj / 1e9, 1),
                                "jacobi_frac_8TBps": round((12 * nnz + 36 * n) / tj / 8e12, 4)}
    A.set_kernel_config()
    out["default_kernel"] = A.level_kernel(0)
    b = np.random.default_rng(4).standard_normal(n) * 1e-3
    bd, xd = A.dev_alloc(8 * n), A.dev_alloc(8 * n)
    A.h2d(bd, b)
    for method in ("pcg", "pbicg"):
        A.h2d(xd, np.zeros(n))
        h, it, sec, rc = A.solve_dev(method, bd, xd)
        out[method] = {"iterations": it, "seconds": round(sec, 4), "iterations_per_s": round(it / sec, 1), "rc": rc, "final_residual": float(h[-1])}
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
