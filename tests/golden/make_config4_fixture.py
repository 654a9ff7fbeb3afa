#!/usr/bin/env python3
"""Regenerates tests/golden/config4_fem_oracle.json: the CPU oracle (oracle/amg_oracle.c, restating
Solver_PBiCG_1 / Solver_PCG_1 of the reference, src/AMG_main_solvers.cpp:358-458, 107-167) on the FULL-SIZE
BASELINE configs[4] stand-in -- P1-FEM  M + dt K  on a Delaunay mesh of 525 825 points
(sparsh_amg_amd.problems.fem_unstructured, seed 20240607, dt = 1e-2; SuiteSparse parabolic_fem itself is not
in the image).  Two right-hand sides:
  * b = 1e-3 (constant): AMG-BiCGStab BREAKS DOWN (the constant vector is almost an eigenvector of
    M + dt K, the shadow residual r0 stays parallel to it and r.r0 / (Ap.r0) ends as 0/0; the reference
    has no breakdown checks: `while (res > tol1)`, src/AMG_main_solvers.cpp:397) -- the residual becomes NaN;
  * b = 1e-3 * N(0,1) (numpy default_rng(4)): converges.
Takes ~6 minutes of 8-thread CPU time.  Data only: iteration counts, residual heads/tails, solution checks.

  --extra  (round 3; adds to the existing file, ~10 more minutes): for the random right-hand side
  * true_residuals_k1_6: ||b - A x_k||_2 computed with scipy from the oracle's iterate after exactly k = 1..6 iterations of
    AMG-PBiCGStab and AMG-PCG (runs capped at k iterations) -- what the device's iterates are held to, independent of either
    side's residual recurrence;
  * thread_sensitivity: AMG-PBiCGStab iteration counts of the oracle itself with 1, 2 and 8 OpenMP threads (its dot products are
    chunked per thread, so the summation order changes) -- the spread the loose iteration-count band of the GPU test has to cover."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from sparsh_amg_amd import problems  # noqa: E402


def main():
    rp, ci, v = problems.fem_unstructured(525825)
    n = len(rp) - 1
    O = oracle.Csr(rp, ci, v)
    out = {"_provenance": "oracle/amg_oracle.c run by tests/golden/make_config4_fixture.py (CPU restatement of the reference algorithm; "
                          "NOT an output of the reference itself: parity unpinned for this input)",
           "input": "problems.fem_unstructured(525825), seed 20240607, dt 1e-2, morton ordering", "nrow": n, "nnz": int(rp[-1])}
    prm = oracle.params(threads=8, max_iter=2000)
    for key, b in (("constant_rhs", np.full(n, 1e-3)), ("random_rhs", np.random.default_rng(4).standard_normal(n) * 1e-3)):
        for method in ("pbicg", "pcg"):
            x, h = oracle.solve(method, O, b, prm=prm)
            nan_at = int(np.flatnonzero(~np.isfinite(h))[0]) if np.any(~np.isfinite(h)) else None
            out.setdefault(key, {})[method] = {
                "iterations": len(h), "first_nonfinite": nan_at, "hist_head": [float(t) for t in h[:12]],
                "hist_tail": [float(t) if np.isfinite(t) else None for t in h[-3:]],
                "xnorm": float(np.linalg.norm(x)) if np.all(np.isfinite(x)) else None,
                "x_head": [float(t) if np.isfinite(t) else None for t in x[:4]]}
            print(key, method, len(h), nan_at, h[-3:], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "config4_fem_oracle.json"), "w") as f:
        json.dump(out, f, indent=1)


def extra():
    import scipy.sparse as sp

    path = os.path.join(ROOT, "tests", "golden", "config4_fem_oracle.json")
    with open(path) as f:
        out = json.load(f)
    rp, ci, v = problems.fem_unstructured(525825)
    n = len(rp) - 1
    O = oracle.Csr(rp, ci, v)
    S = sp.csr_matrix((v, ci, rp), shape=(n, n))
    b = np.random.default_rng(4).standard_normal(n) * 1e-3
    for method in ("pbicg", "pcg"):
        tr, cnt, rec = [], [], []
        for k in range(1, 7):
            x, h = oracle.solve(method, O, b, prm=oracle.params(threads=8, max_iter=k))
            tr.append(float(np.linalg.norm(b - S @ x)))
            cnt.append(len(h))
            rec.append(float(h[-1]))
            print(method, k, len(h), tr[-1], h[-1], flush=True)
        out["random_rhs"][method]["true_residuals_k1_6"] = tr
        out["random_rhs"][method]["history_length_at_cap_k1_6"] = cnt
        out["random_rhs"][method]["recurrence_residual_at_cap_k1_6"] = rec
    sens = {}
    for th in (1, 2, 8):
        x, h = oracle.solve("pbicg", O, b, prm=oracle.params(threads=th, max_iter=2000))
        sens[str(th)] = {"iterations": len(h), "final_residual": float(h[-1]), "true_residual": float(np.linalg.norm(b - S @ x))}
        print("threads", th, sens[str(th)], flush=True)
    out["random_rhs"]["pbicg"]["thread_sensitivity"] = sens
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


def extra2():
    """rounding_sensitivity: the oracle's own AMG-PBiCGStab iteration count when the right-hand side is perturbed at the level of
    one rounding error (b_i * (1 + 2^-52 * s_i), s_i in {-1, 0, 1} from default_rng(seed)): the oracle's reductions are chunked
    independently of the thread count (1, 2 and 8 threads give the same 424 iterations bit for bit), so this -- not a thread sweep --
    is what shows how far rounding alone moves the count on this operator."""
    path = os.path.join(ROOT, "tests", "golden", "config4_fem_oracle.json")
    with open(path) as f:
        out = json.load(f)
    rp, ci, v = problems.fem_unstructured(525825)
    n = len(rp) - 1
    O = oracle.Csr(rp, ci, v)
    b = np.random.default_rng(4).standard_normal(n) * 1e-3
    runs = []
    for seed in (11, 12, 13):
        s = np.random.default_rng(seed).integers(-1, 2, size=n).astype(np.float64)
        bp = b * (1.0 + 2.0 ** -52 * s)
        x, h = oracle.solve("pbicg", O, bp, prm=oracle.params(threads=8, max_iter=2000))
        runs.append({"seed": seed, "iterations": len(h), "final_residual": float(h[-1])})
        print(runs[-1], flush=True)
    out["random_rhs"]["pbicg"]["rounding_sensitivity"] = runs
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    if "--extra2" in sys.argv:
        extra2()
    elif "--extra" in sys.argv:
        extra()
    else:
        main()
