"""Same-process A/B of the fused residual + restriction launch (sparsh_set_paired_restriction) or of the fused last post-sweep
+ prolongation (sparsh_set_fused_prolongation) inside the PCG iteration.

    python tools/fusion_ab.py [--what pair|prolong|diag|box2] [--n 216] [--dim 3] [--iters 96] [--reps 3] [--out FILE]

One handle, one hierarchy; the setting is toggled between timed runs of `iters` PCG iterations (restart every 48, as bench.py
does), alternating on / off so that drift of the box hits both sides alike.  Prints it/s per run and the median of each side.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import sparsh_amg_amd as sa  # noqa: E402
from sparsh_amg_amd import problems  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=216)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--iters", type=int, default=96)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--out", default=None)
    ap.add_argument("--what", default="pair", choices=["pair", "prolong", "diag", "box2", "box1", "zero", "graph", "deferx"])
    args = ap.parse_args()
    rp, ci, v = problems.poisson3d(args.n) if args.dim == 3 else problems.poisson2d(args.n)
    n = len(rp) - 1
    prm = sa.default_params(print_setup=0, print_solve=0, tol=0.0, check_every=1 << 30)
    A = sa.sp_matrix_mg(rp, ci, v).setup(prm)
    if args.what == "graph":  # hipGraph replay of the iteration (sparsh_params.use_graph) against eager launches: two handles, same process
        B = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0, tol=0.0, check_every=1 << 30, use_graph=1))
        handles = {True: B, False: A}
    if args.what == "pair":
        paired = [l for l in range(A.nlevels - 1) if A.level_paired(l)]
        toggle = A.set_paired_restriction
    elif args.what == "box2":
        paired = [(l, A.level_double_sweep(l)) for l in range(A.nlevels) if A.level_double_sweep(l)["grid"][0] > 0]
        toggle = lambda on: A.set_double_sweep(1 if on else 0)  # noqa: E731
    elif args.what == "graph":
        paired = []

        def toggle(on):
            cur[0] = handles[bool(on)]
    elif args.what == "deferx":
        paired = []
        toggle = A.set_deferred_x
    elif args.what == "zero":
        paired = [l for l in range(A.nlevels) if A.level_double_sweep(l)["on"]]
        toggle = A.set_zero_start
    elif args.what == "box1":
        paired = [(l, A.level_marching_ops(l)) for l in range(A.nlevels) if A.level_marching_ops(l)["points_per_thread"] > 0]
        toggle = lambda on: A.set_marching_ops(1 if on else 0)  # noqa: E731
    elif args.what == "diag":
        paired = [l for l in range(A.nlevels) if A.level_constant_diagonal(l)[0]]
        toggle = A.set_constant_diagonal
    else:
        paired = [l for l in range(A.nlevels) if A.level_prolong_fused(l)]
        toggle = A.set_fused_prolongation
    print(f"# {args.dim}D n={args.n}: {n} rows, {A.nlevels} levels, {args.what}: fused levels {paired}", flush=True)
    bd, xd = A.dev_alloc(8 * n), A.dev_alloc(8 * n)
    A.h2d(bd, np.ones(n))
    cur = [A]

    def run(iters):
        H = cur[0]
        left = iters
        while left > 0:
            H.dev_fill(xd, n, 0.0)
            H.krylov_init_dev("pcg", bd, xd)
            m = min(left, 48)
            done, _ = H.krylov_step_dev(m)
            assert done == m
            left -= m
        H.sync()

    rates = {True: [], False: []}
    hist = {}
    for on in (True, False):
        toggle(on)
        run(48)
        hist[on] = np.array(cur[0].krylov_history())
    if args.what == "box1":  # the fused dot products are summed in another order: same histories to rounding
        k = min(30, len(hist[True]), len(hist[False]))
        assert np.allclose(hist[True][:k], hist[False][:k], rtol=1e-9), "histories differ beyond rounding"
    else:
        assert np.array_equal(hist[True], hist[False]), "histories differ between the fused and the separate launches"
    for rep in range(args.reps):
        for on in (True, False):
            toggle(on)
            run(48)  # warm-up under this setting
            t0 = time.perf_counter()
            run(args.iters)
            dt = time.perf_counter() - t0
            # every 48-iteration segment carries one restart (about one more iteration of work): both sides alike
            rates[on].append(args.iters / dt)
            print(f"rep {rep} fused={int(on)}: {args.iters / dt:8.1f} it/s", flush=True)
    rec = {"problem": f"poisson{args.dim}d n={args.n}", "rows": n, "levels": A.nlevels, "what": args.what, "fused_levels": paired,
           "iters_per_run": args.iters, "it_per_s_fused": [round(r, 1) for r in rates[True]],
           "it_per_s_separate": [round(r, 1) for r in rates[False]],
           "median_fused": round(float(np.median(rates[True])), 1), "median_separate": round(float(np.median(rates[False])), 1),
           "histories_bitwise_equal": True}
    rec["gain"] = round(rec["median_fused"] / rec["median_separate"] - 1.0, 4)
    print(json.dumps(rec), flush=True)
    if args.out:
        with open(args.out, "a") as f:
            f.write(json.dumps(rec) + "\n")
    A.close()


if __name__ == "__main__":
    main()
