#!/usr/bin/env python3
"""Per-operator, per-level achieved bandwidth of the resident hierarchy (HIP events, C ABI
sparsh_bench_op).  Usage: python tools/kernel_bench.py [--n 216] [--reps 20] [--levels 4]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=216)
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--ordering", default="morton")
    ap.add_argument("--fem", type=int, default=0, help="use the unstructured P1-FEM stand-in with this many points")
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--cfg", type=int, nargs=4, default=None, metavar=("KIND", "VEC", "NT", "REMAP"))
    ap.add_argument("--remaps", type=int, nargs="*", default=None, help="sweep XCD remap modes for the Jacobi kernel")
    ap.add_argument("--variants", action="store_true", help="A/B all kernel families (interleaved rounds, one process)")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--vlevels", type=int, nargs="*", default=[0, 2, 4])
    ap.add_argument("--no-const", action="store_true", help="build the sliced-diagonal layout without constant-slot folding")
    ap.add_argument("--kinds", type=int, nargs="*", default=[3, 2], help="kernel families of the --remaps sweep")
    ap.add_argument("--tile-ab", action="store_true", help="A/B the LDS-tiled table kernel against the untiled one, level by level (interleaved rounds)")
    a = ap.parse_args()
    if a.fem:
        rp, ci, v = problems.fem_unstructured(a.fem, ordering=a.ordering)
    else:
        rp, ci, v = problems.poisson3d(a.n) if a.dim == 3 else problems.poisson2d(a.n)
    A = sa.sp_matrix_mg(rp, ci, v)
    if a.no_const:
        A.set_const_slots(False)
    A.set_index_compression(2)  # every operator also carries 16-bit delta-coded column indices (read by vec = 4 only)
    A.setup(sa.default_params(print_setup=0, print_solve=0))
    if a.cfg:
        A.set_kernel_config(*a.cfg)
    if a.remaps:
        print(f"{'config':28s} {'op':8s} {'lvl':>3s} {'us':>9s} {'GB/s':>8s} {'frac8T':>7s}")
        for rnd in range(a.rounds):
            for (k, v) in [(kk, 0) for kk in a.kinds]:
                for nt in (1, 0):
                    for rm in a.remaps:
                        A.set_kernel_config(kind=k, vec=v, nt=nt, remap=rm)
                        for l in a.vlevels:
                            if l >= A.nlevels:
                                continue
                            i = A.level_info(l)
                            n, nnz = i["nrow"], i["nnz"]
                            sec = A.bench_op("jacobi", l, a.reps)
                            gbs = (12 * nnz + 36 * n) / sec / 1e9
                            print(f"kind={k} vec={v} nt={nt} remap={rm:<5d} r{rnd:<2d} jacobi   {l:3d} {sec * 1e6:9.1f} {gbs:8.1f} {gbs / 8000:7.3f}", flush=True)
        A.set_kernel_config()
        return
    if a.variants:
        print(f"{'config':28s} {'op':8s} {'lvl':>3s} {'us':>9s} {'GB/s':>8s} {'frac8T':>7s}")
        cfgs = [(k, v, nt, rm) for k in a.kinds for v in ((0, 1, 2, 4) if k == 0 else ((0, 1) if k == 1 else (0,))) for nt in (1, 0) for rm in (1, 16)]
        for rnd in range(a.rounds):
            for (k, v, nt, rm) in cfgs:
                A.set_kernel_config(kind=k, vec=v, nt=nt, remap=rm)
                for l in a.vlevels:
                    if l >= A.nlevels:
                        continue
                    i = A.level_info(l)
                    n, nnz = i["nrow"], i["nnz"]
                    for op, nbytes in (("jacobi", 12 * nnz + 36 * n), ("spmv", 12 * nnz + 20 * n)):
                        sec = A.bench_op(op, l, a.reps)
                        gbs = nbytes / sec / 1e9
                        print(f"kind={k} vec={v} nt={nt} remap={rm} r{rnd:<3d} {op:8s} {l:3d} {sec * 1e6:9.1f} {gbs:8.1f} {gbs / 8000:7.3f}", flush=True)
        A.set_kernel_config()
        return
    if a.tile_ab:
        A.set_tile(True)
        print("tile rows per level:", [A.level_tile_rows(l) for l in range(A.nlevels)])
        print(f"{'op':9s} {'lvl':>3s} {'rows':>10s} {'tiled us':>9s} {'plain us':>9s} {'ratio':>6s}")
        for l in range(min(a.levels, A.nlevels)):
            A.set_tile(True)
            if A.level_tile_rows(l) == 0:
                continue
            for op in ("jacobi", "spmv", "residual"):
                tt, tp = [], []
                for rnd in range(a.rounds):
                    A.set_tile(True)
                    tt.append(A.bench_op(op, l, a.reps))
                    A.set_tile(False)
                    tp.append(A.bench_op(op, l, a.reps))
                print(f"{op:9s} {l:3d} {A.level_info(l)['nrow']:10d} {min(tt) * 1e6:9.2f} {min(tp) * 1e6:9.2f} {min(tt) / min(tp):6.3f}", flush=True)
        return
    print("level formats (3 sliced diagonals, 2 sliced ELL, 0 CSR-stream):", [A.level_format(l)[0] for l in range(A.nlevels)])
    print("sliced-diagonal slots / value blocks per level:", [A.level_layout(l) for l in range(A.nlevels)])
    print(f"{'op':10s} {'lvl':>3s} {'rows':>10s} {'nnz':>10s} {'us':>9s} {'GB/s':>8s} {'frac8T':>7s}")
    for l in range(min(a.levels, A.nlevels)):
        i = A.level_info(l)
        n, nnz, pn, pnnz = i["nrow"], i["nnz"], i["p_ncol"], i["p_nnz"]
        model = {
            "spmv": 12 * nnz + 20 * n,
            "jacobi": 12 * nnz + 36 * n,
            "residual": 12 * nnz + 28 * n,
            "dot": 16 * n,
            "axpby": 24 * n,
        }
        if l + 1 < A.nlevels:
            model["restrict"] = 12 * pnnz + 8 * n + 12 * pn
            model["prolong"] = 12 * pnnz + 20 * n + 8 * pn
        for op, nbytes in model.items():
            sec = A.bench_op(op, l, a.reps)
            gbs = nbytes / sec / 1e9
            print(f"{op:10s} {l:3d} {n:10d} {nnz:10d} {sec * 1e6:9.1f} {gbs:8.1f} {gbs / 8000:7.3f}", flush=True)
    nL = A.level_info(A.nlevels - 1)["nrow"]
    sec = A.bench_op("coarse", A.nlevels - 1, a.reps)
    print(f"{'coarse':10s} {A.nlevels - 1:3d} {nL:10d} {nL * nL:10d} {sec * 1e6:9.1f} {8 * nL * nL / sec / 1e9:8.1f}")


if __name__ == "__main__":
    main()
