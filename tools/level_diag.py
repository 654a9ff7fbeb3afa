#!/usr/bin/env python3
"""Per-level layout / kernel / placement of the 216^3 hierarchy and the fused sweep timed alone on each (run ON the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems
rp, ci, v = problems.poisson3d(int(sys.argv[1]) if len(sys.argv) > 1 else 216)
A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0))
for l in range(A.nlevels - 1):
    i = A.level_info(l)
    fmt, stored = A.level_format(l)
    slots, vblocks, meta = A.level_layout(l)
    nt, remap = A.level_placement(l)
    ts = [A.bench_op(op, l, 50) * 1e6 for op in ("jacobi", "jacobi_pingpong", "spmv", "residual")]
    print(f"level {l}: rows {i['nrow']:9d} kernel {A.level_kernel(l):16s} fmt {fmt} stored {stored:9d} slots {slots:8d} value_blocks {vblocks} meta {meta:9d} nt {int(nt)} remap {remap:3d} | jacobi {ts[0]:6.2f} pingpong {ts[1]:6.2f} spmv {ts[2]:6.2f} resid {ts[3]:6.2f} us", flush=True)
