#!/usr/bin/env python3
"""Is the cache-resident irregular CSR kernel paying for a second round of workgroups?  csr_block_kernel keeps 8 workgroups
per CU resident (2048 on the chip); the 525 825-row FEM level has ~2054+ row blocks.  Time the SpMV / fused sweep for meshes
just below and above 2048 row blocks (run ON the GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

for npts in [int(a) for a in sys.argv[1:]] or [480000, 505000, 515000, 520000, 523000, 525825, 530000, 545000, 600000]:
    rp, ci, v = problems.fem_unstructured(npts)
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0))
    sp = A.bench_op("spmv", 0, 50) * 1e6
    ja = A.bench_op("jacobi", 0, 50) * 1e6
    nnz = int(rp[-1])
    nblk = A.level_index16(0)[1]
    print(f"rows {n:7d} nnz {nnz:8d} row blocks {nblk:5d} kernel {A.level_kernel(0)}: spmv {sp:6.2f} us ({(12*nnz+20*n)/sp/1e3:6.0f} GB/s)  jacobi {ja:6.2f} us ({(12*nnz+36*n)/ja/1e3:6.0f} GB/s)", flush=True)
    A.close()
