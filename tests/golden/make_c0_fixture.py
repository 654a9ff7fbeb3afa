#!/usr/bin/env python3
"""Regenerates tests/golden/c0_matrix.npz from the reference's bundled input files.

Run in the build container only (the files live in /root/reference, which does not travel):
    python tests/golden/make_c0_fixture.py

The fixture is DATA: the 13761 x 13761 P1-FEM Poisson matrix (95065 entries) and its right-hand
side, i.e. BASELINE.json configs[0], parsed by the rules of the reference's native reader
(src/AMG_file_read.cpp:39-72: header "nrow ncol nnz", then 0-based "row col val" triplets sorted by
row; rhs file: "n", then one value per line) and stored as CSR arrays.  Values are kept as the
doubles strtod gives for the file's decimal text, so every consumer sees bit-identical inputs.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c0_matrix.npz")


def main():
    mpath = os.path.join(REF, "matrix_poisson_P1_14401")
    rpath = os.path.join(REF, "matrix_poisson_P1rhs_14401")
    if not (os.path.exists(mpath) and os.path.exists(rpath)):
        sys.exit("reference input files not found (build container only)")
    with open(mpath) as f:
        n, m, nnz = (int(t) for t in f.readline().split())
        body = np.array(f.read().split())
    assert body.size == 3 * nnz, (body.size, nnz)
    rows = body[0::3].astype(np.int64)
    cols = body[1::3].astype(np.int32)
    vals = np.array([float(t) for t in body[2::3]], dtype=np.float64)
    assert np.all(np.diff(rows) >= 0), "entries must be sorted by row (readcoo's assumption)"
    rowptr = np.zeros(n + 1, dtype=np.int32)
    np.add.at(rowptr, rows + 1, 1)
    rowptr = np.cumsum(rowptr, dtype=np.int64).astype(np.int32)
    with open(rpath) as f:
        nb = int(f.readline().split()[0])
        b = np.array([float(t) for t in f.read().split()], dtype=np.float64)
    assert nb == n and b.size == n
    np.savez_compressed(OUT, nrow=np.int32(n), ncol=np.int32(m), rowptr=rowptr, colindex=cols, val=vals, b=b)
    print(f"wrote {OUT}: {n} x {m}, {nnz} entries, ||b|| = {np.linalg.norm(b):.6f}, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
