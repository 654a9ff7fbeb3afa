"""Synthetic CSR inputs for tests and bench (SURVEY.md §8d "Inputs").

All generators return ``(rowptr int32[n+1], col int32[nnz], val float64[nnz])`` with sorted
columns, 0-based, in the layout ``sp_matrix`` holds (reference include/AMG_matrix.hpp:6-32).
"""
from __future__ import annotations

import numpy as np


def _stencil_csr(dims, diag):
    """Laplacian stencil (diag, -1) on a grid dims=(nx,[ny,[nz]]), Dirichlet eliminated,
    lexicographic order (x fastest)."""
    dims = tuple(int(d) for d in dims)
    N = int(np.prod(dims))
    if N * (2 * len(dims) + 1) >= 2**31:
        raise ValueError("nnz does not fit int32 indices")
    idx = np.arange(N, dtype=np.int64)
    strides = [1]
    for d in dims[:-1]:
        strides.append(strides[-1] * d)
    coords = [(idx // s) % d for s, d in zip(strides, dims)]
    # candidate columns in ascending order: -s_k ... -s_0, 0, +s_0 ... +s_k
    offs, masks = [], []
    for k in reversed(range(len(dims))):
        offs.append(-strides[k])
        masks.append(coords[k] > 0)
    offs.append(0)
    masks.append(np.ones(N, dtype=bool))
    for k in range(len(dims)):
        offs.append(strides[k])
        masks.append(coords[k] < dims[k] - 1)
    mask = np.stack(masks, axis=1)  # N x (2d+1)
    cols = idx[:, None] + np.asarray(offs, dtype=np.int64)[None, :]
    vals = np.full((N, len(offs)), -1.0)
    vals[:, len(dims)] = float(diag)
    counts = mask.sum(axis=1)
    rowptr = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    col = cols[mask].astype(np.int32)
    val = vals[mask]
    return rowptr.astype(np.int32), col, val


def poisson2d(n: int):
    """5-pt Laplacian on an n x n grid, stencil (4,-1)  (BASELINE configs[1] at n=1000)."""
    return _stencil_csr((n, n), 4.0)


def poisson3d(n: int, ny: int | None = None, nz: int | None = None):
    """7-pt Laplacian on an n^3 grid, stencil (6,-1)  (BASELINE configs[2] at n=216)."""
    return _stencil_csr((n, ny or n, nz or n), 6.0)


def _morton_order(pts):
    """Z-order (Morton) rank of 2D points in the unit square, 16 bits per axis."""
    q = np.minimum((pts * 65536.0).astype(np.uint64), 65535)

    def spread(v):
        v = (v | (v << 8)) & 0x00FF00FF
        v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v

    return np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1), kind="stable")


def fem_unstructured(npts: int = 525825, seed: int = 20240607, dt: float = 1e-2, ordering: str = "morton"):
    """P1-FEM  M + dt*K  on a Delaunay triangulation of random points in the unit square:
    SPD, irregular 3..12+ nnz/row.  Stand-in for SuiteSparse parabolic_fem (configs[4]),
    which cannot be fetched offline.

    ordering="morton" numbers the nodes along a space-filling curve, the kind of locality a mesh
    generator's numbering has (neighbouring nodes get nearby indices); ordering="random" keeps the
    random point order: every x-gather of a row lands on an unrelated cache line -- a worst case
    no mesh file exhibits, kept as a gather stress."""
    from scipy.spatial import Delaunay
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    pts = rng.random((npts, 2))
    if ordering == "morton":
        pts = pts[_morton_order(pts)]
    elif ordering != "random":
        raise ValueError("ordering must be 'morton' or 'random'")
    tri = Delaunay(pts).simplices.astype(np.int64)  # T x 3
    p0, p1, p2 = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    e1, e2 = p1 - p0, p2 - p0
    area2 = e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]
    area = 0.5 * np.abs(area2)
    # gradients of barycentric basis functions
    g = np.empty((len(tri), 3, 2))
    g[:, 0, 0] = (p1[:, 1] - p2[:, 1]) / area2
    g[:, 0, 1] = (p2[:, 0] - p1[:, 0]) / area2
    g[:, 1, 0] = (p2[:, 1] - p0[:, 1]) / area2
    g[:, 1, 1] = (p0[:, 0] - p2[:, 0]) / area2
    g[:, 2, 0] = (p0[:, 1] - p1[:, 1]) / area2
    g[:, 2, 1] = (p1[:, 0] - p0[:, 0]) / area2
    rows, cols, vals = [], [], []
    for a in range(3):
        for b in range(3):
            k = area * (g[:, a, 0] * g[:, b, 0] + g[:, a, 1] * g[:, b, 1])
            m = area / 12.0 * (2.0 if a == b else 1.0)
            rows.append(tri[:, a])
            cols.append(tri[:, b])
            vals.append(m + dt * k)
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(npts, npts)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def random_spd(n: int, nnz_per_row: int = 7, seed: int = 0):
    """Random sparse strictly diagonally dominant symmetric matrix with ragged rows
    (some rows hold only the diagonal) -- kernel edge-case input."""
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    m = n * max(nnz_per_row - 1, 0) // 2
    r = rng.integers(0, n, size=m)
    c = rng.integers(0, n, size=m)
    keep = r != c
    r, c = r[keep], c[keep]
    v = -rng.random(len(r))
    B = sp.coo_matrix((v, (r, c)), shape=(n, n)).tocsr()
    B = B + B.T
    d = -np.asarray(B.sum(axis=1)).ravel() + 1.0 + rng.random(n)
    A = (B + sp.diags(d)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


def write_coo(path_matrix: str, path_rhs: str, rowptr, col, val, b):
    """Native text format read by ``readcoo`` (reference src/AMG_file_read.cpp:39-72)."""
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    with open(path_matrix, "w") as f:
        f.write(f"{n} {n} {len(col)}\n")
        for r, c, v in zip(rows, col, val):
            f.write(f"{r} {c} {float(v):.17g}\n")
    with open(path_rhs, "w") as f:
        f.write(f"{n}\n")
        for v in b:
            f.write(f"{float(v):.17g}\n")


def read_matrix_market(path: str):
    """A real MatrixMarket reader (1-based indices, `symmetric` expanded, comments skipped) next
    to the reference's native readers (src/AMG_file_read.cpp:39-185, which accept neither);
    e.g. SuiteSparse parabolic_fem.mtx when it is available.  Returns sorted CSR arrays."""
    import scipy.io
    import scipy.sparse as sp

    A = sp.csr_matrix(scipy.io.mmread(path))
    A.sum_duplicates()
    A.sort_indices()
    if A.nnz >= 2**31:
        raise ValueError("nnz does not fit int32 indices")
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
