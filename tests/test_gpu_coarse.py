"""Device coarse direct solver for LARGE coarsest levels (csrc/coarse.cpp, coarse_kernels.hip) and the
reference's own level policy on the device.  GPU box only.

The reference stops coarsening after level1 = 6 levels and hands whatever is left to PARDISO
(/root/reference/src/AMG_phases.cpp:51,77,89; src/AMG_coarse_level_solver.cpp:64-76): 31 250 rows at
100^3.  The device solves such a level with a nested-dissection multifrontal factorisation (default since
round 3: csrc/nd_plan.cpp, nd_solver.cpp, nd_kernels.hip) or with round 2's block-tridiagonal
factorisation of the RCM-ordered operator (form "bt": explicit inverses of the Schur-complement diagonal
blocks), both factored and applied on the GPU.  Checked here: both solvers against scipy's sparse LU,
the fall-back when a coarsest level is refused, and the only >= 1 M-row vector captured
from the reference (SURVEY Appendix A.2, 100^3: 6 levels, 21 PCG iterations, 65 AMG cycles) on the
HIP path to the SURVEY §8d tolerance.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import oracle
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems
from conftest import hist_tolerance

pytestmark = pytest.mark.gpu

QUIET = dict(print_setup=0, print_solve=0)
ONE_LEVEL = dict(max_levels=1, coarse_limit=1 << 30, limit_upper=1 << 30)  # the whole matrix is the "coarsest level"


def _nonsym3d(m):
    """7-pt grid operator with direction-dependent coefficients (nonsymmetric: PARDISO mtype 11 is general)."""
    rp, ci, v = problems.poisson3d(m)
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    v = v.copy()
    v[ci > rows] *= 1.07
    d = ci == rows
    v[d] += 0.01 * (rows[d] % 5)
    return rp, ci, v


def _nonsym2d(m):
    rp, ci, v = problems.poisson2d(m)
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    v = v.copy()
    v[ci > rows] *= 0.9   # (weaker, not stronger, upper couplings: keeps the operator diagonally dominant)
    d = ci == rows
    v[d] += 0.02 * (rows[d] % 7)
    return rp, ci, v


@pytest.mark.parametrize("name,gen,interface", [
    ("p3d_24", lambda: problems.poisson3d(24), True),          # 13 824 rows, bandwidth ~ 24^2: too wide for the interface form
    ("p2d_150", lambda: problems.poisson2d(150), True),        # 22 500 rows, narrow band -> wide blocks, interface form of the solve
    ("p2d_150_plain_chain", lambda: problems.poisson2d(150), False),  # the same through the plain chain of B x B steps
    ("p2d_100_chain_unrolled", lambda: problems.poisson2d(100), True),        # window 128: each chain pass is one triangular product
    ("p2d_100_chain_step_by_step", lambda: problems.poisson2d(100), 2),       # the same, one launch per chain step
    ("nonsym2d_60_window64_short_last_block", lambda: _nonsym2d(60), True),   # window 64, last block (16 rows) shorter than a window
    ("nonsym2d_170", lambda: _nonsym2d(170), True),
    ("nonsym2d_170_chain_step_by_step", lambda: _nonsym2d(170), 2),            # 28 900 rows: interface form, pivoting, last block (100 rows) shorter than a window
    ("nonsym3d_22", lambda: _nonsym3d(22), True),              # pivoting inside the diagonal blocks
    ("p3d_ragged_last_block", lambda: problems.poisson3d(21), True),  # 9261 rows: short last block
])
def test_block_tridiagonal_solver_vs_sparse_lu(name, gen, interface):
    rp, ci, v = gen()
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).set_coarse_form("bt").set_coarse_interface(interface).setup(sa.default_params(**QUIET, **ONE_LEVEL, dense_limit=2000))
    try:
        info = A.coarse_info()
        assert A.nlevels == 1 and info["rows"] == n and not info["dense"]
        assert info["nblocks"] >= 2 and info["block"] % 64 == 0 and info["block"] >= info["bandwidth"]
        narrow = 2 * ((info["bandwidth"] + 63) // 64 * 64) <= info["block"]
        assert (info["window"] > 0) == bool(narrow and interface), info
        if "p2d_100" in name:
            assert info["window"] == 128
        if "window64" in name:
            assert info["window"] == 64
        if "2d" in name:
            assert narrow
        S = sp.csr_matrix((v, ci, rp), shape=(n, n))
        lu = spla.splu(S.tocsc())
        rng = np.random.default_rng(1)
        for b in (np.ones(n), rng.standard_normal(n)):
            x = A.op_coarse(b)
            xr = lu.solve(b)
            assert np.linalg.norm(x - xr) <= 1e-11 * np.linalg.norm(xr), name
            assert np.linalg.norm(b - S @ x) <= 1e-11 * np.linalg.norm(b)
        # deterministic: no atomics anywhere in the factorisation or the solve
        assert np.array_equal(A.op_coarse(np.ones(n)), A.op_coarse(np.ones(n)))
    finally:
        A.close()


def _fem(npts):
    return problems.fem_unstructured(npts)


def _squared(m):
    """A^2 of the 7-pt operator: 25 entries per row, two-cell-wide couplings (what classical coarsening leaves behind): thicker separators."""
    rp, ci, v = problems.poisson3d(m)
    A = sp.csr_matrix((v, ci, rp))
    B = (A @ A).tocsr()
    B.sort_indices()
    return B.indptr.astype(np.int32), B.indices.astype(np.int32), B.data.astype(np.float64)


@pytest.mark.parametrize("name,gen,leaf,merge,top", [
    ("p3d_24", lambda: problems.poisson3d(24), 0, -1, -1),            # 13 824 rows, defaults (leaf 64, merged separators up to 384 rows, 2048 at the root)
    ("p3d_24_plain_bisection", lambda: problems.poisson3d(24), 16, 0, 0),   # small leaves, no merging: a deep tree
    ("p3d_24_big_blocks", lambda: problems.poisson3d(24), 200, 600, 600),      # few levels, wide pivot blocks
    ("p2d_150", lambda: problems.poisson2d(150), 0, -1, -1),
    ("p2d_150_no_top_merge", lambda: problems.poisson2d(150), 0, -1, 0),
    ("nonsym2d_170", lambda: _nonsym2d(170), 0, -1, -1),               # 28 900 rows, nonsymmetric
    ("nonsym3d_22", lambda: _nonsym3d(22), 0, -1, -1),                 # not diagonally dominant: pivoting inside the pivot blocks
    ("p3d_ragged", lambda: problems.poisson3d(21), 32, 100, 300),
    ("fem_20000", lambda: _fem(20000), 0, -1, -1),                     # unstructured P1-FEM mesh: irregular separators, many children per node
    ("p3d_22_squared", lambda: _squared(22), 0, -1, -1),              # 10 648 rows, 25 entries per row: two-cell-wide separators
    ("p3d_40_top_separator_above_1024_rows", lambda: problems.poisson3d(40), 0, -1, -1),  # 64 000 rows: the top pivot block takes the whole-chip inversion
])
def test_nested_dissection_solver_vs_sparse_lu(name, gen, leaf, merge, top):
    rp, ci, v = gen()
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).set_coarse_form("nd", leaf, merge, top).setup(sa.default_params(**QUIET, **ONE_LEVEL, dense_limit=2000))
    try:
        info = A.coarse_info()
        assert A.nlevels == 1 and info["rows"] == n and not info["dense"] and info["form"] == "nested_dissection"
        assert info["nd_levels"] >= 2 and info["nd_launches_per_solve"] == 2 * info["nd_levels"] - 1
        assert info["nd_leaf"] == (leaf or 64)
        if "above_1024" in name:
            assert info["nd_max_pivot"] > 1024
        if "plain_bisection" in name:
            assert info["nd_levels"] > 8
        S = sp.csr_matrix((v, ci, rp), shape=(n, n))
        lu = spla.splu(S.tocsc())
        rng = np.random.default_rng(1)
        for b in (np.ones(n), rng.standard_normal(n)):
            x = A.op_coarse(b)
            xr = lu.solve(b)
            tol = 1e-10 if "squared" in name else 1e-11   # (the squared operator's condition number is the square: both solvers lose the digits)
            assert np.linalg.norm(x - xr) <= tol * np.linalg.norm(xr), name
            # residual: within two orders of the sparse LU's own (explicit pivot-block inverses are forward-, not backward-stable; the FEM
            # operator's solution is 1e4 times its right-hand side, so eps ||A|| ||x|| counts: measured 13x the LU's 1.1e-9 there)
            assert np.linalg.norm(b - S @ x) <= max(1e-11 * np.linalg.norm(b), 100.0 * np.linalg.norm(b - S @ xr))
        # deterministic: no atomics anywhere in the factorisation or the solve
        assert np.array_equal(A.op_coarse(np.ones(n)), A.op_coarse(np.ones(n)))
    finally:
        A.close()


def test_nested_dissection_and_block_tridiagonal_agree_in_a_solve():
    """Same hierarchy (the reference's 6 levels on 64^3: 8192-row coarsest level), coarsest level through either device factorisation."""
    rp, ci, v = problems.poisson3d(64)
    b = np.ones(len(rp) - 1)
    hs, xs = [], []
    for form in ("nd", "bt"):
        A = sa.sp_matrix_mg(rp, ci, v).set_coarse_form(form).setup(sa.default_params(**QUIET, dense_limit=4000))
        info = A.coarse_info()
        assert info["rows"] == 8192 and info["form"] == ("nested_dissection" if form == "nd" else "block_tridiagonal")
        x = np.zeros_like(b)
        h, rc = A.solve("pcg", b, x)
        assert rc == 0
        hs.append(h)
        xs.append(x)
        A.close()
    assert len(hs[0]) == len(hs[1])
    assert np.all(np.abs(hs[0] - hs[1]) <= hist_tolerance(hs[0]) * hs[0])
    assert np.linalg.norm(xs[0] - xs[1]) <= 1e-9 * np.linalg.norm(xs[0])


def test_refused_coarsest_level_extends_the_hierarchy():
    """with the default parameters a coarsest level the direct solver cannot take (here: a random sparsity pattern
    without separators, 12 500 rows after 6 levels) must not fail the setup: the hierarchy is extended by the reference's own
    coarsening rule, as it was before the reference's level policy became the default."""
    rp, ci, v = problems.random_spd(400000, 9, seed=5)
    b = np.ones(len(rp) - 1)
    for form in ("nd", "bt"):
        A = sa.sp_matrix_mg(rp, ci, v).set_coarse_form(form).setup(sa.default_params(**QUIET, dense_limit=4000))
        info = A.coarse_info()
        assert A.nlevels > 6 and info["dense"] and info["extended"], (A.nlevels, info)
        x = np.zeros_like(b)
        h, rc = A.solve("pcg", b, x)
        S = sp.csr_matrix((v, ci, rp))
        assert rc == 0 and np.linalg.norm(b - S @ x) <= 1.0001e-8
        A.close()


def test_dense_and_block_form_agree():
    """Same hierarchy, coarsest level once through the dense inverse, once through the device factors."""
    rp, ci, v = problems.poisson3d(40)
    b = np.ones(len(rp) - 1)
    hs, xs = [], []
    for dl in (8192, 1000):
        A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, dense_limit=dl))
        assert A.coarse_info()["dense"] == (dl == 8192) and A.level_info(A.nlevels - 1)["nrow"] == 4000
        x = np.zeros_like(b)
        h, rc = A.solve("pcg", b, x)
        assert rc == 0
        hs.append(h)
        xs.append(x)
        A.close()
    assert len(hs[0]) == len(hs[1])
    assert np.all(np.abs(hs[0] - hs[1]) <= hist_tolerance(hs[0]) * hs[0])
    assert np.linalg.norm(xs[0] - xs[1]) <= 1e-9 * np.linalg.norm(xs[0])


def test_too_wide_operator_is_refused():
    """A random sparsity pattern has neither band structure nor separators: when the caller's parameters leave no room to extend
    the hierarchy (one level forced) the setup must say so instead of building something huge."""
    rp, ci, v = problems.random_spd(20000, 9, seed=5)
    for form, msg in (("bt", "too wide"), ("nd", "no usable separators")):
        A = sa.sp_matrix_mg(rp, ci, v).set_coarse_form(form)
        with pytest.raises(sa.SparshError) as e:
            A.setup(sa.default_params(**QUIET, **ONE_LEVEL))
        assert msg in str(e.value), str(e.value)
        A.close()


def _hist_ok(h, ref):
    h, ref = np.asarray(h), np.asarray(ref)
    assert len(h) == len(ref), (len(h), len(ref))
    err = np.abs(h - ref) / ref
    assert np.all(err <= hist_tolerance(ref)), f"max rel err {err.max():.3e} at {err.argmax()}"


@pytest.fixture(scope="module")
def p100():
    rp, ci, v = problems.poisson3d(100)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))  # defaults: max_levels = 6 honoured
    yield A
    A.close()


def test_reference_level_policy_100cubed(p100, golden):
    A = p100
    g = golden["poisson3d_100"]["hem"]
    assert [A.level_info(l)["nrow"] for l in range(A.nlevels)] == g["levels_nrow"]       # 6 levels, 31 250-row coarsest
    assert [A.level_info(l)["nnz"] for l in range(A.nlevels)] == g["levels_nnz_stored"]
    info = A.coarse_info()
    assert info["rows"] == 31250 and not info["dense"] and not info["extended"] and info["form"] == "nested_dissection"


def test_golden_100cubed_pcg_on_device(p100, golden):
    """Appendix A.2: Solver_PCG_1 on 7-pt 100^3 -- 21 iterations, residuals of the reference's own CPU run."""
    A = p100
    g = golden["poisson3d_100"]["hem"]["pcg"]
    b = np.ones(A.nrow)
    x = np.zeros(A.nrow)
    h, rc = A.solve("pcg", b, x)
    assert rc == 0 and len(h) == g["iterations"] == 21
    _hist_ok(h, g["hist"])
    assert abs(h[0] - g["hist"][0]) <= 1e-10 * g["hist"][0]
    assert abs(np.linalg.norm(x) - g["xnorm"]) <= 1e-9 * g["xnorm"]
    assert abs(x[0] - g["x0"]) <= 1e-8


def test_golden_100cubed_amg_on_device(p100, golden):
    """Appendix A.2: AMG_Solver_CPU_baseline on 100^3 -- 65 V(7,7) cycles; last two residuals captured."""
    A = p100
    g = golden["poisson3d_100"]["hem"]["amg"]
    x = np.zeros(A.nrow)
    h, rc = A.solve("amg", np.ones(A.nrow), x)
    assert rc == 0 and len(h) == g["cycles"] == 65
    assert np.allclose(h[-2:], g["hist_tail"], rtol=1e-3)


def test_fp32_preconditioner_with_block_coarse_solver():
    """The opt-in float V-cycle keeps the block factors in fp64 and converts b_L / x_L around the solve."""
    rp, ci, v = problems.poisson3d(40)
    b = np.ones(len(rp) - 1)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, dense_limit=1000, precond_fp32=1))
    x = np.zeros_like(b)
    h, rc = A.solve("pcg", b, x)
    S = sp.csr_matrix((v, ci, rp))
    assert rc == 0 and np.linalg.norm(b - S @ x) <= 1.0001e-8
    A.close()
