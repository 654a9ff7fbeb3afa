"""Pin the CPU oracle (oracle/amg_oracle.c) against the reference's own outputs
(SURVEY.md Appendix A -> tests/golden/appendix_a.json).  CPU only."""
import numpy as np
import pytest

import oracle
from sparsh_amg_amd import problems
from conftest import C0_MATRIX, C0_RHS, hist_tolerance

# oracle-vs-MKL tolerance: summation orders differ (MKL's are unknown), so allow 100x the
# reference's own run-to-run noise (Appendix A.4) and never tighter than 1e-11.
def _check_hist(h, ref, scale=1.0):
    h = np.asarray(h)
    ref = np.asarray(ref)
    assert len(h) == len(ref)
    tol = hist_tolerance(ref) * scale
    err = np.abs(h - ref) / ref
    assert np.all(err <= tol), f"max rel err {err.max():.3e} at {err.argmax()}"


@pytest.fixture(scope="module")
def c0(c0_arrays):
    rp, ci, v, b = c0_arrays
    return oracle.Csr(rp, ci, v), b


def test_c0_fixture_is_the_reference_file(have_c0, c0_arrays):
    """tests/golden/c0_matrix.npz holds exactly what the reference's reader makes of its bundled
    files (checked wherever /root/reference exists; the fixture alone travels to the GPU box)."""
    if not have_c0:
        pytest.skip("/root/reference exists in the build container only")
    A, b = oracle.readcoo(C0_MATRIX, C0_RHS)
    rp, ci, v = A.arrays()
    assert np.array_equal(rp, c0_arrays[0]) and np.array_equal(ci, c0_arrays[1])
    assert np.array_equal(v, c0_arrays[2]) and np.array_equal(b, c0_arrays[3])


def test_c0_readcoo_shape(c0):
    A, b = c0
    assert A.shape == (13761, 13761) and A.nnz == 95065
    assert abs(np.linalg.norm(b) - 10.8032) < 1e-3


def test_c0_hierarchy(c0, golden):
    A, _ = c0
    H = oracle.Hierarchy(A)
    g = golden["C0"]["hem"]
    assert [H.A(i).shape[0] for i in range(H.nlevels)] == g["levels_nrow"]
    assert [H.A(i).nnz for i in range(H.nlevels)] == g["levels_nnz_stored"]
    # aggregate sizes of P0
    _, col, _ = H.P(0).arrays()
    counts = np.bincount(col)
    assert (counts == 2).sum() == g["P0_pairs"] and (counts == 1).sum() == g["P0_singletons"]
    _, col, _ = H.P(1).arrays()
    counts = np.bincount(col)
    assert (counts == 2).sum() == g["P1_pairs"] and (counts == 1).sum() == g["P1_singletons"]


def test_c0_amg_history(c0, golden):
    A, b = c0
    x, h = oracle.solve("amg", A, b)
    g = golden["C0"]["hem"]["amg"]
    assert len(h) == g["cycles"]
    # first cycle to 1e-12, whole history to the per-iteration tolerance
    assert abs(h[0] - g["hist"][0]) <= 1e-12 * g["hist"][0]
    _check_hist(h, g["hist"])
    assert abs(np.linalg.norm(x) - g["xnorm"]) <= 1e-10 * g["xnorm"]
    assert abs(x[0] - g["x0"]) <= 1e-9 and abs(x[-1] - g["xlast"]) <= 1e-9


def test_c0_pcg_history(c0, golden):
    A, b = c0
    x, h = oracle.solve("pcg", A, b)
    g = golden["C0"]["hem"]["pcg"]
    assert len(h) == g["iterations"]
    assert abs(h[0] - g["hist"][0]) <= 1e-12 * g["hist"][0]
    _check_hist(h, g["hist"])
    assert abs(np.linalg.norm(x) - g["xnorm"]) <= 1e-10 * g["xnorm"]


def test_c0_pbicg_history(c0, golden):
    A, b = c0
    x, h = oracle.solve("pbicg", A, b)
    g = golden["C0"]["hem"]["pbicg"]
    assert len(h) == g["iterations"]
    _check_hist(h, g["hist"])
    assert abs(np.linalg.norm(x) - g["xnorm"]) <= 1e-10 * g["xnorm"]


def test_c0_cg_bicg_heads(c0, golden):
    A, b = c0
    _, h = oracle.solve("cg", A, b)
    g = golden["C0"]["hem"]["cg"]
    assert abs(len(h) - g["iterations"]) <= 2
    assert np.allclose(h[:4], g["hist_head"], rtol=1e-11)
    _, h = oracle.solve("bicg", A, b)
    g = golden["C0"]["hem"]["bicg"]
    # un-preconditioned BiCGStab decorrelates in rounding after O(100) iterations: pin the head only
    assert np.allclose(h[:3], g["hist_head"], rtol=1e-11)
    assert h[-1] <= 1e-8


def test_c0_beck(c0, golden):
    A, b = c0
    prm = oracle.params(coarsening=1)
    H = oracle.Hierarchy(A, prm)
    g = golden["C0"]["beck"]
    assert [H.A(i).shape[0] for i in range(H.nlevels)] == g["levels_nrow"]
    assert [H.A(i).nnz for i in range(H.nlevels)] == g["levels_nnz_stored"]
    x, h = H.solve(b)
    assert len(h) == g["amg"]["cycles"]
    assert np.allclose(h[:3], g["amg"]["hist_head"], rtol=1e-10)
    assert abs(h[-1] - g["amg"]["last"]) <= 1e-3 * g["amg"]["last"]
    assert abs(np.linalg.norm(x) - g["amg"]["xnorm"]) <= 1e-9 * g["amg"]["xnorm"]


@pytest.fixture(scope="module")
def p2d():
    rp, ci, v = problems.poisson2d(256)
    return oracle.Csr(rp, ci, v), np.ones(len(rp) - 1)


@pytest.fixture(scope="module")
def p3d():
    rp, ci, v = problems.poisson3d(40)
    return oracle.Csr(rp, ci, v), np.ones(len(rp) - 1)


def test_poisson2d_256(p2d, golden):
    A, b = p2d
    g = golden["poisson2d_256"]
    assert A.shape[0] == g["nrow"] and A.nnz == g["nnz"]
    H = oracle.Hierarchy(A)
    assert [H.A(i).shape[0] for i in range(H.nlevels)] == g["hem"]["levels_nrow"]
    assert [H.A(i).nnz for i in range(H.nlevels)] == g["hem"]["levels_nnz_stored"]
    x, h, _ = H.pcg(b)
    gp = g["hem"]["pcg"]
    assert len(h) == gp["iterations"]
    assert np.allclose(h[:5], gp["hist_head"], rtol=1e-10)
    assert np.allclose(h[-2:], gp["hist_tail"], rtol=1e-3)
    assert abs(np.linalg.norm(x) - gp["xnorm"]) <= 1e-10 * gp["xnorm"]
    assert abs(x[0] - gp["x0"]) <= 1e-9
    _, h = oracle.solve("pbicg", A, b)
    gb = g["hem"]["pbicg"]
    assert len(h) == gb["iterations"]
    # iteration 9 is the genuine non-monotone spike (0.604...): a near-breakdown step,
    # sensitive to rounding -> 1e-4 there, 1e-6 elsewhere in the head
    head = np.asarray(gb["hist_head"])
    err = np.abs(h[: len(head)] - head) / head
    assert np.all(err[:9] <= 1e-6) and err[9] <= 1e-3 and err[10] <= 1e-3


def test_poisson3d_40(p3d, golden):
    A, b = p3d
    g = golden["poisson3d_40"]
    H = oracle.Hierarchy(A)
    assert [H.A(i).shape[0] for i in range(H.nlevels)] == g["hem"]["levels_nrow"]
    assert [H.A(i).nnz for i in range(H.nlevels)] == g["hem"]["levels_nnz_stored"]
    x, h = H.solve(b)
    ga = g["hem"]["amg"]
    assert len(h) == ga["cycles"]
    assert np.allclose(h[:5], ga["hist_head"], rtol=1e-11)
    assert np.allclose(h[-2:], ga["hist_tail"], rtol=1e-3)
    assert abs(np.linalg.norm(x) - ga["xnorm"]) <= 1e-10 * ga["xnorm"]
    x, h, _ = H.pcg(b)
    _check_hist(h, g["hem"]["pcg"]["hist"])
    _, h = oracle.solve("pbicg", A, b)
    _check_hist(h, g["hem"]["pbicg"]["hist"])


def test_poisson3d_40_beck(p3d, golden):
    A, b = p3d
    g = golden["poisson3d_40"]["beck"]
    H = oracle.Hierarchy(A, oracle.params(coarsening=1))
    assert [H.A(i).shape[0] for i in range(H.nlevels)] == g["levels_nrow"]
    assert [H.A(i).nnz for i in range(H.nlevels)] == g["levels_nnz_stored"]
    _, h = H.solve(b)
    assert len(h) == g["amg"]["cycles"]
    assert abs(h[0] - g["amg"]["hist_head"][0]) <= 1e-10 * h[0]
    assert abs(h[-1] - g["amg"]["last"]) <= 1e-3 * h[-1]


@pytest.mark.slow
def test_poisson3d_100_level1_6(golden):
    """1 M rows with the reference's own level1 = 6: the coarsest level has 31 250 rows, solved by
    the oracle's banded LU as PARDISO does in the reference.  Whole PCG history vs Appendix A.2."""
    import os

    rp, ci, v = problems.poisson3d(100)
    g = golden["poisson3d_100"]
    assert len(rp) - 1 == g["nrow"] and rp[-1] == g["nnz"]
    A = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(A, oracle.params(threads=min(8, os.cpu_count() or 1)))
    assert [H.A(i).shape[0] for i in range(H.nlevels)] == g["hem"]["levels_nrow"]
    assert [H.A(i).nnz for i in range(H.nlevels)] == g["hem"]["levels_nnz_stored"]
    x, h, _ = H.pcg(np.ones(len(rp) - 1))
    gp = g["hem"]["pcg"]
    _check_hist(h, gp["hist"])
    assert abs(h[0] - gp["hist"][0]) <= 1e-12 * gp["hist"][0]
    assert abs(np.linalg.norm(x) - gp["xnorm"]) <= 1e-10 * gp["xnorm"]
    assert abs(x[0] - gp["x0"]) <= 1e-10


def test_operators_vs_scipy():
    rp, ci, v = problems.random_spd(3000, 9, seed=3)
    A = oracle.Csr(rp, ci, v)
    S = A.to_scipy()
    rng = np.random.default_rng(1)
    x = rng.standard_normal(3000)
    b = rng.standard_normal(3000)
    assert np.allclose(oracle.spmv(A, x), S @ x, rtol=1e-13, atol=1e-13)
    assert np.allclose(oracle.spmv_t(A, x), S.T @ x, rtol=1e-13, atol=1e-13)
    assert np.allclose(oracle.store_residual(A, b, x), b - S @ x, rtol=1e-13, atol=1e-13)
    assert abs(oracle.residual(A, b, x) - np.linalg.norm(S @ x - b)) < 1e-10
    d = S.diagonal()
    xr = x.copy()
    for _ in range(7):
        xr = xr + 0.66667 * (b - S @ xr) / d
    assert np.allclose(oracle.jacobi(A, b, x, 6, 0.66667), xr, rtol=1e-12, atol=1e-12)
    assert abs(oracle.dot(x, b) - x @ b) < 1e-10
    # Galerkin product vs scipy
    H = oracle.Hierarchy(A, oracle.params(limit_upper=1000))
    P = H.P(0).to_scipy()
    Ac = H.A(1).to_scipy()
    ref = (P.T @ S @ P).toarray()
    assert np.abs(Ac.toarray() - ref).max() < 1e-12
    # coarse direct solve
    L = H.nlevels - 1
    AL = H.A(L).to_scipy()
    bc = rng.standard_normal(AL.shape[0])
    xc = H.coarse_solve(bc)
    assert np.linalg.norm(AL @ xc - bc) <= 1e-12 * np.linalg.norm(bc)
