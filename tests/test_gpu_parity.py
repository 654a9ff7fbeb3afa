"""Parity of the HIP path (through the C ABI) against the CPU oracle.  GPU box only.

Tolerances: SpMV-type kernels add products in stored order with separately
rounded multiply and add (-ffp-contract=off on both sides) -> compared BITWISE; reductions
rtol 1e-12; residual histories 1e-6 relative while r_k >= 1e-6 r_0 and 1e-3 below; final
solution 1e-8 relative in the 2-norm.
"""
import numpy as np
import pytest

import oracle
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems
from conftest import hist_tolerance

pytestmark = pytest.mark.gpu

QUIET = dict(print_setup=0, print_solve=0)


def _mk(rp, ci, v, **kw):
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, **kw))
    O = oracle.Csr(rp, ci, v)
    return A, O


CASES = {
    "p2d": lambda: problems.poisson2d(150),
    "p3d": lambda: problems.poisson3d(30),
    "ragged": lambda: problems.random_spd(20000, 9, seed=11),
}


@pytest.fixture(scope="module", params=list(CASES))
def case(request):
    rp, ci, v = CASES[request.param]()
    # every operator also carries its 16-bit delta-coded column indices (only the (0, 4) family below reads them)
    A = sa.sp_matrix_mg(rp, ci, v).set_index_compression(2).setup(sa.default_params(**QUIET))
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O)
    return request.param, A, O, H


def test_kernels_bitwise(case):
    name, A, O, H = case
    rng = np.random.default_rng(7)
    assert A.nlevels == H.nlevels
    for l in range(A.nlevels):
        n = A.level_info(l)["nrow"]
        x = rng.standard_normal(n)
        b = rng.standard_normal(n)
        Ol = H.A(l)
        assert np.array_equal(A.op_spmv(l, x), oracle.spmv(Ol, x)), f"spmv level {l}"
        assert np.array_equal(A.op_residual(l, b, x), oracle.store_residual(Ol, b, x)), f"residual level {l}"
        for sweeps in (1, 2, 7):
            assert np.array_equal(A.op_jacobi(l, b, x, sweeps), oracle.jacobi(Ol, b, x, sweeps - 1)), f"jacobi x{sweeps} level {l}"
        # zero-guess shortcut == full sweeps from x = 0
        assert np.array_equal(A.op_jacobi(l, b, np.zeros(n), 7, x_is_zero=True), oracle.jacobi(Ol, b, np.zeros(n), 6))
        rn = A.op_resnorm(l, b, x)
        assert abs(rn - oracle.residual(Ol, b, x)) <= 1e-12 * rn
        if l + 1 < A.nlevels:
            nc = A.level_info(l + 1)["nrow"]
            xc = rng.standard_normal(nc)
            assert np.array_equal(A.op_restrict(l, x), oracle.transfer_residual(H.P(l), x)), f"restrict level {l}"
            assert np.array_equal(A.op_prolong(l, xc, x), oracle.transfer_solution(H.P(l), xc, x)), f"prolong level {l}"
    nL = A.level_info(A.nlevels - 1)["nrow"]
    bc = rng.standard_normal(nL)
    xg = A.op_coarse(bc)
    xo = H.coarse_solve(bc)
    assert np.linalg.norm(xg - xo) <= 1e-10 * np.linalg.norm(xo)


# (0, 2): CSR-stream with row-lane gathers; (0, 4): the same with 16-bit delta-coded column indices
KERNEL_CONFIGS = [(0, 0), (0, 1), (0, 2), (0, 4), (1, 0), (1, 1), (2, 0), (3, 0)]


@pytest.mark.parametrize("kind,vec", KERNEL_CONFIGS)
def test_kernel_families_bitwise(case, kind, vec):
    """Every SpMV-type kernel family (workgroup / wave CSR-stream, scalar / paired loads, sliced
    ELL) must reproduce the oracle bit for bit, including the fused reductions' consumers."""
    name, A, O, H = case
    rng = np.random.default_rng(17)
    try:
        for nt, remap in ((1, 1), (0, 0), (1, 16), (-1, -1)):
            A.set_kernel_config(kind=kind, vec=vec, nt=nt, remap=remap)
            if (kind, vec) == (0, 4):
                assert A.level_kernel(0) == "csr_rowlane16_kernel" and A.level_index16(0)[0] > 0
            for l in range(A.nlevels):
                n = A.level_info(l)["nrow"]
                x = rng.standard_normal(n)
                b = rng.standard_normal(n)
                Ol = H.A(l)
                assert np.array_equal(A.op_spmv(l, x), oracle.spmv(Ol, x)), f"spmv level {l}"
                assert np.array_equal(A.op_residual(l, b, x), oracle.store_residual(Ol, b, x)), f"residual level {l}"
                assert np.array_equal(A.op_jacobi(l, b, x, 3), oracle.jacobi(Ol, b, x, 2)), f"jacobi level {l}"
                rn = A.op_resnorm(l, b, x)
                assert abs(rn - oracle.residual(Ol, b, x)) <= 1e-12 * rn
                if l + 1 < A.nlevels:
                    nc = A.level_info(l + 1)["nrow"]
                    xc = rng.standard_normal(nc)
                    assert np.array_equal(A.op_restrict(l, x), oracle.transfer_residual(H.P(l), x))
                    assert np.array_equal(A.op_prolong(l, xc, x), oracle.transfer_solution(H.P(l), xc, x))
        # whole solve through this family
        nrow = A.nrow
        bb = np.ones(nrow)
        xx = np.zeros(nrow)
        h, rc = A.solve("pcg", bb, xx)
        xo, ho = oracle.solve("pcg", O, bb)
        _hist_ok(h, ho)
        assert np.linalg.norm(xx - xo) <= 1e-8 * np.linalg.norm(xo)
    finally:
        A.set_kernel_config()


def test_blas1(case):
    _, A, _, _ = case
    rng = np.random.default_rng(8)
    for n in (1, 63, 64, 65, 1000, 123457):
        x = rng.standard_normal(n)
        y = rng.standard_normal(n)
        d = A.op_dot(x, y)
        ref = float(np.dot(x.astype(np.longdouble), y.astype(np.longdouble)))
        scale = float(np.dot(np.abs(x), np.abs(y)))
        assert abs(d - ref) <= 1e-13 * scale
        assert abs(A.op_nrm2(x) - np.linalg.norm(x)) <= 1e-13 * np.linalg.norm(x)
        assert np.array_equal(A.op_axpby(0.3, x, -1.7, y), 0.3 * x + (-1.7) * y)


def test_reductions_reproducible(case):
    _, A, _, _ = case
    rng = np.random.default_rng(9)
    x = rng.standard_normal(200001)
    y = rng.standard_normal(200001)
    vals = {A.op_dot(x, y) for _ in range(5)}
    assert len(vals) == 1  # two-stage fixed-order reduction, no atomics


def _hist_ok(h, ho):
    assert len(h) == len(ho), (len(h), len(ho))
    tol = hist_tolerance(ho)
    err = np.abs(h - ho) / ho
    assert np.all(err <= tol), f"max rel err {err.max():.3e} at {err.argmax()}"


@pytest.mark.parametrize("method", ["amg", "pcg", "pbicg", "cg"])
def test_solver_history(case, method):
    name, A, O, _ = case
    n = A.nrow
    b = np.ones(n)
    x = np.zeros(n)
    h, rc = A.solve(method, b, x)
    xo, ho = oracle.solve(method, O, b)
    assert rc == 0
    if method == "cg":
        # hundreds of un-preconditioned iterations: rounding decorrelates the tail; pin the head
        k = min(20, len(h), len(ho))
        assert np.allclose(h[:k], ho[:k], rtol=1e-9)
        assert abs(len(h) - len(ho)) <= max(3, len(ho) // 50)
    else:
        _hist_ok(h, ho)
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)
    S = O.to_scipy()
    assert np.linalg.norm(b - S @ x) <= 1.0001e-8


def test_bicg_head(case):
    name, A, O, _ = case
    n = A.nrow
    b = np.ones(n)
    x = np.zeros(n)
    h, rc = A.solve("bicg", b, x)
    _, ho = oracle.solve("bicg", O, b)
    k = min(10, len(h), len(ho))
    assert np.allclose(h[:k], ho[:k], rtol=1e-8)
    assert np.linalg.norm(b - O.to_scipy() @ x) <= 1.0001e-8


def test_nonzero_initial_guess(case):
    _, A, O, _ = case
    n = A.nrow
    rng = np.random.default_rng(3)
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    for method in ("amg", "pcg"):
        x = x0.copy()
        h, rc = A.solve(method, b, x)
        xo, ho = oracle.solve(method, O, b, x0=x0)
        _hist_ok(h, ho)
        assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)


def test_fixed_cycle_count(case):
    _, A, O, H = case
    n = A.nrow
    b = np.ones(n)
    x = np.zeros(n)
    h, _ = A.vcycle(b, x, iterations=3)
    xo, ho = H.solve(b, iterations=3)
    assert len(h) == 3 and np.allclose(h, ho, rtol=1e-10)
    assert np.linalg.norm(x - xo) <= 1e-11 * np.linalg.norm(xo)


def test_golden_histories(golden):
    """Device path vs the reference's own outputs."""
    rp, ci, v = problems.poisson3d(40)
    A, _ = _mk(rp, ci, v)
    g = golden["poisson3d_40"]["hem"]
    b = np.ones(A.nrow)
    x = np.zeros(A.nrow)
    h, _ = A.solve("pcg", b, x)
    _hist_ok(h, np.array(g["pcg"]["hist"]))
    x[:] = 0
    h, _ = A.solve("pbicg", b, x)
    _hist_ok(h, np.array(g["pbicg"]["hist"]))
    x[:] = 0
    h, _ = A.solve("amg", b, x)
    assert len(h) == g["amg"]["cycles"]
    assert np.allclose(h[:5], g["amg"]["hist_head"], rtol=1e-9)
    assert abs(np.linalg.norm(x) - g["amg"]["xnorm"]) <= 1e-9 * g["amg"]["xnorm"]
    rp, ci, v = problems.poisson2d(256)
    A, _ = _mk(rp, ci, v)
    g = golden["poisson2d_256"]["hem"]
    x = np.zeros(A.nrow)
    h, _ = A.solve("pcg", np.ones(A.nrow), x)
    assert len(h) == g["pcg"]["iterations"]
    assert np.allclose(h[:5], g["pcg"]["hist_head"], rtol=1e-9)
    assert abs(np.linalg.norm(x) - g["pcg"]["xnorm"]) <= 1e-9 * g["pcg"]["xnorm"]


def test_beck_golden(golden):
    """Device path with Beck coarsening vs the reference's own output."""
    rp, ci, v = problems.poisson3d(40)
    A, _ = _mk(rp, ci, v, coarsening=1)
    g = golden["poisson3d_40"]["beck"]
    assert [A.level_info(l)["nrow"] for l in range(A.nlevels)] == g["levels_nrow"]
    assert [A.level_info(l)["nnz"] for l in range(A.nlevels)] == g["levels_nnz_stored"]
    x = np.zeros(A.nrow)
    h, rc = A.solve("amg", np.ones(A.nrow), x)
    assert rc == 0 and len(h) == g["amg"]["cycles"]
    assert abs(h[0] - g["amg"]["hist_head"][0]) <= 1e-10 * h[0]
    assert abs(h[-1] - g["amg"]["last"]) <= 1e-3 * h[-1]


def test_beck_coarsening():
    rp, ci, v = problems.poisson3d(30)
    A, O = _mk(rp, ci, v, coarsening=1)
    b = np.ones(A.nrow)
    x = np.zeros(A.nrow)
    h, rc = A.solve("amg", b, x)
    H = oracle.Hierarchy(O, oracle.params(coarsening=1))
    xo, ho = H.solve(b)
    _hist_ok(h, ho)
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)


def test_index16_blocks_and_fallbacks():
    """16-bit delta-coded column indices (SURVEY 8f-4 'compressed indices'): blocks whose deltas fit use them, blocks with a gap
    of 65536 or more, an unsorted row or one over-long row keep col[]; general P / R (Beck) go through the same kernel.
    Everything bitwise against the oracle."""
    import scipy.sparse as sp

    rng = np.random.default_rng(21)
    n = 150000
    M = sp.diags([np.full(n - 1, -1.0), np.full(n, 4.0), np.full(n - 1, -1.0)], [-1, 0, 1], format="lil")
    M[5, 70000] = 0.5        # gap of 69994 inside row 5 -> its block keeps 32-bit indices
    M[70000, 5] = 0.5
    M[3000, 3000 + 65535] = 0.25   # the largest gap that still fits (65534 after the +1 neighbour)
    M[90000, :4000] = rng.standard_normal(4000)  # one row longer than the LDS buffer
    M = M.tocsr()
    M.sort_indices()
    A = sa.sp_matrix_mg(M.indptr, M.indices, M.data).set_index_compression(2).setup(sa.default_params(**QUIET, limit_upper=200000))
    assert A.nlevels == 1
    b16, nb = A.level_index16(0)
    assert 0 < b16 < nb and nb - b16 <= 8, (b16, nb)   # a handful of blocks fall back
    Mo = oracle.Csr(M.indptr, M.indices, M.data)
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    A.set_kernel_config(kind=0, vec=4)
    assert A.level_kernel(0) == "csr_rowlane16_kernel"
    mask = np.ones(n, bool)
    mask[90000] = False  # the long row is tree-summed (tolerance), everything else in stored order (bitwise)
    yo = oracle.spmv(Mo, x)
    y = A.op_spmv(0, x)
    assert np.array_equal(y[mask], yo[mask])
    assert abs(y[90000] - yo[90000]) <= 1e-12 * (np.abs(M[90000].toarray()).ravel() @ np.abs(x))
    ro = oracle.store_residual(Mo, b, x)
    assert np.array_equal(A.op_residual(0, b, x)[mask], ro[mask])
    jo = oracle.jacobi(Mo, b, x, 0)
    assert np.array_equal(A.op_jacobi(0, b, x, 1)[mask], jo[mask])
    A.close()
    # mode 0 builds nothing and vec = 4 then runs the 32-bit row-lane kernel; mode 1 leaves small operators alone
    rp, ci, v = problems.poisson3d(30)
    for mode in (0, 1):
        B = sa.sp_matrix_mg(rp, ci, v).set_index_compression(mode).setup(sa.default_params(**QUIET))
        B.set_kernel_config(kind=0, vec=4)
        assert B.level_index16(0)[0] == 0 and B.level_kernel(0) == "csr_rowlane_kernel"
        B.close()
    # Beck: multi-entry P and R through the compressed kernel, whole solve equal to the default family's
    C1 = sa.sp_matrix_mg(rp, ci, v).set_index_compression(2).setup(sa.default_params(**QUIET, coarsening=1))
    H = oracle.Hierarchy(oracle.Csr(rp, ci, v), oracle.params(coarsening=1))
    C1.set_kernel_config(kind=0, vec=4)
    for l in range(C1.nlevels - 1):
        nf = C1.level_info(l)["nrow"]
        nc = C1.level_info(l + 1)["nrow"]
        xf = rng.standard_normal(nf)
        xc = rng.standard_normal(nc)
        assert np.array_equal(C1.op_restrict(l, xf), oracle.transfer_residual(H.P(l), xf)), l
        assert np.array_equal(C1.op_prolong(l, xc, xf), oracle.transfer_solution(H.P(l), xc, xf)), l
        assert np.array_equal(C1.op_spmv(l, xf), oracle.spmv(H.A(l), xf)), l
    bb = np.ones(C1.nrow)
    x1 = np.zeros(C1.nrow)
    h1, _ = C1.solve("pcg", bb, x1)
    C1.set_kernel_config()
    x0 = np.zeros(C1.nrow)
    h0, _ = C1.solve("pcg", bb, x0)
    assert np.array_equal(h0, h1) and np.array_equal(x0, x1)
    C1.close()


def test_edge_cases():
    # single level (n <= limit_upper): the V-cycle is the direct solve
    rp, ci, v = problems.poisson2d(30)
    A, O = _mk(rp, ci, v)
    assert A.nlevels == 1
    b = np.ones(A.nrow)
    x = np.zeros(A.nrow)
    h, rc = A.solve("pcg", b, x)
    assert rc == 0 and len(h) <= 2
    assert np.linalg.norm(b - O.to_scipy() @ x) <= 1e-8
    # rows holding only the diagonal and one very long row (own-workgroup path of the CSR kernel)
    import scipy.sparse as sp

    n = 6000
    rng = np.random.default_rng(5)
    M = sp.random(n, n, density=0.001, random_state=6, format="lil")
    M[17, :] = rng.standard_normal(n)  # 6000 entries > LDS product buffer
    M[512, :] = rng.standard_normal(n)  # a second one at a multiple of the workgroup size (diagonal pick-up by r0, not r0 + tid)
    M[100, :] = 0
    M[101, :] = 0
    M = (M.tocsr() + sp.diags(np.full(n, 50.0))).tocsr()
    M.sort_indices()
    assert M.indptr[101] - M.indptr[100] == 1
    A2 = sa.sp_matrix_mg(M.indptr, M.indices, M.data).setup(sa.default_params(**QUIET, limit_upper=10000))
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    Mo = oracle.Csr(M.indptr, M.indices, M.data)
    yo = oracle.spmv(Mo, x)
    jo = oracle.jacobi(Mo, b, x, 2)
    mask = np.ones(n, bool)
    mask[[17, 512]] = False
    seen = set()
    # default family, then every CSR-stream variant by name: the long row takes the strided partial-sum branch of each
    for kind, vec in ((3, 3), (0, 0), (0, 1), (0, 2), (1, 0)):
        A2.set_kernel_config(kind=kind, vec=vec)
        seen.add(A2.level_kernel(0))
        y = A2.op_spmv(0, x)
        assert np.array_equal(y[mask], yo[mask]), (kind, vec)
        for r in (17, 512):
            assert abs(y[r] - yo[r]) <= 1e-12 * (np.abs(M[r].toarray()).ravel() @ np.abs(x)), (kind, vec, r)
        assert np.allclose(A2.op_jacobi(0, b, x, 3), jo, rtol=1e-12, atol=1e-12), (kind, vec)
    assert any("rowlane" in k for k in seen), seen


def test_max_iter_cap_reports_noconv():
    rp, ci, v = problems.poisson3d(30)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, max_iter=3))
    x = np.zeros(A.nrow)
    h, rc = A.solve("pcg", np.ones(A.nrow), x)
    assert rc == sa.SPARSH_ENOCONV and len(h) == 3


def test_graph_replay_matches_eager():
    """use_graph: one captured PCG iteration replayed -- same launches, same buffers, same bits."""
    rp, ci, v = problems.poisson3d(30)
    n = len(rp) - 1
    b = np.ones(n)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    x0 = np.zeros(n)
    h0, _ = A.solve("pcg", b, x0)
    G = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, use_graph=1))
    for _ in range(2):  # second solve reuses or re-captures the graph
        x1 = np.zeros(n)
        h1, rc = G.solve("pcg", b, x1)
        assert rc == 0 and np.array_equal(h0, h1) and np.array_equal(x0, x1)
    # stepwise interface, device vectors, several step batches on one captured graph
    bd, xd = G.dev_alloc(8 * n), G.dev_alloc(8 * n)
    G.h2d(bd, b)
    G.h2d(xd, np.zeros(n))
    G.krylov_init_dev("pcg", bd, xd)
    G.krylov_step_dev(3)
    G.krylov_step_dev(4)
    hs = G.krylov_history()
    assert np.array_equal(hs, h0[:7])


def test_graph_replay_with_device_factored_coarsest_level():
    """use_graph with a coarsest level above dense_limit: the nested-dissection solve (2 * tree levels - 1 launches, no host
    synchronisation) is captured with the rest of the iteration -- same launches, same buffers, same bits; and changing a kernel
    setting afterwards drops the captured graph instead of replaying stale launches."""
    rp, ci, v = problems.poisson3d(64)   # 262 144 rows -> 6 levels, 8192-row coarsest level
    b = np.ones(len(rp) - 1)
    res = []
    for graph in (0, 1):
        A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, dense_limit=4000, use_graph=graph))
        assert A.coarse_info()["form"] == "nested_dissection"
        x = np.zeros_like(b)
        h, rc = A.solve("pcg", b, x)
        assert rc == 0
        res.append((h, x))
        if graph:
            A.set_kernel_config(0, 3, -1, -1)   # CSR-stream family from now on: the graph of the table-path launches must not be replayed
            x2 = np.zeros_like(b)
            h2, rc2 = A.solve("pcg", b, x2)
            assert rc2 == 0 and A.level_kernel(0).startswith("csr_")
            assert np.array_equal(h2, h) and np.array_equal(x2, x)   # every family is bitwise equal: only a stale replay could differ (or crash)
        A.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def _mixed_coefficients(n):
    """5-pt stencil with a variable diagonal and a partly variable east coupling: slices hold constant
    and non-constant diagonals side by side."""
    rp, ci, v = problems.poisson2d(n)
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))
    rng = np.random.default_rng(5)
    v = v.copy()
    diag = ci == rows
    v[diag] += rng.random(int(diag.sum()))
    east = (ci == rows + 1) & (rows < (len(rp) - 1) // 2)
    v[east] = -0.5 - 0.25 * rng.random(int(east.sum()))
    return rp, ci, v


def test_constant_slot_folding_bitwise():
    """Sliced-diagonal layout: slots whose entries share one value keep it as a scalar and own no value
    block.  Folded, unfolded and oracle results must be bit-identical; the layout report must show the
    folding (all slots on constant-coefficient Poisson, only some on the mixed operator)."""
    rng = np.random.default_rng(23)
    rp3, ci3, v3 = problems.poisson3d(48)
    v3p = v3.copy()   # a few rows off the level-wide stencil: their slices leave the table / record paths
    rows3 = np.repeat(np.arange(len(rp3) - 1), np.diff(rp3))
    dpos = np.flatnonzero(ci3 == rows3)
    v3p[dpos[[5, 7000, 7001, 64000, 110000]]] += np.array([0.5, 0.25, 1.0, 2.0, 0.125])
    v3p[dpos[30000] + 1] = -0.75
    for name, (rp, ci, v) in (("p3d", (rp3, ci3, v3)), ("p3d_perturbed", (rp3, ci3, v3p)), ("mixed", _mixed_coefficients(300))):
        n = len(rp) - 1
        A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, max_levels=2))
        B = sa.sp_matrix_mg(rp, ci, v).set_const_slots(False).setup(sa.default_params(**QUIET, max_levels=2))  # per handle
        assert A.level_format(0)[0] == 3 and B.level_format(0)[0] == 3
        sl, vb, _ = A.level_layout(0)
        slb, vbb, _ = B.level_layout(0)
        assert sl == slb and vbb == slb            # unfolded: one block per slot
        if name == "p3d":
            assert vb == 0                         # every diagonal of every slice is constant
        elif name == "p3d_perturbed":
            assert 0 < vb <= 8                     # only the touched slices keep value blocks
        else:
            assert 0 < vb < sl                     # diagonal slots (and the upper-half east slots) keep blocks
        O = oracle.Csr(rp, ci, v)
        x = rng.standard_normal(n)
        b = rng.standard_normal(n)
        for nt, remap in ((1, 16), (0, 1), (-1, -1)):
            A.set_kernel_config(kind=3, vec=0, nt=nt, remap=remap)
            B.set_kernel_config(kind=3, vec=0, nt=nt, remap=remap)
            try:
                ya, yb = A.op_spmv(0, x), B.op_spmv(0, x)
                assert np.array_equal(ya, yb) and np.array_equal(ya, oracle.spmv(O, x)), name
                ja, jb = A.op_jacobi(0, b, x, 3), B.op_jacobi(0, b, x, 3)
                assert np.array_equal(ja, jb) and np.array_equal(ja, oracle.jacobi(O, b, x, 2)), name
                assert np.array_equal(A.op_residual(0, b, x), oracle.store_residual(O, b, x)), name
                ra, rb = A.op_resnorm(0, b, x), B.op_resnorm(0, b, x)
                assert abs(ra - rb) <= 1e-13 * ra and abs(ra - oracle.residual(O, b, x)) <= 1e-12 * ra
            finally:
                A.set_kernel_config()
                B.set_kernel_config()
        A.close()
        B.close()


def _banded(n, offs, vals):
    import scipy.sparse as sp

    A = sp.diags([np.full(n - abs(o), float(c)) for o, c in zip(offs, vals)], offs, shape=(n, n), format="csr")
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


@pytest.mark.parametrize("name,offs,vals", [
    ("lap1d", [-1, 0, 1], [-1, 2.5, -1]),                          # near-gather variant <3,1>
    ("skew5", [-40, -3, 0, 3, 40], [-1, -0.5, 4, -0.5, -1]),         # table without the -1/0/+1 triple
    ("band7", [-3, -2, -1, 0, 1, 2, 3], [-0.25, -0.5, -1, 5, -1, -0.5, -0.25]),  # <7,3> with near far-bands
])
def test_table_kernel_variants_bitwise(name, offs, vals):
    """Every code path of the stencil-table kernel on synthetic constant-band operators whose first
    and last slices are ragged (n is not a multiple of 64): SpMV, residual, fused Jacobi sweeps and
    the fused residual norm against the oracle, bit for bit."""
    n = 64 * 37 + 29
    rp, ci, v = _banded(n, offs, vals)
    O = oracle.Csr(rp, ci, v)
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, max_levels=2, coarse_limit=1 << 20, limit_upper=1 << 20))
    try:
        assert A.level_kernel(0) == "sdia_tab_kernel", A.level_kernel(0)
        assert np.array_equal(A.op_spmv(0, x), oracle.spmv(O, x))
        assert np.array_equal(A.op_residual(0, b, x), oracle.store_residual(O, b, x))
        rn = A.op_resnorm(0, b, x)
        assert abs(rn - oracle.residual(O, b, x)) <= 1e-12 * rn
        assert np.array_equal(A.op_jacobi(0, b, x, 3), oracle.jacobi(O, b, x, 2))
    finally:
        A.close()


def test_sliced_diagonal_paths_fuzz():
    """Random banded operators -- random offset sets (3..12 bands), some bands constant, some variable,
    some with holes, ragged first/last slices -- drive every path of the sliced-diagonal kernels
    (table, record, > 8 constant slots, mixed, plain, fallback inside a table level).  SpMV, residual
    and fused Jacobi sweeps must equal the oracle bit for bit on every one."""
    import scipy.sparse as sp

    rng = np.random.default_rng(20241004)
    kernels_seen = set()
    for case in range(36):
        n = int(rng.integers(300, 6000))
        nb = int(rng.integers(1, 6))
        offs = sorted(set(int(o) for o in rng.choice(np.arange(1, min(n // 3, 400)), size=nb, replace=False)))
        if case % 3 == 0:
            offs = sorted(set(offs + [1]))       # lexicographic-stencil shape: -1, 0, +1 adjacent
        mode = case % 4                          # 0 all constant, 1 variable diagonal, 2 one variable band, 3 holes + a few odd rows
        rows, cols, vals = [], [], []
        for o in offs:
            for sgn in (-1, 1):
                i = np.arange(max(0, -sgn * o), min(n, n - sgn * o))
                c = float(-rng.integers(1, 4)) / 4.0
                v = np.full(len(i), c)
                if mode == 2 and o == offs[-1] and sgn == 1:
                    v = -rng.random(len(i))
                keep = np.ones(len(i), bool)
                if mode == 3:
                    keep = rng.random(len(i)) > 0.1
                rows.append(i[keep]); cols.append(i[keep] + sgn * o); vals.append(v[keep])
        M = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
        d = np.asarray(abs(M).sum(axis=1)).ravel() + 1.0
        if mode == 0 or mode == 2:
            d[:] = d.max()                       # constant diagonal
        if mode == 3:
            odd = rng.choice(n, size=5, replace=False)
            d[odd] += rng.random(5)              # a few rows off the level-wide stencil
        A = (M + sp.diags(d)).tocsr()
        A.sort_indices()
        rp, ci, v = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
        O = oracle.Csr(rp, ci, v)
        H = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, max_levels=2, coarse_limit=1 << 20, limit_upper=1 << 20))
        try:
            kernels_seen.add(H.level_kernel(0))
            x = rng.standard_normal(n)
            b = rng.standard_normal(n)
            assert np.array_equal(H.op_spmv(0, x), oracle.spmv(O, x)), (case, offs, mode)
            assert np.array_equal(H.op_residual(0, b, x), oracle.store_residual(O, b, x)), (case, offs, mode)
            assert np.array_equal(H.op_jacobi(0, b, x, 2), oracle.jacobi(O, b, x, 1)), (case, offs, mode)
            rn = H.op_resnorm(0, b, x)
            assert abs(rn - oracle.residual(O, b, x)) <= 1e-12 * rn
        finally:
            H.close()
    assert {"sdia_tab_kernel", "sdia_kernel"} <= kernels_seen, kernels_seen


def _box3d(nx, ny, nz):
    """7-pt Laplacian on an nx x ny x nz box (x fastest): line offset nx, plane offset nx*ny."""
    import scipy.sparse as sp

    def lap(n):
        return sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1])

    A = (sp.kron(sp.identity(nz), sp.kron(sp.identity(ny), lap(nx))) + sp.kron(sp.identity(nz), sp.kron(lap(ny), sp.identity(nx)))
         + sp.kron(lap(nz), sp.identity(nx * ny))).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


@pytest.mark.parametrize("name,gen", [
    ("p3d_48", lambda: problems.poisson3d(48)),            # line 48, plane 2304
    ("box_70x40x37", lambda: _box3d(70, 40, 37)),           # line 70 (not a multiple of anything), ragged last tile
    ("p2d_300", lambda: problems.poisson2d(300)),          # 5-pt: every neighbour from LDS
    ("p2d_700", lambda: problems.poisson2d(700)),          # line 700: 2048-row tiles
])
def test_lds_tiled_table_kernel_bitwise(name, gen):
    """sdia_tile_kernel (x tiles staged in LDS; opt-in, see DESIGN.md: measured slower than the untiled
    table kernel) against the oracle and against the untiled table kernel, bit for bit, for every
    epilogue, on every level that uses it; fused reductions to 1e-12."""
    rp, ci, v = gen()
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).set_tile(True).setup(sa.default_params(**QUIET))
    B = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))  # default: untiled
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O)
    rng = np.random.default_rng(41)
    tiled_levels = [l for l in range(A.nlevels) if A.level_tile_rows(l) > 0]
    assert 0 in tiled_levels and A.level_kernel(0) == "sdia_tab_kernel", (name, tiled_levels)
    assert all(B.level_tile_rows(l) == 0 for l in range(B.nlevels))
    for l in tiled_levels:
        nl = A.level_info(l)["nrow"]
        x, b = rng.standard_normal(nl), rng.standard_normal(nl)
        Ol = H.A(l)
        ya = A.op_spmv(l, x)
        assert np.array_equal(ya, oracle.spmv(Ol, x)) and np.array_equal(ya, B.op_spmv(l, x)), (name, l)
        assert np.array_equal(A.op_residual(l, b, x), oracle.store_residual(Ol, b, x)), (name, l)
        for sweeps in (1, 3):
            assert np.array_equal(A.op_jacobi(l, b, x, sweeps), oracle.jacobi(Ol, b, x, sweeps - 1)), (name, l, sweeps)
        rn = A.op_resnorm(l, b, x)
        assert abs(rn - oracle.residual(Ol, b, x)) <= 1e-12 * rn
        if l + 1 < A.nlevels:
            xc = rng.standard_normal(A.level_info(l + 1)["nrow"])
            assert np.array_equal(A.op_prolong(l, xc, x), oracle.transfer_solution(H.P(l), xc, x))
    bb = np.ones(n)
    xa, xb = np.zeros(n), np.zeros(n)
    ha, rca = A.solve("pcg", bb, xa)
    hb, rcb = B.solve("pcg", bb, xb)
    assert rca == 0 and rcb == 0 and len(ha) == len(hb)
    assert np.allclose(ha, hb, rtol=1e-9)  # fused dot products are summed per tile instead of per 4 slices
    xo, ho = oracle.solve("pcg", O, bb)
    _hist_ok(ha, ho)
    A.close()
    B.close()


@pytest.mark.parametrize("nu", [1, 2, 3])
def test_few_sweeps_pcg_matches_oracle(nu):
    """nu = 1: the zero-guess sweep (written by the cg_update kernel inside PCG) is the whole pre-smoothing leg; nu = 2, 3: the
    alternation of the ping-pong buffers differs from nu = 7.  Histories against the oracle with the same nu."""
    rp, ci, v = problems.poisson3d(30)
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, sweeps=nu))
    O = oracle.Csr(rp, ci, v)
    b = np.ones(n)
    x = np.zeros(n)
    h, rc = A.solve("pcg", b, x)
    xo, ho = oracle.solve("pcg", O, b, prm=oracle.params(smooth_iter=nu - 1))  # the CPU path sweeps smooth_iter + 1 times
    _hist_ok(h, ho)
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)
    for mode in (0, 1):  # separate zero-guess launch / fused without the non-temporal streams: same bits
        A.set_fused_zero_sweep(mode)
        x2 = np.zeros(n)
        h2, _ = A.solve("pcg", b, x2)
        assert np.array_equal(h2, h) and np.array_equal(x2, x)
    A.close()


def _line1d(n):
    import scipy.sparse as sp

    T = sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")
    return T.indptr.astype(np.int32), T.indices.astype(np.int32), T.data


@pytest.mark.parametrize("name,gen", [
    ("p3d_30", lambda: problems.poisson3d(30)),
    ("p2d_150", lambda: problems.poisson2d(150)),
    ("line_40001", lambda: _line1d(40001)),  # odd row count: the last aggregate is a single row; levels 0 and 2 pair
    ("line_40000", lambda: _line1d(40000)),
    ("box_40_36_32", lambda: problems.poisson3d(40, 36, 32)),  # levels 1, 2: aggregates one grid line / one plane apart (box_resid_pair_kernel)
])
def test_paired_restriction_bitwise(name, gen):
    """Levels whose aggregates are the row pairs (2J, 2J+1) run store_residual + transfer_residual + the coarse level's
    zero-guess sweep as one launch (OP_RESID_PAIR).  Against the oracle's three separate steps, bit for bit, on every
    such level; and whole solves with the fusion on and off give the same histories and solutions, bit for bit."""
    rp, ci, v = gen()
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, max_iter=40))  # (the line problems need more: capped)
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O)
    rng = np.random.default_rng(77)
    paired = [l for l in range(A.nlevels - 1) if A.level_paired(l)]
    assert 0 in paired, (name, paired)
    assert not A.level_paired(A.nlevels - 2)  # the level above the coarsest keeps the plain restriction (no sweep there)
    if name == "box_40_36_32":
        kinds = [A.level_paired(l) for l in range(A.nlevels - 2)]
        assert kinds[0] == 1 and all(k in (2, 3) for k in kinds[1:3]) and len(set(kinds[1:3])) == 2, kinds  # x pairs, then y and z in some order
    for l in paired:
        nl = A.level_info(l)["nrow"]
        x, b = rng.standard_normal(nl), rng.standard_normal(nl)
        r = oracle.store_residual(H.A(l), b, x)
        bc_o = oracle.transfer_residual(H.P(l), r)
        xc_o = oracle.jacobi(H.A(l + 1), bc_o, np.zeros(len(bc_o)), 0)
        bc, xc = A.op_residual_restrict(l, b, x)
        assert np.array_equal(bc, bc_o), (name, l)
        assert np.array_equal(xc, xc_o), (name, l)
        assert np.array_equal(bc, A.op_restrict(l, A.op_residual(l, b, x))), (name, l)
    b = rng.standard_normal(n)
    out = {}
    for on in (True, False):
        A.set_paired_restriction(on)
        assert (0 in [l for l in range(A.nlevels - 1) if A.level_paired(l)]) == on
        for method in ("amg", "pcg", "pbicg"):
            x = np.zeros(n)
            h, rc = A.solve(method, b, x)
            assert rc in (0, sa.SPARSH_ENOCONV) and len(h) > 0
            out[(on, method)] = (np.array(h), x)
    for method in ("amg", "pcg", "pbicg"):
        assert np.array_equal(out[(True, method)][0], out[(False, method)][0]), (name, method)
        assert np.array_equal(out[(True, method)][1], out[(False, method)][1]), (name, method)
    A.set_paired_restriction(False)
    with pytest.raises(Exception):
        A.op_residual_restrict(0, b, b)
    A.close()


@pytest.mark.parametrize("name,gen", [
    ("p3d_30", lambda: problems.poisson3d(30)),          # level 0 pairs (2J, 2J+1), level 1 pairs rows one grid line apart
    ("p2d_150", lambda: problems.poisson2d(150)),
    ("line_40001", lambda: _line1d(40001)),              # a single-row aggregate at the end
    ("ragged", lambda: problems.random_spd(20000, 9, seed=11)),   # CSR-stream family, members from the matching
    ("fem", lambda: problems.fem_unstructured(60000, seed=5)),
])
def test_fused_prolongation_bitwise(name, gen):
    """The last post-sweep of a level adds its result to the finer level's iterate itself (OP_JACOBI_PROLONG) where the
    aggregates hold one or two rows.  Against the oracle's sweep + transfer_solution, bit for bit, on every such level and in
    every kernel family the level runs; whole solves with the fusion on and off: same histories and solutions, bit for bit."""
    rp, ci, v = gen()
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, max_iter=40))
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O)
    rng = np.random.default_rng(78)
    fused = [l for l in range(A.nlevels) if A.level_prolong_fused(l)]
    assert fused == list(range(1, A.nlevels - 1)), (name, fused, A.nlevels)  # pairwise matching: every level between the ends
    for l in fused:
        nl, nf = A.level_info(l)["nrow"], A.level_info(l - 1)["nrow"]
        x, b, xf = rng.standard_normal(nl), rng.standard_normal(nl), rng.standard_normal(nf)
        want = oracle.transfer_solution(H.P(l - 1), oracle.jacobi(H.A(l), b, x, 0), xf)
        assert np.array_equal(A.op_jacobi_prolong(l, b, x, xf), want), (name, l)
    b = rng.standard_normal(n)
    out = {}
    for on in (True, False):
        A.set_fused_prolongation(on)
        assert bool(A.level_prolong_fused(1)) == on
        for method in ("amg", "pcg", "pbicg"):
            x = np.zeros(n)
            h, rc = A.solve(method, b, x)
            assert rc in (0, sa.SPARSH_ENOCONV) and len(h) > 0
            out[(on, method)] = (np.array(h), x)
    for method in ("amg", "pcg", "pbicg"):
        assert np.array_equal(out[(True, method)][0], out[(False, method)][0]), (name, method)
        assert np.array_equal(out[(True, method)][1], out[(False, method)][1]), (name, method)
    with pytest.raises(Exception):
        A.op_jacobi_prolong(1, b, b, b)
    A.close()


def test_constant_diagonal_levels_bitwise():
    """Levels whose diagonal is one constant hand it to the zero-guess sweeps as an argument (no diag[] stream): same bits as
    with the stream, for the V-cycle alone and inside PCG; an operator with varying diagonal does not qualify."""
    rp, ci, v = problems.poisson3d(30)
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    flags = [A.level_constant_diagonal(l) for l in range(A.nlevels)]
    assert flags[0] == (True, 6.0) and flags[1][0], flags
    b = np.random.default_rng(79).standard_normal(n)
    out = {}
    for on in (True, False):
        A.set_constant_diagonal(on)
        assert A.level_constant_diagonal(0)[0] == on
        for method in ("amg", "pcg"):
            x = np.zeros(n)
            h, rc = A.solve(method, b, x)
            assert rc == 0
            out[(on, method)] = (np.array(h), x)
    for method in ("amg", "pcg"):
        assert np.array_equal(out[(True, method)][0], out[(False, method)][0])
        assert np.array_equal(out[(True, method)][1], out[(False, method)][1])
    A.close()
    rp, ci, v = problems.random_spd(20000, 9, seed=11)
    B = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    assert not B.level_constant_diagonal(0)[0]
    B.close()


@pytest.mark.parametrize("dims", [(30, 30, 30), (19, 11, 23), (64, 9, 11), (5, 40, 31), (100, 60, 40), (300, 20, 12), (700, 6, 5)])
def test_double_sweep_box_grid_bitwise(dims):
    """sdia_box2_kernel: two Jacobi sweeps per launch on box-grid levels (temporal blocking through LDS).  Forced on
    (set_double_sweep(2)) for grids whose tiles are clipped in every direction; against the oracle's sweeps, bit for bit, for
    even and odd sweep counts, from a zero guess, on every level that qualifies; whole solves against the same handle with
    single sweeps: same histories and solutions, bit for bit."""
    nx, ny, nz = dims
    rp, ci, v = problems.poisson3d(nx, ny, nz)
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).set_double_sweep(2).set_marching_ops(2).setup(sa.default_params(**QUIET, max_iter=60))  # (odd sweeps: the marching kernel)
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O)
    info = [A.level_double_sweep(l) for l in range(A.nlevels)]
    assert info[0]["on"] and info[0]["grid"] == [nx, ny, nz], info[0]
    rng = np.random.default_rng(81)
    boxes = [l for l in range(A.nlevels - 1) if info[l]["on"]]
    for l in boxes:
        nl = A.level_info(l)["nrow"]
        g = info[l]["grid"]
        assert g[0] * g[1] * g[2] == nl and A.level_kernel(l) == "sdia_tab_kernel"
        x, b = rng.standard_normal(nl), rng.standard_normal(nl)
        Ol = H.A(l)
        for sweeps in (2, 3, 4, 7):
            assert np.array_equal(A.op_jacobi(l, b, x, sweeps), oracle.jacobi(Ol, b, x, sweeps - 1)), (dims, l, sweeps)
        for sweeps in (2, 3, 4, 5, 7):  # from a zero guess: sweeps 1 - 3 are one launch reading b alone (sparsh_set_zero_start)
            assert np.array_equal(A.op_jacobi(l, b, np.zeros(nl), sweeps, x_is_zero=True), oracle.jacobi(Ol, b, np.zeros(nl), sweeps - 1)), (dims, l, sweeps)
        bz = b.copy()
        bz[:: 7] = 0.0  # zeros in the right-hand side take the plain-division branch of div_const
        assert np.array_equal(A.op_jacobi(l, bz, np.zeros(nl), 3, x_is_zero=True), oracle.jacobi(Ol, bz, np.zeros(nl), 2)), (dims, l)
    A.set_zero_start(False)
    l0 = boxes[0]
    nl = A.level_info(l0)["nrow"]
    x, b = rng.standard_normal(nl), rng.standard_normal(nl)
    assert np.array_equal(A.op_jacobi(l0, b, np.zeros(nl), 7, x_is_zero=True), oracle.jacobi(H.A(l0), b, np.zeros(nl), 6))
    A.set_zero_start(True)
    b = rng.standard_normal(n)
    out = {}
    for mode in (2, 0):
        A.set_double_sweep(mode)
        assert A.level_double_sweep(0)["on"] == (mode == 2)
        for method in ("amg", "pcg"):
            x = np.zeros(n)
            h, rc = A.solve(method, b, x)
            assert rc in (0, sa.SPARSH_ENOCONV) and len(h) > 0
            out[(mode, method)] = (np.array(h), x)
    for method in ("amg", "pcg"):
        assert np.array_equal(out[(2, method)][0], out[(0, method)][0]), (dims, method)
        assert np.array_equal(out[(2, method)][1], out[(0, method)][1]), (dims, method)
    A.close()


def test_double_sweep_inside_a_captured_graph():
    """The captured PCG iteration (use_graph) holds the double-sweep launches with their dynamic LDS and the fused transfer launches:
    replays give the eager solve's bits; switching the kernel off afterwards drops the graph and still gives the same bits."""
    rp, ci, v = problems.poisson3d(48, 40, 36)
    n = len(rp) - 1
    b = np.random.default_rng(83).standard_normal(n)
    A = sa.sp_matrix_mg(rp, ci, v).set_double_sweep(2).setup(sa.default_params(**QUIET))
    assert A.level_double_sweep(0)["on"] and A.level_double_sweep(1)["on"]
    x0 = np.zeros(n)
    h0, rc0 = A.solve("pcg", b, x0)
    G = sa.sp_matrix_mg(rp, ci, v).set_double_sweep(2).setup(sa.default_params(**QUIET, use_graph=1))
    for _ in range(2):
        x1 = np.zeros(n)
        h1, rc = G.solve("pcg", b, x1)
        assert rc == rc0 == 0 and np.array_equal(h0, h1) and np.array_equal(x0, x1)
    G.set_double_sweep(0)
    x2 = np.zeros(n)
    h2, _ = G.solve("pcg", b, x2)
    assert np.array_equal(h0, h2) and np.array_equal(x0, x2)
    A.close()
    G.close()


def test_double_sweep_needs_a_box_grid():
    """A 7-point operator that is not the stencil of a full box (one interior coupling removed) keeps the single sweeps;
    so does a 2D grid; the default mode leaves small levels alone."""
    rp, ci, v = problems.poisson3d(20)
    import scipy.sparse as sp

    M = sp.csr_matrix((v, ci, rp)).tolil()
    r = 20 * 20 * 10 + 20 * 10 + 10
    M[r, r + 1] = 0.0
    M[r + 1, r] = 0.0
    M = M.tocsr()
    M.eliminate_zeros()
    M.sort_indices()
    A = sa.sp_matrix_mg(M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data).set_double_sweep(2).setup(sa.default_params(**QUIET))
    assert not A.level_double_sweep(0)["on"] and A.level_double_sweep(0)["grid"] == [0, 0, 0]
    A.close()
    rp, ci, v = problems.poisson2d(150)
    B = sa.sp_matrix_mg(rp, ci, v).set_double_sweep(2).setup(sa.default_params(**QUIET))
    assert not B.level_double_sweep(0)["on"]
    B.close()
    rp, ci, v = problems.poisson3d(30)
    D = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))  # default: timed, levels of >= 400 000 rows only
    d = D.level_double_sweep(0)
    assert d["grid"] == [30, 30, 30] and not d["on"] and d["double_sweep_us"] == 0.0
    D.close()


@pytest.mark.parametrize("dims", [(30, 30, 30), (19, 11, 23), (64, 9, 11), (6, 40, 31), (100, 60, 40), (300, 20, 12)])
def test_marching_single_stage_kernel(dims):
    """sdia_box1_kernel (box-grid levels; forced on): the vectors it stores are the table kernel's bit for bit -- the last
    post-sweep and its prolongating variant against the oracle, residual + pair restriction against the oracle's three steps,
    A p through the PCG history; its fused dot products against the oracle to 1e-12; whole solves against the table-kernel
    path: AMG (no fused dot in the cycle) bit for bit, PCG histories to rounding."""
    nx, ny, nz = dims
    rp, ci, v = problems.poisson3d(nx, ny, nz)
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).set_marching_ops(2).setup(sa.default_params(**QUIET, max_iter=60))
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O)
    rng = np.random.default_rng(85)
    on = [l for l in range(A.nlevels - 1) if A.level_marching_ops(l)["on"]]
    assert 0 in on, [A.level_marching_ops(l) for l in range(A.nlevels)]
    for l in on:
        nl = A.level_info(l)["nrow"]
        x, b = rng.standard_normal(nl), rng.standard_normal(nl)
        Ol = H.A(l)
        for sweeps in (1, 3):  # plain sweeps through the same kernel
            assert np.array_equal(A.op_jacobi(l, b, x, sweeps), oracle.jacobi(Ol, b, x, sweeps - 1)), (dims, l, sweeps)
        if A.level_prolong_fused(l):  # last post-sweep + transfer_solution into level l - 1
            xf = rng.standard_normal(A.level_info(l - 1)["nrow"])
            want = oracle.transfer_solution(H.P(l - 1), oracle.jacobi(Ol, b, x, 0), xf)
            assert np.array_equal(A.op_jacobi_prolong(l, b, x, xf), want), (dims, l)
        if A.level_paired(l) == 1:  # residual + restriction of row pairs + the coarse zero-guess sweep
            r = oracle.store_residual(Ol, b, x)
            bc_o = oracle.transfer_residual(H.P(l), r)
            xc_o = oracle.jacobi(H.A(l + 1), bc_o, np.zeros(len(bc_o)), 0)
            bc, xc = A.op_residual_restrict(l, b, x)
            assert np.array_equal(bc, bc_o) and np.array_equal(xc, xc_o), (dims, l)
    b = rng.standard_normal(n)
    out = {}
    for mode in (2, 0):
        A.set_marching_ops(mode)
        assert A.level_marching_ops(0)["on"] == (mode == 2)
        for method in ("amg", "pcg"):
            x = np.zeros(n)
            h, rc = A.solve(method, b, x)
            assert rc in (0, sa.SPARSH_ENOCONV) and len(h) > 0
            out[(mode, method)] = (np.array(h), x)
    assert np.array_equal(out[(2, "amg")][0], out[(0, "amg")][0]) and np.array_equal(out[(2, "amg")][1], out[(0, "amg")][1]), dims
    h2, h0 = out[(2, "pcg")][0], out[(0, "pcg")][0]
    k = min(len(h2), len(h0), 25)
    assert abs(len(h2) - len(h0)) <= 1 and np.allclose(h2[:k], h0[:k], rtol=1e-8), dims
    xo, ho = oracle.solve("pcg", O, b)  # the oracle's history, as far as the capped device run goes
    m = min(len(h2), len(ho))
    _hist_ok(h2[:m], ho[:m])
    A.close()


def _stencil7(nx, ny, nz, c):
    """7-point operator on an nx x ny x nz box with a distinct constant per offset: c = (down, south, west, centre, east, north, up)."""
    import scipy.sparse as sp

    def shift(n, k):
        return sp.diags([np.ones(n - 1)], [k], shape=(n, n))

    Ix, Iy, Iz = sp.identity(nx), sp.identity(ny), sp.identity(nz)
    A = (c[3] * sp.kron(Iz, sp.kron(Iy, Ix)) + c[2] * sp.kron(Iz, sp.kron(Iy, shift(nx, -1))) + c[4] * sp.kron(Iz, sp.kron(Iy, shift(nx, 1)))
         + c[1] * sp.kron(Iz, sp.kron(shift(ny, -1), Ix)) + c[5] * sp.kron(Iz, sp.kron(shift(ny, 1), Ix))
         + c[0] * sp.kron(shift(nz, -1), sp.kron(Iy, Ix)) + c[6] * sp.kron(shift(nz, 1), sp.kron(Iy, Ix))).tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)


@pytest.mark.parametrize("dims,coef", [
    ((40, 36, 32), (-2.5, -2.0, -1.5, 10.0, -0.5, -1.0, -0.7)),   # seven different constants: every slot of the stencil is told apart
    ((36, 40, 30), (-0.3, -3.0, -0.6, 9.0, -0.9, -2.1, -0.4)),    # strongest coupling along y
    ((30, 28, 44), (-3.1, -0.2, -0.8, 9.5, -0.6, -0.4, -2.9)),    # ... along z
])
def test_box_kernels_with_a_different_constant_per_offset(dims, coef):
    """The box-grid kernels (double sweep incl. its from-zero variant, plane-marching single stage, residual + restriction along x, y, z)
    on a non-symmetric constant-coefficient stencil: a mix-up of two offsets cannot hide behind equal coefficients.  Operator level,
    against the oracle, bit for bit, on every level that qualifies."""
    nx, ny, nz = dims
    rp, ci, v = _stencil7(nx, ny, nz, coef)
    A = sa.sp_matrix_mg(rp, ci, v).set_double_sweep(2).set_marching_ops(2).setup(sa.default_params(**QUIET))
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O)
    assert A.nlevels == H.nlevels >= 3
    rng = np.random.default_rng(87)
    boxes = [l for l in range(A.nlevels - 1) if A.level_double_sweep(l)["on"]]
    assert 0 in boxes and A.level_double_sweep(0)["grid"] == [nx, ny, nz]
    kinds = set()
    for l in boxes:
        nl = A.level_info(l)["nrow"]
        x, b = rng.standard_normal(nl), rng.standard_normal(nl)
        Ol = H.A(l)
        for sweeps in (1, 2, 3, 7):
            assert np.array_equal(A.op_jacobi(l, b, x, sweeps), oracle.jacobi(Ol, b, x, sweeps - 1)), (dims, l, sweeps)
        for sweeps in (3, 4, 7):
            assert np.array_equal(A.op_jacobi(l, b, np.zeros(nl), sweeps, x_is_zero=True), oracle.jacobi(Ol, b, np.zeros(nl), sweeps - 1)), (dims, l, sweeps)
        assert np.array_equal(A.op_spmv(l, x), oracle.spmv(Ol, x)) and np.array_equal(A.op_residual(l, b, x), oracle.store_residual(Ol, b, x))
        if A.level_paired(l):
            kinds.add(A.level_paired(l))
            r = oracle.store_residual(Ol, b, x)
            bc_o = oracle.transfer_residual(H.P(l), r)
            xc_o = oracle.jacobi(H.A(l + 1), bc_o, np.zeros(len(bc_o)), 0)
            bc, xc = A.op_residual_restrict(l, b, x)
            assert np.array_equal(bc, bc_o) and np.array_equal(xc, xc_o), (dims, l, A.level_paired(l))
        if A.level_prolong_fused(l):
            xf = rng.standard_normal(A.level_info(l - 1)["nrow"])
            want = oracle.transfer_solution(H.P(l - 1), oracle.jacobi(Ol, b, x, 0), xf)
            assert np.array_equal(A.op_jacobi_prolong(l, b, x, xf), want), (dims, l)
    assert kinds, "no level took a fused residual + restriction"
    # one V-cycle and a few PCG / BiCGStab iterations against the oracle (the operator is not symmetric: only the first steps are compared)
    n = len(rp) - 1
    b = rng.standard_normal(n)
    xa = np.zeros(n)
    ha, _ = A.vcycle(b, xa, iterations=3)
    xo, ho = H.solve(b, iterations=3)
    assert np.allclose(ha, ho, rtol=1e-10) and np.linalg.norm(xa - xo) <= 1e-10 * np.linalg.norm(xo)
    A.close()


@pytest.mark.parametrize("name,gen", [
    ("p3d_30", lambda: problems.poisson3d(30)),
    ("ragged", lambda: problems.random_spd(20000, 9, seed=11)),
])
def test_deferred_x_update_bitwise(name, gen):
    """PCG: x += alpha p applied by the direction update at the end of the iteration (xp_update_kernel) instead of by the residual
    update: same histories and solutions bit for bit, through solve() and through the stepwise session."""
    rp, ci, v = gen()
    n = len(rp) - 1
    b = np.random.default_rng(89).standard_normal(n)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    out = {}
    for on in (True, False):
        A.set_deferred_x(on)
        x = np.zeros(n)
        h, rc = A.solve("pcg", b, x)
        assert rc == 0
        bd, xd = A.dev_alloc(8 * n), A.dev_alloc(8 * n)
        A.h2d(bd, b)
        A.h2d(xd, np.zeros(n))
        A.krylov_init_dev("pcg", bd, xd)
        A.krylov_step_dev(3)
        A.krylov_step_dev(2)
        xs = np.zeros(n)
        A.d2h(xs, xd)
        out[on] = (np.array(h), x, np.array(A.krylov_history()), xs)
    for k in range(4):
        assert np.array_equal(out[True][k], out[False][k]), (name, k)
    A.close()
