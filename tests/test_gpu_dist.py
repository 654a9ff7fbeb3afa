"""Row-block partitioned solver on ONE GPU: G virtual ranks (host threads, in-process transport)
must reproduce the single-rank solve.  Exercises partitioning, halo plans, the replicated coarse
levels with their all-gather, and the all-reduced scalars; the RCCL transport differs only in how
bytes move (comm.cpp)."""
import threading

import numpy as np
import pytest

import sparsh_amg_amd as sa
from sparsh_amg_amd import problems

pytestmark = pytest.mark.gpu

QUIET = dict(print_setup=0, print_solve=0)


def run_ranks(rp, ci, v, b, G, method, overlap=False, **kw):
    group = sa.comm_group_create(G)
    out = [None] * G
    errs = []

    def work(r):
        try:
            A = sa.sp_matrix_mg(rp, ci, v)
            A.comm_init_group(group, r)
            A.setup(sa.default_params(**QUIET, **kw))
            if overlap:
                A.set_overlap(True)
            lo, hi, rep = A.local_range(0)
            x = np.zeros(hi - lo)
            if method in ("vcycle3", "comm"):
                h, rc = A.vcycle(b[lo:hi].copy(), x, iterations=3)
            else:
                h, rc = A.solve(method, b[lo:hi].copy(), x)
            out[r] = (lo, hi, rep, x, h, rc, [A.local_range(l) for l in range(A.nlevels)], A.level_format(0)[0])
            if method == "comm":  # the communication micro-benchmarks are collective calls too
                out[r] = out[r] + ((A.bench_comm("halo", 0, 3), A.bench_comm("allreduce", 0, 3), A.bench_comm("allgather", 0, 3)),)
            A.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    alive = [t.is_alive() for t in ts]
    assert not any(alive), "a virtual rank hung"
    assert not errs, errs
    sa.comm_group_destroy(group)
    return out


CASES = {
    "p3d": (lambda: problems.poisson3d(30), dict(replicate_rows=2000)),
    "p2d": (lambda: problems.poisson2d(150), dict(replicate_rows=1500)),
    "ragged": (lambda: problems.random_spd(20000, 9, seed=11), dict(replicate_rows=1500)),
}


@pytest.mark.parametrize("G", [2, 3, 4])
@pytest.mark.parametrize("name", list(CASES))
def test_virtual_ranks_match_single_rank(name, G):
    gen, kw = CASES[name]
    rp, ci, v = gen()
    n = len(rp) - 1
    rng = np.random.default_rng(2)
    b = rng.standard_normal(n)
    A1 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    for method in ("pcg", "amg", "pbicg", "cg"):
        x1 = np.zeros(n)
        h1, rc1 = A1.solve(method, b, x1)
        res = run_ranks(rp, ci, v, b, G, method, **kw)
        # the blocks tile level 0 exactly once
        spans = sorted((r[0], r[1]) for r in res)
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[k][1] == spans[k + 1][0] for k in range(G - 1))
        assert not res[0][2], "level 0 should be partitioned in this test"
        nparts = sum(1 for (lo, hi, rep) in res[0][6] if not rep)
        assert nparts >= 2, "expected at least two partitioned levels"
        x = np.concatenate([r[3] for r in sorted(res, key=lambda t: t[0])])
        for r in res:
            assert r[5] == rc1
            assert np.array_equal(r[4], res[0][4]), "ranks disagree on the residual history"
        h = res[0][4]
        if method == "cg":
            k = min(20, len(h), len(h1))
            assert np.allclose(h[:k], h1[:k], rtol=1e-9)
        else:
            assert len(h) == len(h1)
            tol = np.where(h1 >= 1e-6 * h1[0], 1e-8, 1e-4)
            assert np.all(np.abs(h - h1) <= tol * h1), np.abs(h / h1 - 1).max()
        assert np.linalg.norm(x - x1) <= 1e-8 * np.linalg.norm(x1)


def test_fixed_cycles_bitwise_rows():
    """With reductions out of the picture (fixed number of V-cycles) the partitioned cycle is
    the same arithmetic row by row: the solution matches the single-rank one bitwise."""
    rp, ci, v = problems.poisson3d(30)
    n = len(rp) - 1
    b = np.ones(n)
    A1 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    x1 = np.zeros(n)
    A1.vcycle(b, x1, iterations=3)
    res = run_ranks(rp, ci, v, b, 3, "vcycle3", replicate_rows=2000)
    x = np.concatenate([r[3] for r in sorted(res, key=lambda t: t[0])])
    assert np.array_equal(x, x1)
    # rank-local blocks of a stencil operator keep a sliced layout, halo columns included: sliced
    # diagonals, or sliced ELL for a small block whose halo slices leave it without a stencil table
    assert all(r[7] in (2, 3) for r in res), [r[7] for r in res]


def test_everything_replicated_small_problem():
    rp, ci, v = problems.poisson2d(60)
    n = len(rp) - 1
    b = np.ones(n)
    res = run_ranks(rp, ci, v, b, 2, "pcg")  # default replicate_rows > n: every rank solves it all
    assert all(r[2] for r in res) and all(r[1] - r[0] == n for r in res)
    assert np.array_equal(res[0][3], res[1][3])


def test_eight_ranks_deep_partition():
    """8 virtual ranks, five partitioned levels (forward and backward HEM sweeps alternate, so the
    coarse blocks are handed out in alternating rank order), PCG history vs one rank."""
    rp, ci, v = problems.poisson3d(48)
    n = len(rp) - 1
    b = np.random.default_rng(3).standard_normal(n)
    A1 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    x1 = np.zeros(n)
    h1, _ = A1.solve("pcg", b, x1)
    res = run_ranks(rp, ci, v, b, 8, "pcg", replicate_rows=4000)
    levels = res[0][6]
    assert sum(1 for (lo, hi, rep) in levels if not rep) >= 4
    # the rank owning the first rows of level 1 is the one owning the first rows of level 0
    # (forward sweep), while level 2 (backward sweep on level 1) is handed out in reverse
    first0 = min(range(8), key=lambda r: res[r][6][0][0])
    first1 = min(range(8), key=lambda r: res[r][6][1][0])
    first2 = min(range(8), key=lambda r: res[r][6][2][0])
    assert first0 == first1 and first2 != first1
    x = np.concatenate([r[3] for r in sorted(res, key=lambda t: t[0])])
    h = res[0][4]
    assert len(h) == len(h1)
    tol = np.where(h1 >= 1e-6 * h1[0], 1e-8, 1e-4)
    assert np.all(np.abs(h - h1) <= tol * h1)
    assert np.linalg.norm(x - x1) <= 1e-8 * np.linalg.norm(x1)


@pytest.mark.parametrize("G", [2, 4])
def test_overlapped_exchange_same_results(G):
    """Overlap mode (exchange on a second stream while interior slices run, boundary slices after):
    a different launch schedule, the same arithmetic -- bitwise equal to the non-overlapped run."""
    rp, ci, v = problems.poisson3d(36)
    n = len(rp) - 1
    b = np.random.default_rng(5).standard_normal(n)
    base = run_ranks(rp, ci, v, b, G, "vcycle3", replicate_rows=3000)
    over = run_ranks(rp, ci, v, b, G, "vcycle3", overlap=True, replicate_rows=3000)
    xb = np.concatenate([r[3] for r in sorted(base, key=lambda t: t[0])])
    xo = np.concatenate([r[3] for r in sorted(over, key=lambda t: t[0])])
    assert np.array_equal(xb, xo)
    assert all(r[7] in (2, 3) for r in over)  # sliced blocks: the overlap path is really taken
    hb = run_ranks(rp, ci, v, b, G, "pcg", replicate_rows=3000)
    ho = run_ranks(rp, ci, v, b, G, "pcg", overlap=True, replicate_rows=3000)
    # the fused reductions add their partials in a different order (interior groups, then boundary
    # groups): histories agree to rounding, not bitwise
    h0, h1 = hb[0][4], ho[0][4]
    assert len(h0) == len(h1) and np.all(np.abs(h0 - h1) <= 1e-9 * h0)
    x0 = np.concatenate([r[3] for r in sorted(hb, key=lambda t: t[0])])
    x1 = np.concatenate([r[3] for r in sorted(ho, key=lambda t: t[0])])
    assert np.linalg.norm(x0 - x1) <= 1e-10 * np.linalg.norm(x0)


def test_comm_microbenchmarks_are_collective_and_harmless():
    """sparsh_bench_comm (bench.py's multi-GPU diagnostics): every virtual rank times the halo exchange,
    the scalar all-reduce and the boundary all-gather; the calls return and a later solve is unaffected."""
    rp, ci, v = problems.poisson3d(30)
    n = len(rp) - 1
    b = np.ones(n)
    res = run_ranks(rp, ci, v, b, 3, "comm", replicate_rows=2000)
    for r in res:
        t_halo, t_red, t_gather = r[8]
        assert t_halo > 0 and t_red > 0 and t_gather > 0, r[8]
    ref = run_ranks(rp, ci, v, b, 3, "vcycle3", replicate_rows=2000)
    for r, q in zip(res, ref):
        assert np.array_equal(r[3], q[3]) and np.array_equal(r[4], q[4])
    A1 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    assert A1.bench_comm("halo", 0, 3) == -1.0   # single GPU: no such step
    A1.close()


def test_transport_failure_is_reported_not_swallowed():
    """A failed halo exchange must surface as SPARSH_ECOMM from the solver (and stay sticky on the
    handle), not return SPARSH_OK with garbage.  The in-process transport fails on every rank from its
    40th exchange on (fault-injection hook)."""
    rp, ci, v = problems.poisson3d(30)
    n = len(rp) - 1
    G = 2
    group = sa.comm_group_create(G)
    res = [None] * G
    errs = []
    gate = threading.Barrier(G)

    def work(r):
        try:
            A = sa.sp_matrix_mg(rp, ci, v)
            A.comm_init_group(group, r)
            A.setup(sa.default_params(**QUIET, replicate_rows=2000))
            lo, hi, _ = A.local_range(0)
            b = np.ones(hi - lo)
            x = np.zeros(hi - lo)
            h, rc = A.solve("pcg", b, x)          # healthy transport first
            assert rc == 0
            A.sync()
            gate.wait(timeout=120)
            if r == 0:
                sa.comm_group_fail_after(group, 40)
            gate.wait(timeout=120)
            codes, msgs = [], []
            for method in ("pcg", "amg"):
                try:
                    A.solve(method, b, np.zeros(hi - lo))
                    codes.append(0)
                except sa.SparshError as e:
                    codes.append(e.code)
                    msgs.append(str(e))
            res[r] = (codes, msgs)
            A.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a virtual rank hung"
    assert not errs, errs
    sa.comm_group_destroy(group)
    for codes, msg in res:
        assert codes == [sa.SPARSH_ECOMM, sa.SPARSH_ECOMM], (codes, msg)   # second call: sticky
