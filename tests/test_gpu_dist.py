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


def run_ranks(rp, ci, v, b, G, method, overlap=False, deep=None, kcfg=None, idx16=None, double_sweep=None, **kw):
    group = sa.comm_group_create(G)
    out = [None] * G
    errs = []

    def work(r):
        try:
            A = sa.sp_matrix_mg(rp, ci, v)
            A.comm_init_group(group, r)
            if idx16 is not None:
                A.set_index_compression(idx16)
            if double_sweep is not None:
                A.set_double_sweep(double_sweep)
            if kcfg is not None:
                A.set_kernel_config(*kcfg)
            if deep is not None or overlap:
                A.set_deep_halo(bool(deep) and not overlap)  # the overlapped schedule belongs to the per-sweep exchange path
            A.setup(sa.default_params(**QUIET, **kw))
            if overlap:
                A.set_overlap(True)
            lo, hi, rep = A.local_range(0)
            ex0 = A.exchanges_issued()
            x = np.zeros(hi - lo)
            if method in ("vcycle3", "comm"):
                h, rc = A.vcycle(b[lo:hi].copy(), x, iterations=3)
            else:
                h, rc = A.solve(method, b[lo:hi].copy(), x)
            out[r] = (lo, hi, rep, x, h, rc, [A.local_range(l) for l in range(A.nlevels)], A.level_format(0)[0])
            exchanges = A.exchanges_issued() - ex0
            if method == "comm":  # the communication micro-benchmarks are collective calls too
                out[r] = out[r] + ((A.bench_comm("halo", 0, 3), A.bench_comm("allreduce", 0, 3), A.bench_comm("allgather", 0, 3)),)
            out[r] = out[r] + ({"exchanges": exchanges},)
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


@pytest.mark.parametrize("deep", [True, False])
def test_compressed_indices_on_rank_local_operators(deep):
    """16-bit delta-coded column indices on the rank-local operators of a partitioned solve (halo columns are numbered after
    the own ones, so some rows are not ascending locally: their blocks keep 32-bit indices): fixed-cycle solution bitwise equal
    to the single-rank one, for the deep-halo and the per-sweep exchange schedule."""
    for gen, kw in ((lambda: problems.poisson3d(30), dict(replicate_rows=2000)),
                    (lambda: problems.fem_unstructured(40000, seed=5), dict(replicate_rows=3000))):
        rp, ci, v = gen()
        n = len(rp) - 1
        b = np.random.default_rng(4).standard_normal(n)
        A1 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
        x1 = np.zeros(n)
        A1.vcycle(b, x1, iterations=3)
        A1.close()
        res = run_ranks(rp, ci, v, b, 3, "vcycle3", deep=deep, kcfg=(0, 4, -1, -1), idx16=2, **kw)
        x = np.concatenate([r[3] for r in sorted(res, key=lambda t: t[0])])
        assert np.array_equal(x, x1)


def test_hierarchy_broadcast_from_rank0():
    """Multi-GPU setup: rank 0 builds the hierarchy once and the others receive its byte image through the transport; the
    partitioned solve is bit for bit the one obtained when every rank runs the whole setup itself."""
    for gen, kw in ((lambda: problems.poisson3d(30), dict(replicate_rows=2000)),
                    (lambda: problems.poisson3d(24), dict(replicate_rows=1000, coarsening=1)),          # Beck: general P / R in the image
                    (lambda: problems.random_spd(20000, 9, seed=11), dict(replicate_rows=1500, dense_limit=512, coarse_limit=40000, max_levels=3))):
        rp, ci, v = gen()
        n = len(rp) - 1
        b = np.random.default_rng(3).standard_normal(n)
        G = 3
        res = {}
        for share in (True, False):
            group = sa.comm_group_create(G)
            out = [None] * G
            errs = []

            def work(r):
                try:
                    A = sa.sp_matrix_mg(rp, ci, v)
                    A.comm_init_group(group, r)
                    A.set_setup_broadcast(share)
                    A.setup(sa.default_params(**QUIET, **kw))
                    lo, hi, _ = A.local_range(0)
                    x = np.zeros(hi - lo)
                    h, rc = A.solve("pcg", b[lo:hi].copy(), x)
                    out[r] = (lo, x, h, rc, A.setup_share_info(), [A.level_info(l)["nrow"] for l in range(A.nlevels)], A.coarse_info())
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
            res[share] = out
        for r in range(G):
            built, nbytes = res[True][r][4]
            assert built == (r == 0) and nbytes > 0
            assert res[False][r][4] == (True, 0)
            assert res[True][r][5] == res[False][r][5] and res[True][r][6] == res[False][r][6]
            assert np.array_equal(res[True][r][2], res[False][r][2]) and res[True][r][3] == res[False][r][3]
            assert np.array_equal(res[True][r][1], res[False][r][1])


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


@pytest.mark.parametrize("G", [2, 3])
def test_replicated_levels_take_the_fused_single_gpu_paths(G):
    """Several GPUs: the replicated coarse levels are whole levels on every rank, so they run the double sweeps, the fused
    residual + restriction and the prolongating last post-sweep exactly as one GPU does.  Two partitioned levels, four replicated
    ones (double sweep forced on); histories against the single-GPU run of the same configuration."""
    rp, ci, v = problems.poisson3d(48, 40, 36)
    n = len(rp) - 1
    b = np.random.default_rng(5).standard_normal(n)
    A1 = sa.sp_matrix_mg(rp, ci, v).set_double_sweep(2).setup(sa.default_params(**QUIET))
    assert A1.nlevels >= 5 and A1.level_double_sweep(2)["on"] and A1.level_paired(2) and A1.level_prolong_fused(3)
    for method in ("pcg", "amg"):
        x1 = np.zeros(n)
        h1, rc1 = A1.solve(method, b, x1)
        res = run_ranks(rp, ci, v, b, G, method, double_sweep=2, replicate_rows=20000)
        reps = [rep for (lo, hi, rep) in res[0][6]]
        assert reps[:2] == [False, False] and all(reps[2:]), reps
        x = np.concatenate([r[3] for r in sorted(res, key=lambda t: t[0])])
        for r in res:
            assert r[5] == rc1 and np.array_equal(r[4], res[0][4])
        h = res[0][4]
        assert len(h) == len(h1)
        tol = np.where(h1 >= 1e-6 * h1[0], 1e-8, 1e-4)
        assert np.all(np.abs(h - h1) <= tol * h1), np.abs(h / h1 - 1).max()
        assert np.linalg.norm(x - x1) <= 1e-8 * np.linalg.norm(x1)
    A1.close()


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


@pytest.mark.parametrize("G", [2, 3, 8])
def test_deep_halo_same_rows_fewer_exchanges(G):
    """Deep-halo smoothing (default): a rank carries K = sweeps + 1 ghost layers, exchanges them once per smoothing leg
    and sweeps a shrinking row set.  Row by row it is the arithmetic of the per-sweep-exchange schedule and of the
    one-rank cycle: a fixed number of V-cycles gives BITWISE the same solution; the transport is called ~4x less."""
    rp, ci, v = problems.poisson3d(48)
    n = len(rp) - 1
    b = np.random.default_rng(9).standard_normal(n)
    A1 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    x1 = np.zeros(n)
    A1.vcycle(b, x1, iterations=3)
    deep = run_ranks(rp, ci, v, b, G, "vcycle3", deep=True, replicate_rows=4000)
    flat = run_ranks(rp, ci, v, b, G, "vcycle3", deep=False, replicate_rows=4000)
    xd = np.concatenate([r[3] for r in sorted(deep, key=lambda t: t[0])])
    xf = np.concatenate([r[3] for r in sorted(flat, key=lambda t: t[0])])
    assert np.array_equal(xd, x1) and np.array_equal(xf, x1)
    assert sum(1 for (lo, hi, rep) in deep[0][6] if not rep) >= 3          # several partitioned levels
    ed, ef = deep[0][-1]["exchanges"], flat[0][-1]["exchanges"]
    assert ed * 3 < ef, (ed, ef)
    # other sweep counts cut the legs differently (K = sweeps + 1 layers)
    for nu in (1, 2, 4):
        A1 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, sweeps=nu))
        x1 = np.zeros(n)
        A1.vcycle(b, x1, iterations=3)
        res = run_ranks(rp, ci, v, b, G, "vcycle3", deep=True, replicate_rows=4000, sweeps=nu)
        x = np.concatenate([r[3] for r in sorted(res, key=lambda t: t[0])])
        assert np.array_equal(x, x1), nu


def test_deep_halo_unstructured_and_beck():
    """Ghost layers on a ragged operator (layers found by graph distance, owners anywhere) and with Beck's general P / R."""
    for gen, kw in ((lambda: problems.fem_unstructured(30000, seed=5), dict(replicate_rows=3000)),
                    (lambda: problems.poisson3d(36), dict(replicate_rows=3000, coarsening=1))):
        rp, ci, v = gen()
        n = len(rp) - 1
        b = np.random.default_rng(1).standard_normal(n) * 1e-3
        A1 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, **{k: w for k, w in kw.items() if k != "replicate_rows"}))
        x1 = np.zeros(n)
        A1.vcycle(b, x1, iterations=3)
        h1p, _ = A1.solve("pcg", b, np.zeros(n))
        for G in (2, 4):
            res = run_ranks(rp, ci, v, b, G, "vcycle3", deep=True, **kw)
            assert not res[0][2]
            x = np.concatenate([r[3] for r in sorted(res, key=lambda t: t[0])])
            assert np.array_equal(x, x1)
            resp = run_ranks(rp, ci, v, b, G, "pcg", deep=True, **kw)
            hp = resp[0][4]
            assert len(hp) == len(h1p)
            tol = np.where(h1p >= 1e-6 * h1p[0], 1e-8, 1e-4)
            assert np.all(np.abs(hp - h1p) <= tol * h1p)


def test_deep_halo_local_operator_every_family_every_prefix():
    """The rank-local operator of a deep-halo level (own rows | padding | ghost layers) through every kernel family over
    every row prefix a smoothing leg launches: bitwise equal.  Guards two things the ghost rows exposed: a prefix that
    ends inside a slice (the table kernel takes x[r+1] from the neighbouring lane's gather, so lanes past the prefix must
    keep their own row), and ghost rows whose neighbours in other layers sit at local offsets of the opposite sign
    (+plane <-> -plane: such a slice matches the level's stencil as a set but not in order, and must leave the table path)."""
    rp, ci, v = problems.poisson3d(48)
    G = 2
    group = sa.comm_group_create(G)
    errs, seen = [], []

    def work(r):
        try:
            A = sa.sp_matrix_mg(rp, ci, v)
            A.comm_init_group(group, r)
            A.setup(sa.default_params(**QUIET, replicate_rows=60000))
            info = A.deep_info(0)
            assert info["K"] == 8 and info["layer_end"][0] == info["npad"]
            lo, hi, _ = A.local_range(0)
            xe = np.random.default_rng(5 + r).standard_normal(info["cols"])
            for rows in [hi - lo] + info["layer_end"][:-1]:
                ys = {}
                for kind in (0, 1, 2, 3):
                    A.set_kernel_config(kind=kind)
                    ys[A.level_kernel(0)] = A.deep_prefix_spmv(0, rows, xe)
                seen.append(set(ys))
                ref = ys["csr_block_kernel"]
                for name, y in ys.items():
                    assert np.array_equal(y, ref), (r, rows, name, int(np.flatnonzero(y != ref)[0]))
            A.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts) and not errs, errs
    sa.comm_group_destroy(group)
    assert any("sdia_tab_kernel" in s for s in seen)


def _tuned_run(rp, ci, v, b, G, delay_us, **kw):
    """G virtual ranks with the measured schedule (replicate_rows = 0) over a transport whose every call costs delay_us more."""
    group = sa.comm_group_create(G)
    sa.comm_group_set_delay(group, delay_us)
    out = [None] * G
    errs = []

    def work(r):
        try:
            A = sa.sp_matrix_mg(rp, ci, v)
            A.comm_init_group(group, r)
            A.setup(sa.default_params(**QUIET, replicate_rows=0, **kw))
            lo, hi, _ = A.local_range(0)
            x = np.zeros(hi - lo)
            A.vcycle(b[lo:hi].copy(), x, iterations=3)
            out[r] = (lo, x, A.comm_schedule(), A.comm_measured(), [A.local_range(l)[2] for l in range(A.nlevels)])
            A.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    ts = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in ts), "a virtual rank hung"
    assert not errs, errs
    sa.comm_group_destroy(group)
    return out


@pytest.mark.parametrize("G", [2, 4])
def test_measured_schedule_moves_with_the_transport_and_keeps_the_bits(G):
    """the multi-rank schedule is chosen from measurements, not constants.  Two / four virtual ranks on one GPU, the
    in-process transport once as it is (a device copy: a few microseconds per call) and once with 400 us added to every call (a slow
    link).  The tuner must see the difference (measured exchange / all-reduce / all-gather times), replicate more levels over the
    slow link than over the fast one, never choose the exchange-per-sweep schedule on the slow link, hand every rank the same
    table -- and whatever it chooses, three V-cycles give the single-rank iterate bit for bit."""
    rp, ci, v = problems.poisson3d(64)  # 262 144 rows
    n = len(rp) - 1
    b = np.random.default_rng(7).standard_normal(n)
    A1 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    x1 = np.zeros(n)
    A1.vcycle(b, x1, iterations=3)
    A1.close()
    fast = _tuned_run(rp, ci, v, b, G, 0.0)
    slow = _tuned_run(rp, ci, v, b, G, 400.0)
    for res in (fast, slow):
        assert all(r[2] is not None and r[3] is not None for r in res)
        assert all(r[2] == res[0][2] and r[3] == res[0][3] for r in res), "ranks disagree on the schedule"
        x = np.concatenate([r[1] for r in sorted(res, key=lambda t: t[0])]) if not res[0][4][0] else res[0][1]
        assert np.array_equal(x, x1)
        # the table says what the engine then did
        assert [c["partitioned"] for c in res[0][2]] == [not rep for rep in res[0][4]]
    mf, ms = fast[0][3], slow[0][3]
    assert ms["exchange_us"] >= mf["exchange_us"] + 300 and ms["allreduce_us"] >= mf["allreduce_us"] + 300 and ms["allgather_us"] >= mf["allgather_us"] + 300
    nf = sum(c["partitioned"] for c in fast[0][2])
    ns = sum(c["partitioned"] for c in slow[0][2])
    assert ns <= nf, (nf, ns)
    for c in slow[0][2]:
        if c["partitioned"]:
            assert c["deep_halo"] and c["model_us_deep_halo"] < c["model_us_exchange_per_sweep"]
    # over a 400 us link a 262 144-row problem is cheaper replicated than partitioned (the whole V-cycle is ~1 ms)
    assert ns == 0, slow[0][2]
