"""Host half of the product (coarsening, Galerkin, coarse inverse) vs the oracle; and the C-ABI
load/export check.  CPU only -- no compute kernel is called."""
import os
import re

import numpy as np
import pytest

import oracle
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems
from conftest import C0_MATRIX, C0_RHS, ROOT


def _same_hierarchy(A: sa.sp_matrix_mg, H: oracle.Hierarchy):
    assert A.nlevels == H.nlevels
    for l in range(A.nlevels):
        rp, ci, v, _ = A.level_csr(l, "A")
        orp, oci, ov = H.A(l).arrays()
        assert np.array_equal(rp, orp) and np.array_equal(ci, oci)
        # the aggregation-specialised Galerkin product adds in the same order as the oracle's
        # two-product form: bitwise equal
        assert np.array_equal(v, ov), f"level {l}: max diff {np.abs(v - ov).max()}"
        if l + 1 < A.nlevels:
            rp, ci, v, ncol = A.level_csr(l, "P")
            orp, oci, ov = H.P(l).arrays()
            assert ncol == H.P(l).shape[1]
            assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and np.array_equal(v, ov)


@pytest.mark.parametrize("gen,kw", [("poisson2d", dict(n=96)), ("poisson3d", dict(n=24)), ("random_spd", dict(n=9000, nnz_per_row=8, seed=5))])
def test_hierarchy_matches_oracle_hem(gen, kw):
    rp, ci, v = getattr(problems, gen)(**kw)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0), host_only=True)
    H = oracle.Hierarchy(oracle.Csr(rp, ci, v))
    _same_hierarchy(A, H)


@pytest.mark.parametrize("gen,kw", [("poisson3d", dict(n=24)), ("random_spd", dict(n=9000, nnz_per_row=8, seed=6))])
def test_hierarchy_matches_oracle_beck(gen, kw):
    rp, ci, v = getattr(problems, gen)(**kw)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, coarsening=1), host_only=True)
    H = oracle.Hierarchy(oracle.Csr(rp, ci, v), oracle.params(coarsening=1))
    assert A.nlevels == H.nlevels
    for l in range(A.nlevels):
        S = A.level_scipy(l, "A")
        O = H.A(l).to_scipy()
        assert S.shape == O.shape and S.nnz == O.nnz
        assert abs(S - O).max() <= 1e-13 * abs(O).max()


def test_c0_readcoo_equals_fixture(have_c0, c0_arrays):
    if not have_c0:
        pytest.skip("/root/reference exists in the build container only")
    A, b = sa.readcoo(C0_MATRIX, C0_RHS)
    rp, ci, v = A.rowptr, A.colindex, A.val
    assert np.array_equal(rp, c0_arrays[0]) and np.array_equal(ci, c0_arrays[1])
    assert np.array_equal(v, c0_arrays[2]) and np.array_equal(b, c0_arrays[3])


def test_c0_hierarchy_golden(c0_arrays, golden):
    rp, ci, v, b = c0_arrays
    A = sa.sp_matrix_mg(rp, ci, v)
    A.setup(sa.default_params(print_setup=0), host_only=True)
    g = golden["C0"]["hem"]
    assert [A.level_info(l)["nrow"] for l in range(A.nlevels)] == g["levels_nrow"]
    assert [A.level_info(l)["nnz"] for l in range(A.nlevels)] == g["levels_nnz_stored"]
    assert abs(np.linalg.norm(b) - 10.8032) < 1e-3


def test_golden_level_shapes(golden):
    for key, gen in (("poisson2d_256", lambda: problems.poisson2d(256)), ("poisson3d_40", lambda: problems.poisson3d(40))):
        rp, ci, v = gen()
        A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0), host_only=True)
        g = golden[key]["hem"]
        assert [A.level_info(l)["nrow"] for l in range(A.nlevels)] == g["levels_nrow"]
        assert [A.level_info(l)["nnz"] for l in range(A.nlevels)] == g["levels_nnz_stored"]


def test_coarse_inverse():
    rp, ci, v = problems.poisson3d(20)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0), host_only=True)
    AL = A.level_scipy(A.nlevels - 1, "A")
    inv = A.coarse_inverse()
    n = AL.shape[0]
    assert 2000 <= n <= 4000
    err = np.abs(AL @ inv - np.eye(n)).max()
    assert err < 1e-12
    # against the oracle's direct solve
    H = oracle.Hierarchy(oracle.Csr(rp, ci, v))
    b = np.random.default_rng(0).standard_normal(n)
    assert np.allclose(inv @ b, H.coarse_solve(b), rtol=1e-11, atol=1e-13)


def test_coarse_inverse_needs_pivoting():
    # non-symmetric, not diagonally dominant: exercises the row interchanges of the banded LU
    rng = np.random.default_rng(3)
    n = 600
    import scipy.sparse as sp

    M = sp.random(n, n, density=0.01, random_state=4, format="csr") + sp.diags(rng.standard_normal(n) * 0.05)
    M = (M + sp.diags(np.ones(n - 1), 1) + sp.diags(np.ones(n - 1) * 0.5, -1)).tocsr()
    M.sort_indices()
    A = sa.sp_matrix_mg(M.indptr, M.indices, M.data).setup(sa.default_params(print_setup=0), host_only=True)
    assert A.nlevels == 1
    inv = A.coarse_inverse()
    assert np.abs(M @ inv - np.eye(n)).max() < 1e-8


def test_extended_hierarchy():
    # max_levels too small for coarse_limit (rows alone deciding): hierarchy is extended (documented deviation)
    rp, ci, v = problems.poisson2d(200)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, max_levels=2, coarse_limit=5000, coarse_factor_mb=0), host_only=True)
    assert A.nlevels > 2 and A.coarse_info()["extended"]
    assert A.level_info(A.nlevels - 1)["nrow"] <= 5000
    A.close()
    # default: the reference's own coarsest level stays when its nested-dissection factors are estimated affordable (20 000 rows of a 2D operator)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, max_levels=2, coarse_limit=5000), host_only=True)
    assert A.nlevels == 2 and not A.coarse_info()["extended"]
    A.close()
    # ... and is given up when they are not: the same with a 1 MB budget
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, max_levels=2, coarse_limit=5000, coarse_factor_mb=1), host_only=True)
    assert A.nlevels > 2 and A.coarse_info()["extended"]
    A.close()


def test_index16_encoder_roundtrip():
    """Host encoder of the 16-bit delta column form (csr_rowlane16_kernel): decodes back to colindex; blocks with a gap of
    65536 or more, an unsorted row or one over-long row keep their 32-bit indices."""
    import scipy.sparse as sp

    rp, ci, v = problems.poisson3d(48)            # +-2304 neighbours: every block fits
    b16, nb = sa.index16_roundtrip(rp, ci)
    assert b16 == nb > 100
    rp, ci, v = problems.fem_unstructured(30000, seed=3)
    b16, nb = sa.index16_roundtrip(rp, ci)
    assert b16 == nb
    n = 200000
    M = sp.diags([np.ones(n - 1), 2 * np.ones(n), np.ones(n - 1)], [-1, 0, 1], format="lil")
    M[7, 150000] = 1.0                      # gap > 65535 inside a row
    M[100000, :3000] = 1.0                  # a row longer than the LDS buffer of the kernel
    M = M.tocsr()
    M.sort_indices()
    b16, nb = sa.index16_roundtrip(M.indptr, M.indices)
    assert 0 < nb - b16 <= 4
    ci2 = M.indices.copy()
    s0, s1 = M.indptr[150000], M.indptr[150001]
    ci2[s0:s1] = ci2[s0:s1][::-1]           # an unsorted row: its block falls back, nothing breaks
    b16b, nb2 = sa.index16_roundtrip(M.indptr, ci2)
    assert nb2 == nb and b16b == b16 - 1


def test_hierarchy_image_roundtrip():
    """The byte image rank 0 broadcasts in a multi-GPU setup reproduces the hierarchy array by array (HEM: aggregation
    P; Beck: general P / R; dense coarse inverse or none), and a truncated image is refused, not half-read."""
    for gen, kw in ((lambda: problems.poisson3d(20), {}), (lambda: problems.poisson2d(60), dict(coarsening=1)),
                    (lambda: problems.poisson2d(80), dict(max_levels=2, dense_limit=256, coarse_limit=100000))):   # 3200-row coarsest level, no dense inverse
        rp, ci, v = gen()
        A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0, **kw), host_only=True)
        nbytes = A.hierarchy_roundtrip()
        assert nbytes > 12 * (A.level_info(1)["nnz"] if A.nlevels > 1 else 0)
        for cut in (0, 8, nbytes // 2, nbytes - 1):
            with pytest.raises(sa.SparshError):
                A.hierarchy_roundtrip(cut)
        A.close()


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sparsh_amg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(sparsh_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    import ctypes

    L = ctypes.CDLL(sa.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_no_gpu_fails_loudly():
    if sa.device_count() > 0:
        pytest.skip("a GPU is visible")
    rp, ci, v = problems.poisson2d(40)
    A = sa.sp_matrix_mg(rp, ci, v)
    with pytest.raises(sa.SparshError) as e:
        A.setup(sa.default_params(print_setup=0))
    assert e.value.code == sa.SPARSH_ENODEV
    # solvers refuse to run without a device-side setup: no CPU fallback
    A.setup(sa.default_params(print_setup=0), host_only=True)
    with pytest.raises(sa.SparshError) as e:
        A.solve("pcg", np.ones(A.nrow), np.zeros(A.nrow))
    assert e.value.code == sa.SPARSH_ESTATE


def test_product_does_not_touch_oracle():
    # the product tree must not reference oracle/ in any form
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sparsh_amg_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"amg_oracle|import oracle|from oracle|oracle/", txt):
                    bad.append(f)
    assert not bad, bad


@pytest.mark.parametrize("args", [("12", "3", "8"), ("40", "2", "8"), ("9", "3", "3"), ("30", "2", "2"), ("70", "2", "64")])
def test_block_tridiagonal_plan_on_host(tmp_path_factory, args):
    """csrc/coarse.cpp bt_make_plan (RCM blocks, diag/out/in pieces, inward/outward schedule) driven through a
    dense host emulation of the device kernels' algebra: the twisted block factorisation must solve A x = b."""
    import subprocess

    d = tmp_path_factory.getbasetemp() / "btplan"
    d.mkdir(exist_ok=True)
    exe = d / "bt_plan_check"
    if not exe.exists():
        lib_dir = os.path.join(ROOT, "sparsh_amg_amd")
        cmd = ["g++", "-std=c++17", "-O2", "-D__HIP_PLATFORM_AMD__", f"-I{os.path.join(lib_dir, 'csrc')}", "-I/opt/rocm/include",
               os.path.join(ROOT, "tests", "cpp", "bt_plan_check.cpp"), "-o", str(exe), f"-L{lib_dir}", "-lsparsh_amg",
               f"-Wl,-rpath,{lib_dir}", "-L/opt/rocm/lib", "-L/opt/rocm/lib/llvm/lib", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib/llvm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    assert "relerr" in r.stdout


def _nd_plan_check(tmp_path_factory, args, env=None):
    import subprocess

    d = tmp_path_factory.getbasetemp() / "ndplan"
    d.mkdir(exist_ok=True)
    exe = d / "nd_plan_check"
    if not exe.exists():
        csrc = os.path.join(ROOT, "sparsh_amg_amd", "csrc")
        cmd = ["g++", "-std=c++17", "-O2", f"-I{csrc}", os.path.join(ROOT, "tests", "cpp", "nd_plan_check.cpp"), os.path.join(csrc, "nd_plan.cpp"), "-o", str(exe)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=600, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    return r.stdout


@pytest.mark.parametrize("args", [
    ("grid3", "10", "32"), ("grid3", "16", "64"), ("grid3", "12", "32", "0"), ("grid2", "40", "32"), ("grid2", "100", "64"),
    ("rand", "3000", "32"), ("blocks", "2000", "16"), ("grid3", "3", "64"), ("grid2", "7", "4"), ("box", "9006020", "16", "48"),
    ("grid3", "16", "32", "96", "1024"), ("arrow", "300", "16"), ("diag", "200", "16"),
])
def test_nested_dissection_plan_solves_on_the_host(tmp_path_factory, args):
    """csrc/nd_plan.cpp without a GPU: ordering, update sets, front positions, extend-add maps, forward segments and backward
    gather lists run through a plain host emulation of the device kernels' algebra (tests/cpp/nd_plan_check.cpp): the multifrontal
    factorisation must solve A x = b to 1e-12 on grids, a box, random nonsymmetric patterns and a matrix made of disconnected pieces."""
    out = _nd_plan_check(tmp_path_factory, list(args))
    assert "relerr" in out


def test_nested_dissection_plan_of_the_100cubed_coarsest_level(tmp_path_factory):
    """Shape of the plan for a 25 x 25 x 50 grid operator (the 31 250-row coarsest level of 100^3 has this graph up to the
    aggregation pattern): the point of the form is few dependent launches and small factors against the 31-step chain and the
    260 MB of the block-tridiagonal form."""
    out = _nd_plan_check(tmp_path_factory, ["box", "25025050", "64"], {"ND_PLAN_ONLY": "1"})
    head = out.splitlines()[0].split()
    levels, factor_mb = int(head[head.index("levels") + 1]), float(head[head.index("factor_MB") + 1])
    assert levels <= 9 and factor_mb < 140.0, out


def test_reference_level_policy_is_the_default():
    """Up to coarse_limit (40 000) rows left over, max_levels = 6 is honoured exactly as the reference does
    (src/AMG_phases.cpp:51,77): 100^3 -> 6 levels with a 31 250-row coarsest level for the direct solver."""
    rp, ci, v = problems.poisson3d(100)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0), host_only=True)
    assert [A.level_info(l)["nrow"] for l in range(A.nlevels)] == [1000000, 500000, 250000, 125000, 62500, 31250]
    info = A.coarse_info()
    assert info["rows"] == 31250 and not info["dense"] and not info["extended"]
    with pytest.raises(sa.SparshError):
        A.coarse_inverse()  # no dense inverse above dense_limit


def test_multi_rank_schedule_decision_follows_the_measured_transport():
    """The decision half of the multi-rank tuner (Engine::decide_comm_schedule) without a device: a variable-coefficient 100^3
    operator (8 B per entry streamed: a sweep over 1 M rows costs ~20 us on one GPU) on 8 ranks, from a free link to a hopeless one --
    fewer levels stay partitioned as a transport call gets dearer, a link that costs more than a whole replicated V-cycle partitions
    nothing, latency-bound links choose deep halos (4 exchanges per level and cycle instead of 16) and a link whose calls are free
    chooses the exchange-per-sweep schedule (no redundant ghost-row sweeps).  On the constant-coefficient operator (25 B per row on the
    table path: the same sweep costs 8 us) a 15 us link is not worth one partitioned level: round 2's replicate_rows = 1.5 M, derived."""
    rp, ci, v = problems.poisson3d(100)
    dev = dict(sweep_floor_us=3.3, sweep_us_per_MB=0.2)   # 5 TB/s
    link = lambda lat: dict(exchange_us=lat, exchange_us_per_MB=10.0, allreduce_us=lat, allgather_us=2 * lat, allgather_us_per_MB=10.0)  # noqa: E731
    vv = v * (1.0 + 0.1 * np.random.default_rng(0).random(len(v)))
    A = sa.sp_matrix_mg(rp, ci, vv).setup(sa.default_params(print_setup=0, print_solve=0), host_only=True)
    npart = []
    for lat in (0.0, 5.0, 15.0, 100.0, 5000.0):
        t = A.plan_comm_schedule(8, **link(lat), **dev)
        assert t is not None and len(t) == A.nlevels and [c["rows"] for c in t] == [A.level_info(l)["nrow"] for l in range(A.nlevels)]
        parts = [c["partitioned"] for c in t]
        assert parts == sorted(parts, reverse=True) and not parts[-1]        # a prefix of levels; the coarsest one never
        npart.append(sum(parts))
        for c in t[:-1]:
            assert c["boundary_rows"] > 0 and c["model_us_replicated"] > 0
        if lat == 0.0:
            assert npart[-1] >= 3 and not any(c["deep_halo"] for c in t)     # free exchanges: no reason to sweep ghost rows redundantly
        if lat == 15.0:
            assert npart[-1] >= 1 and all(c["deep_halo"] for c in t if c["partitioned"])
            assert all(c["model_us_deep_halo"] < c["model_us_exchange_per_sweep"] for c in t[:-1])
    assert npart == sorted(npart, reverse=True) and npart[-1] == 0, npart
    A.close()
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0), host_only=True)
    assert not any(c["partitioned"] for c in A.plan_comm_schedule(8, **link(15.0), **dev))
    assert sum(c["partitioned"] for c in A.plan_comm_schedule(8, **link(0.0), **dev)) >= 3
    A.close()


def test_refused_coarsest_level_extends_the_hierarchy_host_side():
    """ADVICE r2 (host half; the device half is tests/test_gpu_coarse.py::test_refused_coarsest_level_extends_the_hierarchy): with the
    default parameters a 200 000-row random sparsity pattern leaves 6 250 rows after 6 levels... below dense_limit, so force the issue
    with dense_limit = 2000: the direct solver's plan refuses a graph without separators (factors would exceed a quarter of the dense
    inverse), and the setup answers by extending the hierarchy with the reference's own coarsening rule instead of failing."""
    rp, ci, v = problems.random_spd(400000, 9, seed=5)
    for form in ("nd", "bt"):
        A = sa.sp_matrix_mg(rp, ci, v).set_coarse_form(form).setup(sa.default_params(print_setup=0, print_solve=0, dense_limit=4000), host_only=True)
        info = A.coarse_info()
        assert A.nlevels > 6 and info["dense"] and info["extended"] and info["rows"] <= 4000, (form, A.nlevels, info)
        A.close()
    # a graph WITH separators of the same size is accepted as it is: 6 levels, no dense inverse
    rp, ci, v = problems.poisson2d(632)   # 399 424 rows -> 12 482 after 5 halvings
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(print_setup=0, print_solve=0, dense_limit=4000), host_only=True)
    info = A.coarse_info()
    assert A.nlevels == 6 and not info["dense"] and not info["extended"] and info["rows"] > 4000
    A.close()
