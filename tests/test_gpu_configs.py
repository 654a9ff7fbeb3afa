"""BASELINE.json configurations on the GPU: the C++ drop-in entry points end to end, the 2D / 3D /
unstructured configs against the oracle at sizes the oracle finishes in seconds, and the full-size
10 M-row config through size-independent properties."""
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
import sparsh_amg_amd as sa
from sparsh_amg_amd import problems
from conftest import ROOT, hist_tolerance

pytestmark = pytest.mark.gpu
QUIET = dict(print_setup=0, print_solve=0)
LIB_DIR = os.path.join(ROOT, "sparsh_amg_amd")


def _hist_ok(h, ho):
    assert len(h) == len(ho), (len(h), len(ho))
    err = np.abs(h - ho) / ho
    assert np.all(err <= hist_tolerance(ho)), f"max rel err {err.max():.3e} at {err.argmax()}"


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("drv")
    exe = d / "dropin_driver"
    cmd = ["g++", "-std=c++17", "-O1", f"-I{os.path.join(ROOT, 'include')}", os.path.join(ROOT, "tests", "cpp", "dropin_driver.cpp"),
           "-o", str(exe), f"-L{LIB_DIR}", "-lsparsh_amg", f"-Wl,-rpath,{LIB_DIR}", "-L/opt/rocm/lib", "-L/opt/rocm/lib/llvm/lib",
           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib/llvm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    rp, ci, v = problems.poisson2d(120)
    b = np.ones(len(rp) - 1)
    mf, rf = str(d / "m.txt"), str(d / "r.txt")
    problems.write_coo(mf, rf, rp, ci, v, b)
    return str(exe), mf, rf, (rp, ci, v, b)


@pytest.mark.parametrize("entry,method", [("mi", "amg"), ("ci", "amg"), ("cpu", "amg"), ("pcg1", "pcg"), ("pcg4", "pcg"),
                                          ("pbicg1", "pbicg"), ("pbicg4", "pbicg"), ("cg2", "cg"), ("bicg1", "bicg")])
def test_cpp_entry_points_on_device(driver, entry, method):
    exe, mf, rf, (rp, ci, v, b) = driver
    r = subprocess.run([exe, mf, rf, entry], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"RESULT (\S+) residual (\S+) x0 (\S+)", r.stdout)
    assert m, r.stdout[-2000:]
    assert float(m.group(2)) <= 1.001e-8
    xo, ho = oracle.solve(method, oracle.Csr(rp, ci, v), b)
    assert abs(float(m.group(3)) - xo[0]) <= 1e-9 * abs(xo[0])
    out = r.stdout
    if method == "amg":
        # reference's stdout format: "Level k:\t<n>" lines and "<cycle> <residual>" lines
        assert re.search(r"^Level 0:\t14400$", out, re.M) and re.search(r"^1 \d", out, re.M)
        assert len(re.findall(r"^\d+ [0-9.e+-]+$", out, re.M)) == len(ho)
    elif method in ("pcg", "cg"):
        assert len(re.findall(r"^\d+\t[0-9.e+-]+$", out, re.M)) == len(ho)


def test_solver_objects_api(driver, tmp_path):
    """AMG_solver / AMG_GPU1_solver / AMG_GPU_solver (solver-object layer of the reference API)."""
    _, mf, rf, (rp, ci, v, b) = driver
    exe = tmp_path / "solver_objects"
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", f"-I{os.path.join(ROOT, 'include')}", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "cpp", "solver_objects.cpp"), "-o", str(exe), f"-L{LIB_DIR}", "-lsparsh_amg", "-L/opt/rocm/lib",
           "-lamdhip64", f"-Wl,-rpath,{LIB_DIR}", "-L/opt/rocm/lib/llvm/lib", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib/llvm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), mf, rf], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = r.stdout
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    x = np.zeros(A.nrow)
    A.vcycle(b, x, iterations=3)
    lv = re.search(r"LEVELS (\d+)((?: \d+:\d+)+) P0 (\d+)x(\d+)", out)
    assert lv and int(lv.group(1)) == A.nlevels
    sizes = [tuple(int(t) for t in s.split(":")) for s in lv.group(2).split()]
    assert sizes == [(A.level_info(l)["nrow"], A.level_info(l)["nnz"]) for l in range(A.nlevels)]
    assert (int(lv.group(3)), int(lv.group(4))) == (A.nrow, A.level_info(0)["p_ncol"])
    for tag in ("HOST3", "HELPER3", "DEVICE3"):
        m = re.search(tag + r" (\S+) (\S+)", out)
        assert m, out[-1500:]
        assert float(m.group(1)) == x[0] and abs(float(m.group(2)) - np.linalg.norm(x)) <= 1e-12 * np.linalg.norm(x)
    m = re.search(r"CI_SOLVE (\S+)", out)
    assert m and float(m.group(1)) <= 1.001e-8


def test_gpu_matrix_object_api(driver, tmp_path):
    """sp_matrix_gpu::smooth_jacobi and residual() (device-operator layer of the reference API)
    against the oracle's Jacobi / residual on the same operator."""
    _, mf, rf, (rp, ci, v, b) = driver
    exe = tmp_path / "gpu_matrix_objects"
    cmd = ["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", f"-I{os.path.join(ROOT, 'include')}", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "cpp", "gpu_matrix_objects.cpp"), "-o", str(exe), f"-L{LIB_DIR}", "-lsparsh_amg", "-L/opt/rocm/lib",
           "-lamdhip64", f"-Wl,-rpath,{LIB_DIR}", "-L/opt/rocm/lib/llvm/lib", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib/llvm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), mf, rf], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    m = re.search(r"GPUMAT (\S+) (\S+) (\S+) (\S+) (\S+)", r.stdout)
    assert m, r.stdout[-1500:]
    O = oracle.Csr(rp, ci, v)
    n = len(rp) - 1
    x = 0.001 * (np.arange(n) % 17) - 0.003
    r0 = oracle.residual(O, b, x)
    x5 = oracle.jacobi(O, b, x, 4)      # iteration+1 = 5 sweeps
    r5 = oracle.residual(O, b, x5)
    x11 = oracle.jacobi(O, b, x5, 5)    # 6 more
    r11 = oracle.residual(O, b, x11)
    got = [float(m.group(k)) for k in range(1, 6)]
    assert abs(got[0] - r0) <= 1e-12 * r0 and abs(got[1] - r5) <= 1e-12 * r5 and abs(got[2] - r11) <= 1e-12 * r11
    assert got[3] == x11[0] and got[4] == x11[n // 2]  # sweeps are bitwise the oracle's


def test_building_blocks_api(tmp_path):
    """jacobi_smoother / residual / store_residual / transfer_* / coarsen_matrix / HEM_Prolongator /
    beck_prolongator / Direct_Solver_Pardiso under the reference's names: a hand-composed two-grid
    cycle against the same composition of oracle primitives."""
    rp, ci, v = problems.poisson2d(48)
    n = len(rp) - 1
    b = 1.0 + 0.01 * (np.arange(n) % 13)
    mf, rf = str(tmp_path / "m.txt"), str(tmp_path / "r.txt")
    problems.write_coo(mf, rf, rp, ci, v, b)
    exe = tmp_path / "building_blocks"
    cmd = ["g++", "-std=c++17", "-O1", f"-I{os.path.join(ROOT, 'include')}", os.path.join(ROOT, "tests", "cpp", "building_blocks.cpp"),
           "-o", str(exe), f"-L{LIB_DIR}", "-lsparsh_amg", f"-Wl,-rpath,{LIB_DIR}", "-L/opt/rocm/lib", "-L/opt/rocm/lib/llvm/lib",
           "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib/llvm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([str(exe), mf, rf], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "mis_prolongator" in r.stderr
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O, oracle.params(max_levels=2, limit_upper=100, limit_lower=10))
    P, Ac = H.P(0), H.A(1)
    Hb = oracle.Hierarchy(O, oracle.params(max_levels=2, limit_upper=100, limit_lower=10, coarsening=1))
    Pb = Hb.P(0)
    sh = [int(t) for t in re.search(r"SHAPES ((?:\d+ ?)+)", r.stdout).group(1).split()]
    assert sh == [P.shape[0], P.shape[1], P.nnz, Ac.shape[0], Ac.nnz, Pb.shape[1], Pb.nnz]
    _, _, acv = Ac.arrays()
    m = re.search(r"ACSUM (\S+) (\S+)", r.stdout)
    assert float(m.group(1)) == float(np.sum(np.abs(acv)))   # same values, summed in the same order
    x = np.zeros(n)
    r0 = oracle.residual(O, b, x)
    x = oracle.jacobi(O, b, x, 6)
    res = oracle.store_residual(O, b, x)
    bc = oracle.transfer_residual(P, res)
    xc = H.coarse_solve(bc)
    x = oracle.transfer_solution(P, xc, x)
    x = oracle.jacobi(O, b, x, 6)
    r1 = oracle.residual(O, b, x)
    g = [float(t) for t in re.search(r"TWOGRID (\S+) (\S+) (\S+) (\S+) (\S+) (\S+)", r.stdout).groups()]
    assert abs(g[0] - r0) <= 1e-12 * r0 and g[4] == bc[0]          # restriction is bitwise the oracle's
    # the coarse solve is an explicit inverse here and a banded LU in the oracle: rounding-level differences
    assert abs(g[5] - xc[len(xc) // 2]) <= 1e-10 * abs(xc[len(xc) // 2])
    assert abs(g[1] - r1) <= 1e-8 * r1 and abs(g[2] - x[0]) <= 1e-10 * abs(x[0]) and abs(g[3] - x[n // 2]) <= 1e-10 * abs(x[n // 2])


def test_sor_entry_point_is_a_stub(driver):
    exe, mf, rf, _ = driver
    r = subprocess.run([exe, mf, rf, "sor"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "not part of the MI355X build" in r.stderr


def test_config1_2d_poisson_pcg_vs_oracle():
    """configs[1]: 5-pt 2D Poisson (here 500^2 = 250k rows so the oracle takes seconds)."""
    rp, ci, v = problems.poisson2d(500)
    n = len(rp) - 1
    b = np.ones(n)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    x = np.zeros(n)
    h, rc = A.solve("pcg", b, x)
    xo, ho = oracle.solve("pcg", oracle.Csr(rp, ci, v), b, prm=oracle.params(threads=sa.host_cpus()))
    assert rc == 0
    _hist_ok(h, ho)
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)


def test_config4_unstructured_pbicgstab_vs_oracle():
    """configs[4]: irregular-nnz SPD matrix (P1-FEM M + dt K on a Delaunay mesh, the offline stand-in
    for SuiteSparse parabolic_fem), AMG-BiCGStab."""
    rp, ci, v = problems.fem_unstructured(60000, seed=7, ordering="random")  # worst-case gather order
    n = len(rp) - 1
    lens = np.diff(rp)
    assert lens.min() <= 4 and lens.max() >= 10  # genuinely ragged
    rng = np.random.default_rng(4)
    b = rng.standard_normal(n) * 1e-3
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    O = oracle.Csr(rp, ci, v)
    S = O.to_scipy()
    x = np.zeros(n)
    h, rc = A.solve("pbicg", b, x)
    xo, ho = oracle.solve("pbicg", O, b)
    assert rc == 0
    # BiCGStab on this matrix creeps from 1e-7 to 1e-8 non-monotonically; that tail amplifies
    # rounding differences (the reference's own un-preconditioned BiCGStab shows the same,
    # SURVEY Appendix A.1), so: strict on the head, loose on length, strict on the answer.
    # (measured: the device/oracle gap starts at 1e-14 relative and grows ~10x per iteration)
    k = min(8, len(h), len(ho))
    assert np.all(np.abs(h[:k] - ho[:k]) <= 1e-6 * ho[:k])
    assert abs(len(h) - len(ho)) <= max(3, len(ho) // 3)
    assert np.linalg.norm(b - S @ x) <= 1.001e-8
    assert np.linalg.norm(x - xo) <= 1e-5 * np.linalg.norm(xo)
    # the SPD solver on the same matrix is held to the strict history tolerance
    x = np.zeros(n)
    h, rc = A.solve("pcg", b, x)
    xo, ho = oracle.solve("pcg", O, b)
    assert rc == 0
    _hist_ok(h, ho)
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)
    # SpMV stress on the ragged rows, every level, bitwise
    H = oracle.Hierarchy(O)
    for l in range(A.nlevels):
        xx = rng.standard_normal(A.level_info(l)["nrow"])
        assert np.array_equal(A.op_spmv(l, xx), oracle.spmv(H.A(l), xx))


def test_config2_full_size_properties():
    """configs[2] at full size (216^3 = 10 077 696 rows): properties that need no oracle run."""
    import scipy.sparse as sp

    rp, ci, v = problems.poisson3d(216)
    n = len(rp) - 1
    assert n == 10077696 and rp[-1] == 70263936
    S = sp.csr_matrix((v, ci, rp), shape=(n, n))
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    levels = [A.level_info(l)["nrow"] for l in range(A.nlevels)]
    # the reference's 6 levels would leave 314 928 rows (above coarse_limit): the hierarchy is extended by the same coarsening rule
    # until the device direct solver can take over (<= 40 000 rows) and that level is factored by nested dissection
    assert levels == [10077696, 5038848, 2519424, 1259712, 629856, 314928, 157464, 78732, 39366]
    info = A.coarse_info()
    assert info["extended"] and info["form"] == "nested_dissection" and info["nd_launches_per_solve"] <= 19
    b = np.ones(n)
    x = np.zeros(n)
    h, rc = A.solve("pcg", b, x)
    assert rc == 0 and h[-1] <= 1e-8 and np.all(np.diff(np.log(h)) < 0.5)
    # independent residual on the host
    # CG tracks the residual by recurrence (as the reference does); at a 1e-12 reduction it may
    # differ from the true residual by O(eps * ||A|| * ||x||)
    true_r = np.linalg.norm(b - S @ x)
    assert abs(true_r - h[-1]) <= 100 * np.finfo(float).eps * 12.0 * np.linalg.norm(x)
    assert true_r <= 5e-8
    # SpMV bitwise against a scalar-order host product on a sample of rows
    rng = np.random.default_rng(0)
    xx = rng.standard_normal(n)
    y = A.op_spmv(0, xx)
    rows = rng.integers(0, n, size=2000)
    for i in rows:
        s = 0.0
        for j in range(rp[i], rp[i + 1]):
            s += v[j] * xx[ci[j]]
        assert y[i] == s
    # the V(7,7) cycle from a zero guess is linear and symmetric (SPD preconditioner)
    b1 = rng.standard_normal(n)
    b2 = rng.standard_normal(n)
    z1 = np.zeros(n)
    A.vcycle(b1, z1, iterations=1)
    z2 = np.zeros(n)
    A.vcycle(b2, z2, iterations=1)
    z12 = np.zeros(n)
    A.vcycle(2.0 * b1 - 0.5 * b2, z12, iterations=1)
    assert np.linalg.norm(z12 - (2.0 * z1 - 0.5 * z2)) <= 1e-12 * np.linalg.norm(z12)
    assert abs(z1 @ b2 - z2 @ b1) <= 1e-10 * abs(z1 @ b2)


def test_level_policy_beyond_coarse_limit():
    """What happens when level1 = 6 levels leave more than coarse_limit rows (sparsh_params.coarse_factor_mb / extend_until): 130^3
    leaves 68 657 rows.  Default: the reference's own coarsest level is KEPT, because its nested-dissection factors are estimated
    affordable (n^(4/3) growth, ~0.3 GB < 1 GB) -- the hierarchy is exactly the reference's.  coarse_factor_mb = 0: the row rule alone
    decides and the hierarchy is extended by the same coarsening rule to the first level of <= coarse_limit rows; with extend_until = 4000
    on to <= 4000 rows as in round 2.  All three are the reference's coarsening applied to the same operators, and all three converge
    to the same solution."""
    import scipy.sparse as sp

    rp, ci, v = problems.poisson3d(130)   # 2 197 000 rows
    n = len(rp) - 1
    S = sp.csr_matrix((v, ci, rp), shape=(n, n))
    b = np.ones(n)
    xs, lv, its = [], [], []
    for kw in (dict(), dict(coarse_factor_mb=0), dict(coarse_factor_mb=0, extend_until=4000)):
        A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, **kw))
        levels = [A.level_info(l)["nrow"] for l in range(A.nlevels)]
        info = A.coarse_info()
        if not kw:
            assert A.nlevels == 6 and levels[-1] == 68657 and not info["extended"] and info["form"] == "nested_dissection"
            assert info["bytes"] < 1 << 30   # the estimate that admitted the level was not optimistic
        elif "extend_until" not in kw:
            assert A.nlevels == 7 and 20000 < levels[-1] <= 40000 and info["extended"] and info["form"] == "nested_dissection" and levels[:6] == lv[0]
        else:
            assert levels[-1] <= 4000 and info["dense"] and info["extended"] and levels[:7] == lv[1]
        x = np.zeros(n)
        h, rc = A.solve("pcg", b, x)
        assert rc == 0 and np.linalg.norm(b - S @ x) <= 5e-8
        xs.append(x)
        lv.append(levels)
        its.append(len(h))
        A.close()
    assert np.linalg.norm(xs[0] - xs[1]) <= 1e-7 * np.linalg.norm(xs[0]) and np.linalg.norm(xs[0] - xs[2]) <= 1e-7 * np.linalg.norm(xs[0])
    assert its[0] <= its[1] <= its[2]   # a shallower hierarchy with an exact coarsest solve needs no more iterations


@pytest.mark.parametrize("form", ["nd", "bt"])
def test_config1_full_size_properties(form):
    """configs[1] at its stated size (5-pt 2D Poisson 1000^2 = 1 000 000 rows) with the DEFAULT, i.e. the reference's, level
    policy (/root/reference/src/AMG_phases.cpp:51,77,89): 6 levels and a 31 250-row coarsest level for the device direct solver --
    the nested-dissection factorisation (default) and the block-tridiagonal one in its interface form (window 128) --
    through properties that need no oracle run at this size."""
    import scipy.sparse as sp

    rp, ci, v = problems.poisson2d(1000)
    n = len(rp) - 1
    assert n == 1000000 and rp[-1] == 4996000
    S = sp.csr_matrix((v, ci, rp), shape=(n, n))
    A = sa.sp_matrix_mg(rp, ci, v).set_coarse_form(form).setup(sa.default_params(**QUIET))
    try:
        levels = [A.level_info(l)["nrow"] for l in range(A.nlevels)]
        assert levels == [1000000, 500000, 250000, 125000, 62500, 31250]
        info = A.coarse_info()
        assert info["rows"] == 31250 and not info["dense"] and not info["extended"]
        if form == "bt":
            assert info["form"] == "block_tridiagonal" and info["window"] == 128
        else:
            assert info["form"] == "nested_dissection" and info["nd_launches_per_solve"] <= 16 and info["bytes"] < 100e6
        b = np.ones(n)
        x = np.zeros(n)
        h, rc = A.solve("pcg", b, x)
        assert rc == 0 and h[-1] <= 1e-8 and len(h) < 60 and np.all(np.diff(np.log(h)) < 0.7)
        true_r = np.linalg.norm(b - S @ x)
        # CG tracks the residual by recurrence (as the reference does): the true one differs by O(eps ||A|| ||x||), and ||x|| ~ 7e7 here
        # (b = 1 on a 1000^2 Laplacian): measured 5.5e-8 against 7.0e-9 on the recurrence
        assert abs(true_r - h[-1]) <= 100 * np.finfo(float).eps * 8.0 * np.linalg.norm(x)
        assert true_r <= 2e-7
        # stand-alone AMG V(7,7) cycles (AMG_Solver_CPU_baseline) converge too
        xa = np.zeros(n)
        ha, rca = A.solve("amg", b, xa)
        assert rca == 0 and ha[-1] <= 1e-8 and np.linalg.norm(b - S @ xa) <= 2e-7
        # the coarsest-level solve itself: residual of A_L x = b_L through the level's own SpMV
        L = A.nlevels - 1
        rng = np.random.default_rng(0)
        bl = rng.standard_normal(levels[-1])
        xl = A.op_coarse(bl)
        assert np.linalg.norm(bl - A.op_spmv(L, xl)) <= 1e-10 * np.linalg.norm(bl)
        # SpMV bitwise against a scalar-order host product on a sample of rows
        xx = rng.standard_normal(n)
        y = A.op_spmv(0, xx)
        for i in rng.integers(0, n, size=2000):
            sacc = 0.0
            for j in range(rp[i], rp[i + 1]):
                sacc += v[j] * xx[ci[j]]
            assert y[i] == sacc
        # the V(7,7) cycle from a zero guess is linear and symmetric (SPD preconditioner)
        b1 = rng.standard_normal(n)
        b2 = rng.standard_normal(n)
        z1 = np.zeros(n)
        A.vcycle(b1, z1, iterations=1)
        z2 = np.zeros(n)
        A.vcycle(b2, z2, iterations=1)
        z12 = np.zeros(n)
        A.vcycle(2.0 * b1 - 0.5 * b2, z12, iterations=1)
        assert np.linalg.norm(z12 - (2.0 * z1 - 0.5 * z2)) <= 1e-12 * np.linalg.norm(z12)
        assert abs(z1 @ b2 - z2 @ b1) <= 1e-10 * abs(z1 @ b2)
    finally:
        A.close()


@pytest.mark.parametrize("n", [1, 2, 5, 63, 64, 65, 257])
def test_tiny_matrices(n):
    """Sizes around the 64-lane wave / slice boundary and degenerate ones; single-level solves."""
    import scipy.sparse as sp

    rng = np.random.default_rng(n)
    if n == 1:
        M = sp.csr_matrix(np.array([[2.5]]))
    else:
        M = (sp.diags([-1.0 * np.ones(n - 1), 2.5 * np.ones(n), -1.0 * np.ones(n - 1)], [-1, 0, 1])).tocsr()
    M.sort_indices()
    rp, ci, v = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    O = oracle.Csr(rp, ci, v)
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    assert np.array_equal(A.op_spmv(0, x), oracle.spmv(O, x))
    assert np.array_equal(A.op_jacobi(0, b, x, 3), oracle.jacobi(O, b, x, 2))
    for method in ("amg", "pcg", "pbicg", "cg", "bicg"):
        xs = np.zeros(n)
        h, rc = A.solve(method, b, xs)
        assert rc == 0, method
        assert np.linalg.norm(b - M @ xs) <= 1.001e-8, method


def test_unsymmetric_convection_diffusion_bicgstab():
    """A genuinely unsymmetric operator (upwind convection-diffusion): the reference treats every
    matrix as general (PARDISO mtype 11, BiCGStab).  Device vs oracle."""
    import scipy.sparse as sp

    n = 120
    N = n * n
    I = sp.identity(n)
    T = sp.diags([-1.0 * np.ones(n - 1), 2.0 * np.ones(n), -1.0 * np.ones(n - 1)], [-1, 0, 1])
    Cx = sp.diags([-1.0 * np.ones(n - 1), 1.0 * np.ones(n)], [-1, 0])  # upwind d/dx
    M = (sp.kron(I, T) + sp.kron(T, I) + 0.8 * sp.kron(I, Cx) + 0.3 * sp.kron(Cx, I)).tocsr()
    M.sort_indices()
    assert abs(M - M.T).max() > 0.1
    rp, ci, v = M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(np.float64)
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O)
    rng = np.random.default_rng(1)
    for l in range(A.nlevels):  # operators of an unsymmetric hierarchy, bitwise
        nl = A.level_info(l)["nrow"]
        xx = rng.standard_normal(nl)
        bb = rng.standard_normal(nl)
        assert np.array_equal(A.op_spmv(l, xx), oracle.spmv(H.A(l), xx))
        assert np.array_equal(A.op_jacobi(l, bb, xx, 2), oracle.jacobi(H.A(l), bb, xx, 1))
    b = np.ones(N)
    for method in ("pbicg", "bicg", "amg"):
        x = np.zeros(N)
        h, rc = A.solve(method, b, x)
        xo, ho = oracle.solve(method, O, b)
        assert rc == 0
        k = min(8, len(h), len(ho))
        assert np.all(np.abs(h[:k] - ho[:k]) <= 1e-6 * ho[:k]), method
        assert abs(len(h) - len(ho)) <= max(3, len(ho) // 4), (method, len(h), len(ho))
        # BiCGStab stops on the recurrence residual (as the reference does); the true residual
        # drifts from it on this operator -- hold the device to the oracle's own drift
        true_dev, true_orc = np.linalg.norm(b - M @ x), np.linalg.norm(b - M @ xo)
        assert h[-1] <= 1e-8 and true_dev <= max(1.001e-8, 5 * true_orc), (method, true_dev, true_orc)
        assert np.linalg.norm(x - xo) <= 1e-6 * np.linalg.norm(xo)


@pytest.mark.parametrize("gen", ["p3d", "p2d"])
def test_fp32_preconditioner_mode(gen):
    """Opt-in mixed precision: float V-cycle inside the fp64 CG.  Not a parity mode --
    checked for what it promises: same solution to the same tolerance, about the same iteration
    count, fp64 residual recurrence intact."""
    rp, ci, v = problems.poisson3d(48) if gen == "p3d" else problems.poisson2d(300)
    n = len(rp) - 1
    import scipy.sparse as sp

    S = sp.csr_matrix((v, ci, rp), shape=(n, n))
    b = np.ones(n)
    A64 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    x64 = np.zeros(n)
    h64, rc = A64.solve("pcg", b, x64)
    A32 = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, precond_fp32=1))
    x32 = np.zeros(n)
    h32, rc32 = A32.solve("pcg", b, x32)
    assert rc == 0 and rc32 == 0
    assert len(h32) <= len(h64) + 3, (len(h32), len(h64))
    assert np.allclose(h32[:3], h64[:3], rtol=1e-3)  # same preconditioner up to float rounding
    assert np.linalg.norm(b - S @ x32) <= 5e-8
    assert np.linalg.norm(x32 - x64) <= 1e-7 * np.linalg.norm(x64)
    xb32 = np.zeros(n)
    hb32, rcb = A32.solve("pbicg", b, xb32)  # the mode also serves AMG-BiCGStab
    assert rcb == 0 and np.linalg.norm(b - S @ xb32) <= 5e-8 and np.linalg.norm(xb32 - x64) <= 1e-7 * np.linalg.norm(x64)
    # the fp64 entry points are untouched by the mode: AMG stand-alone still matches bitwise
    xa, xb = np.zeros(n), np.zeros(n)
    A64.vcycle(b, xa, iterations=2)
    A32.vcycle(b, xb, iterations=2)
    assert np.array_equal(xa, xb)


@pytest.mark.parametrize("name,gen", [
    ("p3d_48", lambda: problems.poisson3d(48)),
    ("p2d_300", lambda: problems.poisson2d(300)),
    ("fem_unstructured", lambda: problems.fem_unstructured(40000, seed=3)),   # no sliced-diagonal mirror anywhere: CSR float values
    ("random_spd", lambda: problems.random_spd(30000, 9, seed=11)),
])
def test_fp32_preconditioner_against_float_oracle(name, gen):
    """z = V32(r) of the device (sparsh_op_precond_f32) against oracle_vcycle_f32, the float restatement of
    the same cycle (same order of operations, every operand rounded to float).  Float tolerance: the sweeps
    agree to rounding; the coarsest solve differs in form (float inverse GEMV on the device, double banded LU
    rounded to float in the oracle), which enters at float accuracy times the coarse condition number.
    Also: the float cycle really is a float cycle (it differs from the fp64 one at the 1e-7..1e-5 level),
    and the mode no longer refuses operators without the sliced-diagonal layout."""
    rp, ci, v = gen()
    n = len(rp) - 1
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET, precond_fp32=1))
    O = oracle.Csr(rp, ci, v)
    H = oracle.Hierarchy(O)
    assert A.nlevels == H.nlevels
    rng = np.random.default_rng(12)
    for r in (np.ones(n), rng.standard_normal(n)):
        zd = A.op_precond_f32(r)
        zo = H.vcycle_f32(r)
        z64, _ = H.solve(r, iterations=1)           # fp64 V-cycle from a zero guess
        err = np.linalg.norm(zd - zo) / np.linalg.norm(zo)
        gap64 = np.linalg.norm(zo - z64) / np.linalg.norm(z64)
        assert err <= 2e-5, (name, err)
        assert 1e-9 < gap64 < 1e-3, (name, gap64)  # float rounding is visible, and only float rounding
        assert err <= 50 * gap64 + 1e-7
    # the whole mode on this operator: fp64 CG around the float cycle converges to the same tolerance
    b = np.ones(n)
    x = np.zeros(n)
    h, rc = A.solve("pcg", b, x)
    assert rc == 0 and np.linalg.norm(b - O.to_scipy() @ x) <= 5e-8
    A.close()


@pytest.fixture(scope="module")
def config4_full():
    """BASELINE configs[4] stand-in at FULL size (525 825 rows, 3.67 M entries, 3..12+ entries per row) and the
    CPU oracle's results on it (tests/golden/config4_fem_oracle.json, made by make_config4_fixture.py)."""
    import json

    with open(os.path.join(ROOT, "tests", "golden", "config4_fem_oracle.json")) as f:
        g = json.load(f)
    rp, ci, v = problems.fem_unstructured(525825)
    assert len(rp) - 1 == g["nrow"] and int(rp[-1]) == g["nnz"]
    A = sa.sp_matrix_mg(rp, ci, v).setup(sa.default_params(**QUIET))
    import scipy.sparse as sp

    yield A, sp.csr_matrix((v, ci, rp)), g
    A.close()


def test_config4_full_size_bicgstab_converges(config4_full):
    """One successful full-size AMG-BiCGStab run (random right-hand side): parity with the oracle on the head of
    the history (BiCGStab amplifies rounding ~10x per iteration afterwards), iteration count within 15 %, true
    residual below tol; AMG-PCG on the same system to the strict history tolerance on its head."""
    A, S, g = config4_full
    n = A.nrow
    assert A.level_kernel(0) in ("csr_block_kernel", "sell_kernel")  # irregular rows: no sliced-diagonal layout
    b = np.random.default_rng(4).standard_normal(n) * 1e-3
    go = g["random_rhs"]["pbicg"]
    x = np.zeros(n)
    h, rc = A.solve("pbicg", b, x)
    assert rc == 0 and go["first_nonfinite"] is None
    # measured: device/oracle gap 1e-11 relative for four iterations, then ~100x per iteration (1e-8, 1e-4, ...):
    # BiCGStab on this operator is that sensitive to rounding, so the count is loosely held
    hh = np.array(go["hist_head"])
    assert np.all(np.abs(h[:4] - hh[:4]) <= 1e-8 * hh[:4])
    assert np.all(np.abs(h[4:6] - hh[4:6]) <= 1e-3 * hh[4:6])
    assert abs(len(h) - go["iterations"]) <= 0.4 * go["iterations"], (len(h), go["iterations"])
    # The stopping rule is the reference's: on the RECURRENCE residual (src/AMG_main_solvers.cpp:437,397).  BiCGStab's recurrence drifts from
    # b - A x by rounding times the largest intermediate quantities of the run, which depend on the path taken (r3: 334 iterations, recurrence
    # 9.9e-9, true 1.22e-8; r2, other coarse factorisation, 322 iterations: both below 1e-8; the oracle's 424-iteration path: 4.572e-9 / 4.586e-9).
    # Held here: (a) the recurrence meets tol, (b) the true residual is within a factor 1.5 of it, and (c) -- the property a caller relies on --
    # a restart from the returned x (which recomputes r = b - A x) reaches tol in the TRUE residual after a handful of iterations.
    true_r = np.linalg.norm(b - S @ x)
    assert h[-1] <= 1e-8 and true_r <= 1.5e-8, (h[-1], true_r)
    x2 = x.copy()
    h2, rc2 = A.solve("pbicg", b, x2)
    assert rc2 == 0 and len(h2) <= 40 and np.linalg.norm(b - S @ x2) <= 1.001e-8, (len(h2), np.linalg.norm(b - S @ x2))  # measured: 13 iterations, 9.05e-9
    assert abs(np.linalg.norm(x) - go["xnorm"]) <= 1e-5 * go["xnorm"]
    gp = g["random_rhs"]["pcg"]
    x[:] = 0
    h, rc = A.solve("pcg", b, x)
    assert rc == 0 and abs(len(h) - gp["iterations"]) <= 2
    hh = np.array(gp["hist_head"])
    assert np.all(np.abs(h[:len(hh)] - hh) <= 1e-6 * hh)
    assert np.linalg.norm(b - S @ x) <= 1.001e-8 and abs(np.linalg.norm(x) - gp["xnorm"]) <= 1e-7 * gp["xnorm"]


def test_config4_iterates_true_residual_vs_oracle(config4_full):
    """Assertions that cannot be fitted to the device's output (VERDICT r2 item 6 i): the run is capped at exactly k = 1..6
    iterations and the TRUE residual ||b - A x_k||_2 of the device's iterate, computed on the host with scipy, is held to the same
    quantity of the oracle's iterate (fixture: tests/golden/config4_fem_oracle.json, true_residuals_k1_6) -- independent of either
    side's residual recurrence.  AMG-PCG: 1e-8 relative for all six.  AMG-PBiCGStab: 1e-8 for k <= 4; iterations 5 and 6 at 1e-3,
    because on this operator BiCGStab amplifies the rounding difference between two correct implementations ~100x per iteration
    (measured device/oracle gap 1e-11, ..., 1e-8, 1e-4; the fixture's rounding_sensitivity shows the oracle's own count moving when
    its right-hand side is perturbed by one ulp)."""
    A, S, g = config4_full
    n = A.nrow
    b = np.random.default_rng(4).standard_normal(n) * 1e-3
    try:
        for method in ("pcg", "pbicg"):
            go = g["random_rhs"][method]
            for k in range(1, 7):
                A.set_stopping(1e-8, k, 1)
                x = np.zeros(n)
                h, rc = A.solve(method, b, x, allow=(sa.SPARSH_ENOCONV,))
                assert rc == sa.SPARSH_ENOCONV and len(h) == go["history_length_at_cap_k1_6"][k - 1], (method, k, rc, len(h))
                tr = np.linalg.norm(b - S @ x)
                ref = go["true_residuals_k1_6"][k - 1]
                tol = 1e-8 if (method == "pcg" or k <= 4) else 1e-3
                assert abs(tr - ref) <= tol * ref, (method, k, tr, ref)
    finally:
        A.set_stopping(1e-8, 100000, 1)


def test_config4_kernel_families_agree_bitwise_over_six_iterations():
    """VERDICT r2 item 6 ii: the first six AMG-PBiCGStab and AMG-PCG iterations on the full-size irregular operator through the
    workgroup CSR-stream kernel (CSR-order gathers), the row-lane kernel, the wave kernel and the 16-bit-index kernel: residual
    histories and iterates equal bit for bit (same rounded products, same order of additions in every family)."""
    rp, ci, v = problems.fem_unstructured(525825)
    n = len(rp) - 1
    b = np.random.default_rng(4).standard_normal(n) * 1e-3
    ref = {}
    seen = set()
    for name, cfg, idx16 in (("default", None, None), ("csr_block", (0, 1, -1, -1), None), ("csr_rowlane", (0, 2, -1, -1), None),
                             ("csr_wave", (1, 1, -1, -1), None), ("csr_rowlane16", (0, 4, -1, -1), 2)):
        A = sa.sp_matrix_mg(rp, ci, v)
        if idx16 is not None:
            A.set_index_compression(idx16)
        A.setup(sa.default_params(**QUIET))
        if cfg:
            A.set_kernel_config(*cfg)
        seen.add(A.level_kernel(0))
        for method in ("pbicg", "pcg"):
            A.set_stopping(1e-8, 6, 1)
            x = np.zeros(n)
            h, rc = A.solve(method, b, x, allow=(sa.SPARSH_ENOCONV,))
            assert rc == sa.SPARSH_ENOCONV and np.all(np.isfinite(h))
            if name == "default":
                ref[method] = (h.copy(), x.copy())
            else:
                assert np.array_equal(h, ref[method][0]), (name, method)
                assert np.array_equal(x, ref[method][1]), (name, method)
        A.close()
    assert {"csr_block_kernel", "csr_rowlane_kernel", "csr_wave_kernel", "csr_rowlane16_kernel"} <= seen, seen


def test_config4_iteration_count_band_is_backed_by_the_oracles_own_sensitivity(config4_full):
    """VERDICT r2 item 6 iii: the loose band on the AMG-PBiCGStab iteration count is justified by data, not by the device's number.
    The fixture records (a) that the oracle's count does not depend on its thread count (chunked reductions: 1, 2, 8 threads give the
    same history), and (b) how far the oracle's OWN count moves when its right-hand side is perturbed by one rounding error per
    entry: 424 unperturbed; 298, 485 and one breakdown (NaN at 361) perturbed.  The device's count has to lie within the spread of
    the converged oracle runs widened by a quarter on either side."""
    A, S, g = config4_full
    go = g["random_rhs"]["pbicg"]
    ts = go["thread_sensitivity"]
    assert len({ts[k]["iterations"] for k in ts}) == 1
    runs = go["rounding_sensitivity"]
    # one of the three perturbed oracle runs does not even converge: it ends in 0/0 after 361 iterations (final residual NaN) --
    # BiCGStab without breakdown checks (src/AMG_main_solvers.cpp:397) on this operator is that fragile in the oracle itself
    assert any(r["final_residual"] is None or not np.isfinite(r["final_residual"]) for r in runs)
    counts = [go["iterations"]] + [r["iterations"] for r in runs if r["final_residual"] is not None and np.isfinite(r["final_residual"])]
    lo, hi = min(counts), max(counts)
    assert hi >= 1.3 * lo  # converged oracle runs alone: 298 ... 485 iterations
    b = np.random.default_rng(4).standard_normal(A.nrow) * 1e-3
    x = np.zeros(A.nrow)
    h, rc = A.solve("pbicg", b, x)
    assert rc == 0 and 0.75 * lo <= len(h) <= 1.25 * hi, (len(h), counts)
    # (recurrence residual at tol, true residual within a factor 1.5 of it: see test_config4_full_size_bicgstab_converges)
    assert h[-1] <= 1e-8 and np.linalg.norm(b - S @ x) <= 1.5e-8


def test_config4_full_size_breakdown_like_the_oracle(config4_full):
    """Constant right-hand side: the constant vector is almost an eigenvector of M + dt K, AMG-BiCGStab
    stagnates and then divides 0/0 (the reference has no breakdown checks, src/AMG_main_solvers.cpp:397).  The
    device must do what the algorithm does: same head, same stagnation level, NaN reported as SPARSH_ENUMERIC
    -- not a hang, not SPARSH_OK.  AMG-PCG on the same system converges on both."""
    A, S, g = config4_full
    n = A.nrow
    b = np.full(n, 1e-3)
    go = g["constant_rhs"]["pbicg"]
    assert go["first_nonfinite"] is not None       # the oracle breaks down
    A.set_stopping(1e-8, 8000, 1)
    x = np.zeros(n)
    h, rc = A.solve("pbicg", b, x, allow=(sa.SPARSH_ENUMERIC,))
    assert rc == sa.SPARSH_ENUMERIC, (rc, len(h))
    hh = np.array(go["hist_head"][:4])
    assert np.all(np.abs(h[:4] - hh) <= 1e-5 * hh)  # degenerate Krylov space: rounding shows from the first iteration (measured 1e-7)
    fin = h[np.isfinite(h)]
    # All stagnate orders of magnitude above tol and then hit 0/0 -- the oracle at 7.2e-3 after 1854 iterations, the device at 5.6e-3
    # after 691 (round 2, block-tridiagonal coarse factors) and at 3.4e-5 after 1795 (round 3, nested-dissection factors): where and
    # at what level rounding trips a degenerate BiCGStab is chaotic, so the level is NOT held (round 2 held it to a band fitted to
    # its own output); held: the run never reaches tol, ends in the breakdown, and says so.
    assert len(fin) > 100 and fin.min() > 1e-8 and len(fin) < len(h)
    assert len(h) < 8000
    gp = g["constant_rhs"]["pcg"]
    A.set_stopping(1e-8, 100000, 1)
    x = np.zeros(n)
    hp, rc = A.solve("pcg", b, x)
    assert rc == 0 and abs(len(hp) - gp["iterations"]) <= 2
    # |x| ~ 530 here (mass-dominated rows, b = 1e-3): the true residual sits a few 1e-8 above the recurrence's
    # (||A|| ||x|| eps drift); the stopping rule is the reference's, on the recurrence
    assert hp[-1] <= 1e-8 and np.linalg.norm(b - S @ x) <= 1e-7


def test_placement_search_changes_nothing_but_pointers():
    """Setup-time placement search (Engine::tune_placement): on a level whose three sweep vectors are about the size of the
    Infinity Cache the engine picks, by timing, which of its buffers hold iterate / twin / Krylov residual.  Pointers only:
    residual histories and solutions are bit for bit those of a handle that keeps the allocation order."""
    rp, ci, v = problems.poisson3d(168)  # 4.7 M rows: 114 MB of sweep vectors, just above the smallest size the search looks at (96 MiB)
    n = len(rp) - 1
    b = np.ones(n)
    res = []
    for search in (True, False):
        A = sa.sp_matrix_mg(rp, ci, v).set_placement_search(search).setup(sa.default_params(**QUIET))
        info = A.placement_info()
        if search:
            assert 0 < info["triples"] <= 260 and 0 < info["chosen_us"] <= info["initial_us"] <= info["worst_us"], info
        else:
            assert info["triples"] == 0
        x = np.zeros(n)
        h, rc = A.solve("pcg", b, x)
        xa = np.zeros(n)
        ha, rca = A.solve("amg", b, xa)
        res.append((h, rc, x, ha, rca, xa))
        A.close()
    for u, w in zip(res[0], res[1]):
        assert np.array_equal(u, w)
