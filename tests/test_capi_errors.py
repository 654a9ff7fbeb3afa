"""Error behaviour of the C ABI and the readers (CPU only, no kernel runs)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import sparsh_amg_amd as sa
from sparsh_amg_amd import problems
from conftest import ROOT

LIB_DIR = os.path.join(ROOT, "sparsh_amg_amd")


def test_create_rejects_bad_arguments():
    rp = np.array([0, 1, 2], dtype=np.int32)
    ci = np.array([0, 1], dtype=np.int32)
    v = np.array([1.0, 1.0])
    h = C.c_void_p()
    ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
    assert sa.lib.sparsh_create_csr(0, 2, rp.ctypes.data_as(ip), ci.ctypes.data_as(ip), v.ctypes.data_as(dp), C.byref(h)) == sa.SPARSH_EINVAL
    bad = np.array([1, 1, 2], dtype=np.int32)  # rowptr[0] != 0
    assert sa.lib.sparsh_create_csr(2, 2, bad.ctypes.data_as(ip), ci.ctypes.data_as(ip), v.ctypes.data_as(dp), C.byref(h)) == sa.SPARSH_EINVAL
    assert b"rowptr" in sa.lib.sparsh_last_error()
    assert sa.lib.sparsh_create_csr(2, 2, rp.ctypes.data_as(ip), None, None, C.byref(h)) == sa.SPARSH_EINVAL


def test_state_and_range_errors():
    rp, ci, v = problems.poisson2d(80)
    A = sa.sp_matrix_mg(rp, ci, v)
    with pytest.raises(sa.SparshError) as e:
        A.level_info(0)  # before any setup
    assert e.value.code == sa.SPARSH_ESTATE
    A.setup(sa.default_params(print_setup=0), host_only=True)
    with pytest.raises(sa.SparshError) as e:
        A.level_info(99)
    assert e.value.code == sa.SPARSH_EINVAL
    with pytest.raises(sa.SparshError) as e:
        A.setup(sa.default_params(print_setup=0, sweeps=0), host_only=True)
    assert e.value.code == sa.SPARSH_EINVAL
    with pytest.raises(sa.SparshError) as e:
        A.set_kernel_config(kind=7)
    assert e.value.code == sa.SPARSH_EINVAL
    A.set_kernel_config()


def test_singular_coarse_matrix_is_reported():
    import scipy.sparse as sp

    n = 300
    M = sp.diags([np.ones(n - 1), np.ones(n - 1)], [-1, 1]).tolil()
    M[0, :] = 0  # empty row: singular
    M = M.tocsr()
    M.sort_indices()
    A = sa.sp_matrix_mg(M.indptr, M.indices, M.data)
    with pytest.raises(sa.SparshError) as e:
        A.setup(sa.default_params(print_setup=0), host_only=True)
    assert e.value.code == sa.SPARSH_ENUMERIC and "singular" in str(e.value)


def test_default_params_follow_env(monkeypatch):
    monkeypatch.setenv("SPARSH_NU", "6")
    monkeypatch.setenv("SPARSH_COARSENING", "beck")
    monkeypatch.setenv("SPARSH_TOL", "1e-6")
    p = sa.default_params()
    assert p.sweeps == 6 and p.coarsening == 1 and p.tol == 1e-6
    monkeypatch.delenv("SPARSH_NU")
    monkeypatch.delenv("SPARSH_COARSENING")
    monkeypatch.delenv("SPARSH_TOL")
    p = sa.default_params()
    # reference macros (include/AMG.hpp:15-27); 7 sweeps = smooth_iter + 1 of the CPU path
    assert (p.omega, p.tol, p.sweeps, p.max_levels, p.limit_upper, p.limit_lower, p.coarsening) == (0.66667, 1e-8, 7, 6, 4000, 2000, 0)


def test_read_coo_new_format(tmp_path):
    """The second reader of the reference (banner + % comments, 0-based triplets, rhs appended)."""
    rp, ci, v = problems.poisson2d(9)
    n = len(rp) - 1
    b = np.linspace(1, 2, n)
    f = tmp_path / "m.mtx"
    rows = np.repeat(np.arange(n), np.diff(rp))
    with open(f, "w") as o:
        o.write("%%MatrixMarket matrix coordinate real general\n% a comment\n% another\n")
        o.write(f"{n} {n} {len(ci)}\n")
        for r, c, val in zip(rows, ci, v):
            o.write(f"{r} {c} {val:.17g}\n")
        for val in b:
            o.write(f"{val:.17g}\n")
    drv = tmp_path / "drv.cpp"
    drv.write_text(
        '#include "AMG.hpp"\n#include <cstdio>\n'
        "int main(int, char** argv){ sp_matrix_mg* A; double* b; read_coo_new_format(argv[1], A, b);\n"
        ' std::printf("%d %d %d %.17g %d %.17g\\n", A->nrow, A->rowptr[A->nrow], A->colindex[5], A->val[5], A->rowptr[3], b[7]);\n'
        " return 0; }\n"
    )
    exe = tmp_path / "drv"
    cmd = ["g++", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", str(drv), "-o", str(exe), f"-L{LIB_DIR}", "-lsparsh_amg",
           f"-Wl,-rpath,{LIB_DIR}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib/llvm/lib", "-L/opt/rocm/lib", "-L/opt/rocm/lib/llvm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.check_output([str(exe), str(f)], text=True).split()
    assert int(out[0]) == n and int(out[1]) == len(ci) and int(out[2]) == ci[5] and float(out[3]) == v[5]
    assert int(out[4]) == rp[3] and float(out[5]) == b[7]


def test_matrix_market_reader(tmp_path):
    import scipy.io
    import scipy.sparse as sp

    rp, ci, v = problems.poisson2d(7)
    S = sp.csr_matrix((v, ci, rp))
    f = str(tmp_path / "sym.mtx")
    scipy.io.mmwrite(f, sp.tril(S), symmetry="symmetric")  # 1-based, lower triangle only
    rp2, ci2, v2 = problems.read_matrix_market(f)
    assert np.array_equal(rp2, rp) and np.array_equal(ci2, ci) and np.array_equal(v2, v)


def test_create_csr_validates_its_input():
    """One O(nnz) pass at create time: a non-monotone rowptr or an out-of-range column must be refused
    (SPARSH_EINVAL) instead of becoming an out-of-bounds access on the device."""
    rp, ci, v = problems.poisson2d(12)
    bad = ci.copy()
    bad[17] = len(rp) - 1  # == ncol: one past the last column
    with pytest.raises(sa.SparshError) as e:
        sa.sp_matrix_mg(rp, bad, v)
    assert e.value.code == sa.SPARSH_EINVAL and "colindex" in str(e.value)
    bad[17] = -3
    with pytest.raises(sa.SparshError) as e:
        sa.sp_matrix_mg(rp, bad, v)
    assert e.value.code == sa.SPARSH_EINVAL
    rp2 = rp.copy()
    rp2[5], rp2[6] = rp2[6], rp2[5] - 1  # decreasing
    with pytest.raises(sa.SparshError) as e:
        sa.sp_matrix_mg(rp2, ci, v)
    assert e.value.code == sa.SPARSH_EINVAL and "rowptr" in str(e.value)
    sa.sp_matrix_mg(rp, ci, v).close()  # the untouched arrays pass
