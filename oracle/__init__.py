"""ctypes binding of the CPU oracle (oracle/libamg_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.  Nothing under ``sparsh_amg_amd/`` imports this.
Parity unpinned: see the header of oracle/amg_oracle.h and DESIGN.md section 2.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libamg_oracle.so")

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)


class OCsr(C.Structure):
    _fields_ = [
        ("nrow", C.c_int),
        ("ncol", C.c_int),
        ("nnz", C.c_int),
        ("rowptr", c_int_p),
        ("col", c_int_p),
        ("val", c_dbl_p),
        ("diag", c_dbl_p),
        ("helper", c_dbl_p),
    ]


class OParams(C.Structure):
    _fields_ = [
        ("threads", C.c_int),
        ("omega", C.c_double),
        ("tol", C.c_double),
        ("limit_upper", C.c_int),
        ("limit_lower", C.c_int),
        ("max_levels", C.c_int),
        ("smooth_iter", C.c_int),
        ("coarsening", C.c_int),
        ("max_iter", C.c_int),
    ]


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc); returns the library path."""
    src = os.path.join(_HERE, "amg_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libamg_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    P = C.POINTER
    L.oracle_default_params.argtypes = [P(OParams)]
    L.oracle_csr_from.restype = P(OCsr)
    L.oracle_csr_from.argtypes = [C.c_int, C.c_int, c_int_p, c_int_p, c_dbl_p]
    L.oracle_csr_free.argtypes = [P(OCsr)]
    L.oracle_fill_diagonal.argtypes = [P(OCsr)]
    L.oracle_sort_columns.argtypes = [P(OCsr)]
    L.oracle_readcoo.restype = C.c_int
    L.oracle_readcoo.argtypes = [C.c_char_p, C.c_char_p, P(P(OCsr)), P(c_dbl_p)]
    L.oracle_set_threads.argtypes = [C.c_int]
    L.oracle_spmv.argtypes = [P(OCsr), c_dbl_p, c_dbl_p]
    L.oracle_spmv_t.argtypes = [P(OCsr), c_dbl_p, c_dbl_p]
    L.oracle_jacobi.argtypes = [P(OCsr), c_dbl_p, c_dbl_p, C.c_int, C.c_double]
    L.oracle_residual.restype = C.c_double
    L.oracle_residual.argtypes = [P(OCsr), c_dbl_p, c_dbl_p]
    L.oracle_store_residual.argtypes = [P(OCsr), c_dbl_p, c_dbl_p, c_dbl_p]
    L.oracle_transfer_residual.argtypes = [P(OCsr), c_dbl_p, c_dbl_p]
    L.oracle_transfer_solution.argtypes = [P(OCsr), c_dbl_p, c_dbl_p]
    L.oracle_dot.restype = C.c_double
    L.oracle_dot.argtypes = [C.c_int, c_dbl_p, c_dbl_p]
    L.oracle_nrm2.restype = C.c_double
    L.oracle_nrm2.argtypes = [C.c_int, c_dbl_p]
    L.oracle_hem_prolongator.restype = P(OCsr)
    L.oracle_hem_prolongator.argtypes = [P(OCsr), C.c_int]
    L.oracle_beck_prolongator.restype = P(OCsr)
    L.oracle_beck_prolongator.argtypes = [P(OCsr)]
    L.oracle_transpose.restype = P(OCsr)
    L.oracle_transpose.argtypes = [P(OCsr)]
    L.oracle_spgemm.restype = P(OCsr)
    L.oracle_spgemm.argtypes = [P(OCsr), P(OCsr)]
    L.oracle_coarsen_matrix.restype = P(OCsr)
    L.oracle_coarsen_matrix.argtypes = [P(OCsr), P(OCsr)]
    L.oracle_amg_setup.restype = C.c_void_p
    L.oracle_amg_setup.argtypes = [P(OCsr), P(OParams)]
    L.oracle_amg_free.argtypes = [C.c_void_p]
    L.oracle_amg_levels.restype = C.c_int
    L.oracle_amg_levels.argtypes = [C.c_void_p]
    L.oracle_amg_A.restype = P(OCsr)
    L.oracle_amg_A.argtypes = [C.c_void_p, C.c_int]
    L.oracle_amg_P.restype = P(OCsr)
    L.oracle_amg_P.argtypes = [C.c_void_p, C.c_int]
    L.oracle_coarse_solve.argtypes = [C.c_void_p, c_dbl_p, c_dbl_p]
    L.oracle_amg_solve.restype = C.c_int
    L.oracle_amg_solve.argtypes = [C.c_void_p, c_dbl_p, c_dbl_p, C.c_int, c_dbl_p, C.c_int]
    for name in ("oracle_solver_amg", "oracle_solver_cg", "oracle_solver_pcg", "oracle_solver_bicg", "oracle_solver_pbicg"):
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = [P(OCsr), c_dbl_p, c_dbl_p, P(OParams), c_dbl_p, C.c_int]
    L.oracle_vcycle_f32.argtypes = [C.c_void_p, c_dbl_p, c_dbl_p]
    L.oracle_time_spmv.restype = C.c_double
    L.oracle_time_spmv.argtypes = [P(OCsr), c_dbl_p, c_dbl_p, C.c_int, C.c_int]
    L.oracle_stream_triad.restype = C.c_double
    L.oracle_stream_triad.argtypes = [C.c_long, C.c_int, C.c_int]
    L.oracle_pcg_presetup.restype = C.c_int
    L.oracle_pcg_presetup.argtypes = [C.c_void_p, c_dbl_p, c_dbl_p, C.c_int, c_dbl_p, C.c_int, c_dbl_p]
    _lib = L
    return L


def _dp(a: np.ndarray):
    return a.ctypes.data_as(c_dbl_p)


def _ip(a: np.ndarray):
    return a.ctypes.data_as(c_int_p)


def params(**kw) -> OParams:
    p = OParams()
    lib().oracle_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


class Csr:
    """Owning handle on an oracle-side CSR copy of (rowptr, col, val)."""

    def __init__(self, rowptr, col, val, ncol=None, ptr=None, own=True):
        L = lib()
        if ptr is not None:
            self.ptr = ptr
        else:
            rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
            col = np.ascontiguousarray(col, dtype=np.int32)
            val = np.ascontiguousarray(val, dtype=np.float64)
            n = len(rowptr) - 1
            self.ptr = L.oracle_csr_from(n, n if ncol is None else ncol, _ip(rowptr), _ip(col), _dp(val))
            L.oracle_fill_diagonal(self.ptr)
        self.own = own

    @classmethod
    def wrap(cls, ptr, own):
        return cls(None, None, None, ptr=ptr, own=own)

    @property
    def shape(self):
        s = self.ptr.contents
        return (s.nrow, s.ncol)

    @property
    def nnz(self):
        s = self.ptr.contents
        return s.rowptr[s.nrow]

    def arrays(self):
        s = self.ptr.contents
        nnz = s.rowptr[s.nrow]
        rp = np.ctypeslib.as_array(s.rowptr, shape=(s.nrow + 1,)).copy()
        ci = np.ctypeslib.as_array(s.col, shape=(max(nnz, 1),))[:nnz].copy()
        v = np.ctypeslib.as_array(s.val, shape=(max(nnz, 1),))[:nnz].copy()
        return rp, ci, v

    def to_scipy(self):
        import scipy.sparse as sp

        rp, ci, v = self.arrays()
        return sp.csr_matrix((v, ci, rp), shape=self.shape)

    def __del__(self):
        if getattr(self, "own", False) and getattr(self, "ptr", None):
            try:
                lib().oracle_csr_free(self.ptr)
            except Exception:
                pass
            self.ptr = None


def readcoo(matrixfile: str, rhsfile: str):
    L = lib()
    A = C.POINTER(OCsr)()
    b = c_dbl_p()
    rc = L.oracle_readcoo(matrixfile.encode(), rhsfile.encode(), C.byref(A), C.byref(b))
    if rc != 0:
        raise IOError(f"oracle_readcoo failed rc={rc}")
    n = A.contents.nrow
    bv = np.ctypeslib.as_array(b, shape=(n,)).copy()
    L.oracle_fill_diagonal(A)
    return Csr.wrap(A, own=True), bv


def spmv(A: Csr, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(A.shape[0])
    lib().oracle_spmv(A.ptr, _dp(x), _dp(y))
    return y


def time_spmv(A: Csr, reps=10, threads=0):
    """Average seconds of one CPU y = A x (cpu_baseline leg of bench.py)."""
    x = np.ones(A.shape[1])
    y = np.empty(A.shape[0])
    return lib().oracle_time_spmv(A.ptr, _dp(x), _dp(y), reps, threads)


def stream_triad(n=1 << 26, reps=5, threads=0):
    """Host STREAM-triad rate in GB/s (cpu_baseline leg of bench.py)."""
    return lib().oracle_stream_triad(n, reps, threads)


def spmv_t(A: Csr, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty(A.shape[1])
    lib().oracle_spmv_t(A.ptr, _dp(x), _dp(y))
    return y


def jacobi(A: Csr, b, x, iteration=6, omega=0.66667):
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.array(x, dtype=np.float64)
    lib().oracle_jacobi(A.ptr, _dp(b), _dp(x), iteration, omega)
    return x


def residual(A: Csr, b, x):
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().oracle_residual(A.ptr, _dp(b), _dp(x))


def store_residual(A: Csr, b, x):
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    r = np.empty(A.shape[0])
    lib().oracle_store_residual(A.ptr, _dp(b), _dp(x), _dp(r))
    return r


def transfer_residual(P: Csr, r):
    return spmv_t(P, r)


def transfer_solution(P: Csr, xc, xf):
    xc = np.ascontiguousarray(xc, dtype=np.float64)
    xf = np.array(xf, dtype=np.float64)
    lib().oracle_transfer_solution(P.ptr, _dp(xc), _dp(xf))
    return xf


def dot(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    return lib().oracle_dot(len(x), _dp(x), _dp(y))


def nrm2(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().oracle_nrm2(len(x), _dp(x))


class Hierarchy:
    """oracle_amg_setup handle (AMG_solver::AMG_solver_setup_jacobi)."""

    def __init__(self, A: Csr, prm: OParams | None = None):
        self.A0 = A  # keep alive: level 0 aliases the caller's matrix
        self.prm = prm if prm is not None else params()
        self.ptr = lib().oracle_amg_setup(A.ptr, C.byref(self.prm))

    @property
    def nlevels(self):
        return lib().oracle_amg_levels(self.ptr)

    def A(self, level) -> Csr:
        return Csr.wrap(lib().oracle_amg_A(self.ptr, level), own=False)

    def P(self, level) -> Csr:
        return Csr.wrap(lib().oracle_amg_P(self.ptr, level), own=False)

    def coarse_solve(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.empty_like(b)
        lib().oracle_coarse_solve(self.ptr, _dp(b), _dp(x))
        return x

    def solve(self, b, x0=None, iterations=-1, hist_cap=4096):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
        hist = np.zeros(hist_cap)
        c = lib().oracle_amg_solve(self.ptr, _dp(b), _dp(x), iterations, _dp(hist), hist_cap)
        return x, hist[: min(c, hist_cap)].copy()

    def vcycle_f32(self, r):
        """z = V32(r): the float restatement of one V-cycle from a zero guess (fp32-preconditioner checker)."""
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.empty_like(r)
        lib().oracle_vcycle_f32(self.ptr, _dp(r), _dp(z))
        return z

    def pcg(self, b, x0=None, max_it=1 << 30, hist_cap=4096):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
        hist = np.zeros(hist_cap)
        sec = C.c_double(0.0)
        c = lib().oracle_pcg_presetup(self.ptr, _dp(b), _dp(x), max_it, _dp(hist), hist_cap, C.byref(sec))
        return x, hist[: min(c, hist_cap)].copy(), sec.value

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                lib().oracle_amg_free(self.ptr)
            except Exception:
                pass
            self.ptr = None


def solve(method: str, A: Csr, b, x0=None, prm: OParams | None = None, hist_cap=8192):
    """method in {amg, cg, pcg, bicg, pbicg} -> (x, residual history)."""
    fn = getattr(lib(), "oracle_solver_" + method)
    prm = prm if prm is not None else params()
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
    hist = np.zeros(hist_cap)
    c = fn(A.ptr, _dp(b), _dp(x), C.byref(prm), _dp(hist), hist_cap)
    return x, hist[: min(c, hist_cap)].copy()
