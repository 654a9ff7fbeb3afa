"""sparsh_amg_amd -- MI355X-native AMG solve phase behind the SParSH-AMG API.

This package is a thin ctypes binding of ``libsparsh_amg.so`` (HIP kernels for gfx950 + host
setup + C ABI, sources in ``csrc/``; C ABI in ``include/sparsh_amg.h``).  There is no Python or
CPU compute path: if the native library is missing, importing fails loudly, and if no GPU is
visible every solver call raises.

Python names mirror the reference's C++ entry points (include/AMG.hpp:40-85 of
cmgcds/SParSH-AMG): ``AMG_Solver_CPU_GPU_MI(A, b, x)``, ``Solver_PCG_4(A, b, x)`` ...
where ``A`` is an :class:`sp_matrix_mg`, and ``b``/``x`` are float64 numpy vectors (``x`` is
the initial guess on entry and is overwritten with the solution, as in the reference).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import problems  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsparsh_amg.so")

SPARSH_AMG, SPARSH_CG, SPARSH_PCG, SPARSH_BICG, SPARSH_PBICG = 0, 1, 2, 3, 4
METHODS = {"amg": SPARSH_AMG, "cg": SPARSH_CG, "pcg": SPARSH_PCG, "bicg": SPARSH_BICG, "pbicg": SPARSH_PBICG}
SPARSH_OK, SPARSH_EINVAL, SPARSH_ENODEV, SPARSH_ESTATE, SPARSH_ENUMERIC, SPARSH_ENOCONV, SPARSH_ECOMM = 0, -1, -2, -3, -4, -5, -6

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)


class SparshError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sparsh error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    """``sparsh_params`` (include/sparsh_amg.h); defaults = macros of the reference's AMG.hpp:15-27."""

    _fields_ = [
        ("omega", C.c_double),
        ("tol", C.c_double),
        ("sweeps", C.c_int),
        ("max_levels", C.c_int),
        ("limit_upper", C.c_int),
        ("limit_lower", C.c_int),
        ("coarsening", C.c_int),
        ("max_iter", C.c_int),
        ("coarse_limit", C.c_int),
        ("host_threads", C.c_int),
        ("device", C.c_int),
        ("print_setup", C.c_int),
        ("print_solve", C.c_int),
        ("check_every", C.c_int),
        ("use_graph", C.c_int),
        ("replicate_rows", C.c_int),
        ("precond_fp32", C.c_int),
        ("dense_limit", C.c_int),
        ("extend_until", C.c_int),
        ("coarse_factor_mb", C.c_int),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C sparsh_amg_amd/csrc`).  There is no pure-Python/CPU fallback."
        )
    L = C.CDLL(LIB_PATH)
    H = C.c_void_p
    P = C.POINTER
    sig = {
        "sparsh_last_error": (C.c_char_p, []),
        "sparsh_version": (C.c_int, []),
        "sparsh_device_count": (C.c_int, []),
        "sparsh_host_cpus": (C.c_int, []),
        "sparsh_default_params": (None, [P(Params)]),
        "sparsh_create_csr": (C.c_int, [C.c_int, C.c_int, c_int_p, c_int_p, c_dbl_p, P(H)]),
        "sparsh_destroy": (None, [H]),
        "sparsh_setup": (C.c_int, [H, P(Params)]),
        "sparsh_setup_host": (C.c_int, [H, P(Params)]),
        "sparsh_set_stopping": (C.c_int, [H, C.c_double, C.c_int, C.c_int]),
        "sparsh_set_kernel_config": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int]),
        "sparsh_level_placement": (C.c_int, [H, C.c_int, c_int_p, c_int_p]),
        "sparsh_level_format": (C.c_int, [H, C.c_int, c_int_p, C.POINTER(C.c_long)]),
        "sparsh_level_layout": (C.c_int, [H, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "sparsh_set_const_slots": (C.c_int, [H, C.c_int]),
        "sparsh_set_tile": (C.c_int, [H, C.c_int]),
        "sparsh_set_index_compression": (C.c_int, [H, C.c_int]),
        "sparsh_set_alternate_sweeps": (C.c_int, [H, C.c_int]),
        "sparsh_set_paired_restriction": (C.c_int, [H, C.c_int]),
        "sparsh_set_fused_prolongation": (C.c_int, [H, C.c_int]),
        "sparsh_set_constant_diagonal": (C.c_int, [H, C.c_int]),
        "sparsh_set_double_sweep": (C.c_int, [H, C.c_int]),
        "sparsh_set_marching_ops": (C.c_int, [H, C.c_int]),
        "sparsh_set_zero_start": (C.c_int, [H, C.c_int]),
        "sparsh_set_deferred_x": (C.c_int, [H, C.c_int]),
        "sparsh_level_marching_ops": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "sparsh_level_double_sweep": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                                  C.POINTER(C.c_double)]),
        "sparsh_level_constant_diagonal": (C.c_int, [H, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
        "sparsh_level_prolong_fused": (C.c_int, [H, C.c_int, C.POINTER(C.c_int)]),
        "sparsh_op_jacobi_prolong": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p]),
        "sparsh_level_paired": (C.c_int, [H, C.c_int, C.POINTER(C.c_int)]),
        "sparsh_set_fused_zero_sweep": (C.c_int, [H, C.c_int]),
        "sparsh_set_placement_search": (C.c_int, [H, C.c_int]),
        "sparsh_placement_info": (C.c_int, [H, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), c_int_p, C.POINTER(C.c_double)]),
        "sparsh_debug_index16_roundtrip": (C.c_int, [C.c_int, c_int_p, c_int_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "sparsh_set_setup_broadcast": (C.c_int, [H, C.c_int]),
        "sparsh_debug_hierarchy_roundtrip": (C.c_long, [H, C.c_long]),
        "sparsh_setup_share_info": (C.c_int, [H, C.POINTER(C.c_int), C.POINTER(C.c_long)]),
        "sparsh_level_index16": (C.c_int, [H, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
        "sparsh_level_tile_rows": (C.c_int, [H, C.c_int, c_int_p]),
        "sparsh_bench_comm": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_dbl_p]),
        "sparsh_level_kernel": (C.c_char_p, [H, C.c_int]),
        "sparsh_num_levels": (C.c_int, [H]),
        "sparsh_level_info": (C.c_int, [H, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p]),
        "sparsh_level_csr": (C.c_int, [H, C.c_int, C.c_int, c_int_p, c_int_p, c_dbl_p]),
        "sparsh_coarse_inverse": (C.c_int, [H, c_dbl_p]),
        "sparsh_op_precond_f32": (C.c_int, [H, c_dbl_p, c_dbl_p]),
        "sparsh_coarse_info": (C.c_int, [H, c_int_p, C.POINTER(C.c_long)]),
        "sparsh_coarse_window": (C.c_int, [H, c_int_p]),
        "sparsh_set_coarse_interface": (C.c_int, [H, C.c_int]),
        "sparsh_set_coarse_block": (C.c_int, [H, C.c_int]),
        "sparsh_set_coarse_form": (C.c_int, [H, C.c_int, C.c_int, C.c_int]),
        "sparsh_coarse_nd_info": (C.c_int, [H, c_int_p]),
        "sparsh_set_coarse_top_merge": (C.c_int, [H, C.c_int]),
        "sparsh_setup_seconds": (C.c_double, [H]),
        "sparsh_vcycle": (C.c_int, [H, c_dbl_p, c_dbl_p, C.c_int, c_dbl_p, C.c_int, c_int_p]),
        "sparsh_vcycle_dev": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_int, c_dbl_p, C.c_int, c_int_p]),
        "sparsh_solve": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, C.c_int, c_int_p]),
        "sparsh_solve_dev": (C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p, C.c_int, c_dbl_p, C.c_int, c_int_p, c_dbl_p]),
        "sparsh_set_device": (C.c_int, [C.c_int]),
        "sparsh_comm_unique_id": (C.c_int, [C.c_char_p]),
        "sparsh_comm_init_rccl": (C.c_int, [H, C.c_char_p, C.c_int, C.c_int]),
        "sparsh_local_range": (C.c_int, [H, C.c_int, c_int_p, c_int_p, c_int_p]),
        "sparsh_set_overlap": (C.c_int, [H, C.c_int]),
        "sparsh_comm_group_create": (C.c_int, [C.c_int, P(C.c_void_p)]),
        "sparsh_comm_group_destroy": (None, [C.c_void_p]),
        "sparsh_set_deep_halo": (C.c_int, [H, C.c_int]),
        "sparsh_dist_deep_op": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p]),
        "sparsh_dist_deep_op_get": (C.c_int, [H, c_int_p, c_int_p, c_dbl_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
        "sparsh_deep_info": (C.c_int, [H, C.c_int, c_int_p]),
        "sparsh_deep_layer_end": (C.c_int, [H, C.c_int, C.c_int, c_int_p]),
        "sparsh_deep_prefix_spmv": (C.c_int, [H, C.c_int, C.c_int, c_dbl_p, c_dbl_p]),
        "sparsh_exchanges_issued": (C.c_long, [H]),
        "sparsh_comm_group_fail_after": (C.c_int, [C.c_void_p, C.c_int]),
        "sparsh_comm_group_set_delay": (C.c_int, [C.c_void_p, C.c_double]),
        "sparsh_set_comm_tuning": (C.c_int, [H, C.c_int]),
        "sparsh_comm_schedule": (C.c_int, [H, C.c_int, c_int_p, c_dbl_p]),
        "sparsh_comm_measured": (C.c_int, [H, c_dbl_p]),
        "sparsh_plan_comm_schedule": (C.c_int, [H, C.c_int, c_dbl_p]),
        "sparsh_comm_init_group": (C.c_int, [H, C.c_void_p, C.c_int]),
        "sparsh_dist_local_op": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p]),
        "sparsh_dist_local_op_get": (C.c_int, [H, c_int_p, c_int_p, c_dbl_p, c_int_p, c_int_p, c_int_p, c_int_p]),
        "sparsh_krylov_init_dev": (C.c_int, [H, C.c_int, C.c_void_p, C.c_void_p]),
        "sparsh_krylov_step_dev": (C.c_int, [H, C.c_int, c_int_p, c_dbl_p]),
        "sparsh_krylov_history": (C.c_int, [H, c_dbl_p, C.c_int, c_int_p]),
        "sparsh_op_spmv": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p]),
        "sparsh_op_jacobi": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p, C.c_int, C.c_int]),
        "sparsh_op_residual": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p]),
        "sparsh_op_resnorm": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p]),
        "sparsh_op_restrict": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p]),
        "sparsh_op_residual_restrict": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p]),
        "sparsh_op_prolong": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p]),
        "sparsh_op_coarse": (C.c_int, [H, c_dbl_p, c_dbl_p]),
        "sparsh_op_dot": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p, c_dbl_p]),
        "sparsh_op_nrm2": (C.c_int, [H, C.c_int, c_dbl_p, c_dbl_p]),
        "sparsh_op_axpby": (C.c_int, [H, C.c_int, C.c_double, c_dbl_p, C.c_double, c_dbl_p]),
        "sparsh_bench_op": (C.c_int, [H, C.c_int, C.c_int, C.c_int, c_dbl_p]),
        "sparsh_dev_alloc": (C.c_int, [H, C.c_long, P(C.c_void_p)]),
        "sparsh_dev_free": (C.c_int, [H, C.c_void_p]),
        "sparsh_dev_fill": (C.c_int, [H, C.c_void_p, C.c_long, C.c_double]),
        "sparsh_h2d": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_long]),
        "sparsh_d2h": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_long]),
        "sparsh_sync": (C.c_int, [H]),
        "sparsh_profile": (C.c_int, [H, C.c_int]),
        "sparsh_profile_read": (C.c_int, [H, c_dbl_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return L


lib = _load()


def _dp(a):
    return a.ctypes.data_as(c_dbl_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _check(rc, allow=()):
    if rc != SPARSH_OK and rc not in allow:
        raise SparshError(rc, lib.sparsh_last_error().decode())
    return rc


def default_params(**kw) -> Params:
    p = Params()
    lib.sparsh_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(f"sparsh_params has no field {k}")
        setattr(p, k, v)
    return p


def set_device(device: int):
    _check(lib.sparsh_set_device(device))


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _check(lib.sparsh_comm_unique_id(buf))
    return buf.raw


def comm_group_create(nranks: int):
    g = C.c_void_p()
    _check(lib.sparsh_comm_group_create(nranks, C.byref(g)))
    return g


def comm_group_destroy(g):
    lib.sparsh_comm_group_destroy(g)


def comm_group_set_delay(g, microseconds: float):
    """In-process test transport: every transport call first occupies the caller's stream this long (a slow link)."""
    _check(lib.sparsh_comm_group_set_delay(g, float(microseconds)))


def comm_group_fail_after(g, ncalls: int):
    """Fault injection (tests): every rank's halo exchange number `ncalls` and all later ones fail."""
    _check(lib.sparsh_comm_group_fail_after(g, int(ncalls)))


def index16_roundtrip(rowptr, colindex):
    """Test hook (host only): (row blocks in 16-bit delta form, row blocks) of a CSR pattern; raises if the form does not decode
    back to colindex."""
    rp = np.ascontiguousarray(rowptr, dtype=np.int32)
    ci = np.ascontiguousarray(colindex, dtype=np.int32)
    a, b = C.c_long(), C.c_long()
    _check(lib.sparsh_debug_index16_roundtrip(len(rp) - 1, _ip(rp), _ip(ci), C.byref(a), C.byref(b)))
    return a.value, b.value


def device_count() -> int:
    return lib.sparsh_device_count()


def host_cpus() -> int:
    """CPUs this process may use (affinity mask capped by the cgroup quota)."""
    return lib.sparsh_host_cpus()


class sp_matrix_mg:
    """Host CSR container named after the reference's class (include/AMG_cpu_matrix.hpp:12-51).

    Holds rowptr/colindex/val as contiguous int32/float64 numpy arrays (aliased by the native
    handle, never copied) and owns the native solver handle.
    """

    def __init__(self, rowptr, colindex, val, ncol=None):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        self.colindex = np.ascontiguousarray(colindex, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        self.nrow = len(self.rowptr) - 1
        self.ncol = self.nrow if ncol is None else int(ncol)
        self.nnz = int(self.rowptr[-1])
        if len(self.colindex) < self.nnz or len(self.val) < self.nnz:
            raise ValueError("colindex/val shorter than rowptr[-1]")
        h = C.c_void_p()
        _check(lib.sparsh_create_csr(self.nrow, self.ncol, _ip(self.rowptr), _ip(self.colindex), _dp(self.val), C.byref(h)))
        self._h = h
        self.params = None

    # -- lifecycle -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib.sparsh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- per-handle kernel / layout choices (A/B measurements; results are bitwise identical) -----
    def set_const_slots(self, enable=True):
        """Layout option read by setup(): fold constant diagonals of a slice into one scalar."""
        _check(lib.sparsh_set_const_slots(self._h, int(bool(enable))))
        return self

    def set_setup_broadcast(self, enable=True):
        """Multi-GPU: rank 0 builds the hierarchy and broadcasts it (default) / every rank builds its own; before setup."""
        _check(lib.sparsh_set_setup_broadcast(self._h, int(bool(enable))))
        return self

    def hierarchy_roundtrip(self, truncate_to=-1):
        """Test hook: hierarchy -> byte image -> hierarchy, compared array by array; returns the image size."""
        r = lib.sparsh_debug_hierarchy_roundtrip(self._h, int(truncate_to))
        if r < 0:
            _check(int(r))
        return int(r)

    def setup_share_info(self):
        """(this rank ran the host setup itself, bytes of the broadcast hierarchy image)."""
        a, b = C.c_int(), C.c_long()
        _check(lib.sparsh_setup_share_info(self._h, C.byref(a), C.byref(b)))
        return bool(a.value), b.value

    def set_placement_search(self, enable=True):
        """Setup-time choice of the buffers that hold the finest level's sweep vectors (default on); before setup."""
        _check(lib.sparsh_set_placement_search(self._h, int(bool(enable))))
        return self

    def placement_info(self):
        """Sweep time (us) of the chosen / worst / initial buffer triple and the number of triples timed at setup."""
        a, b, c, k, t = C.c_double(), C.c_double(), C.c_double(), C.c_int(), C.c_double()
        _check(lib.sparsh_placement_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(k), C.byref(t)))
        return {"chosen_us": round(a.value, 2), "worst_us": round(b.value, 2), "initial_us": round(c.value, 2), "triples": k.value,
                "seconds": round(t.value, 3)}

    def set_fused_zero_sweep(self, enable=2):
        """PCG: cg_update also writes the zero-guess sweep of the V-cycle: 0 off, 1 fused, 2 fused + non-temporal streams (default)."""
        _check(lib.sparsh_set_fused_zero_sweep(self._h, int(enable)))
        return self

    def set_alternate_sweeps(self, mode=1):
        """Alternate the walking direction of consecutive sweeps of a smoothing leg: 0 never, 1 large streaming levels (default), 2 always."""
        _check(lib.sparsh_set_alternate_sweeps(self._h, int(mode)))
        return self

    def set_double_sweep(self, mode=1):
        """Two Jacobi sweeps per launch on box-grid levels: 0 never, 1 where the setup times it faster (default), 2 wherever a plan exists.
        Read by setup; afterwards it can be switched between 0 and the setup's value."""
        _check(lib.sparsh_set_double_sweep(self._h, int(mode)))
        return self

    def level_double_sweep(self, level):
        on = C.c_int(0)
        dims, plan = (C.c_int * 3)(), (C.c_int * 3)()
        t1, t2 = C.c_double(0.0), C.c_double(0.0)
        _check(lib.sparsh_level_double_sweep(self._h, int(level), C.byref(on), dims, plan, C.byref(t1), C.byref(t2)))
        return {"on": bool(on.value), "grid": list(dims), "points_per_thread": plan[0], "lines_per_tile": plan[1], "planes_per_chunk": plan[2],
                "two_single_sweeps_us": round(t1.value, 2), "double_sweep_us": round(t2.value, 2)}

    def set_deferred_x(self, enable=True):
        """PCG: x += alpha p rides in the direction update at the end of the iteration (p read once for both)."""
        _check(lib.sparsh_set_deferred_x(self._h, 1 if enable else 0))
        return self

    def set_zero_start(self, enable=True):
        """Double-sweep levels: a leg that starts from a zero guess runs sweeps 1 - 3 as one launch that reads only the right-hand side."""
        _check(lib.sparsh_set_zero_start(self._h, 1 if enable else 0))
        return self

    def set_marching_ops(self, mode=1):
        """SpMV + dot, last post-sweep (+ dot / + prolongation), residual + pair restriction of box-grid levels through the plane-marching
        kernel: 0 never, 1 where the setup times it faster (default), 2 wherever a plan exists.  Read by setup."""
        _check(lib.sparsh_set_marching_ops(self._h, int(mode)))
        return self

    def level_marching_ops(self, level):
        on = C.c_int(0)
        plan = (C.c_int * 3)()
        t1, t2 = C.c_double(0.0), C.c_double(0.0)
        _check(lib.sparsh_level_marching_ops(self._h, int(level), C.byref(on), plan, C.byref(t1), C.byref(t2)))
        return {"on": bool(on.value), "points_per_thread": plan[0], "lines_per_tile": plan[1], "planes_per_chunk": plan[2],
                "table_kernel_us": round(t1.value, 2), "marching_kernel_us": round(t2.value, 2)}

    def set_constant_diagonal(self, enable=True):
        """Levels with one constant diagonal: the zero-guess sweeps take it as an argument instead of streaming diag[]."""
        _check(lib.sparsh_set_constant_diagonal(self._h, 1 if enable else 0))
        return self

    def level_constant_diagonal(self, level):
        """(qualifies under the current configuration, the level's first diagonal entry)."""
        f, v = C.c_int(0), C.c_double(0.0)
        _check(lib.sparsh_level_constant_diagonal(self._h, int(level), C.byref(f), C.byref(v)))
        return bool(f.value), v.value

    def set_fused_prolongation(self, enable=True):
        """The last post-sweep of a level adds its result to the finer level's iterate itself (aggregates of one or two rows)."""
        _check(lib.sparsh_set_fused_prolongation(self._h, 1 if enable else 0))
        return self

    def level_prolong_fused(self, level):
        """Whether `level`'s last post-sweep prolongates into level - 1 itself: 0 no, 1 row pairs (2J, 2J+1), 2 member records."""
        v = C.c_int(0)
        _check(lib.sparsh_level_prolong_fused(self._h, int(level), C.byref(v)))
        return v.value

    def op_jacobi_prolong(self, level, b, x, xf):
        """xf + P_{level-1} J(x) through the fused launch (levels where level_prolong_fused() is true)."""
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        xf = np.array(xf, dtype=np.float64)
        _check(lib.sparsh_op_jacobi_prolong(self._h, int(level), _dp(b), _dp(x), _dp(xf)))
        return xf

    def set_paired_restriction(self, enable=True):
        """Residual + restriction (+ the coarse zero-guess sweep) as one launch on levels whose aggregates are the row pairs (2J, 2J+1)."""
        _check(lib.sparsh_set_paired_restriction(self._h, 1 if enable else 0))
        return self

    def level_paired(self, level):
        """Whether `level` takes the fused residual + restriction launch: 0 no, 1 row pairs (2J, 2J+1), 2 / 3 box grid paired along y / z."""
        v = C.c_int(0)
        _check(lib.sparsh_level_paired(self._h, int(level), C.byref(v)))
        return v.value

    def set_index_compression(self, mode=1):
        """16-bit delta-coded column indices for the CSR-stream family (call before setup); see sparsh_set_index_compression."""
        _check(lib.sparsh_set_index_compression(self._h, int(mode)))
        return self

    def level_index16(self, level):
        """(row blocks of the level's operator in 16-bit index form, row blocks)."""
        a, b = C.c_long(), C.c_long()
        _check(lib.sparsh_level_index16(self._h, level, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_tile(self, enable=True):
        """LDS-tiled variant of the table kernel on whole-level launches of grid stencils (default on)."""
        _check(lib.sparsh_set_tile(self._h, int(bool(enable))))
        return self

    def level_tile_rows(self, level):
        r = C.c_int(0)
        _check(lib.sparsh_level_tile_rows(self._h, level, C.byref(r)))
        return r.value

    def set_kernel_config(self, kind=3, vec=3, nt=-1, remap=-1):
        """Select the SpMV-type kernel family of this handle; see sparsh_set_kernel_config."""
        _check(lib.sparsh_set_kernel_config(self._h, int(kind), int(vec), int(nt), int(remap)))
        return self

    def level_placement(self, level):
        nt, remap = C.c_int(0), C.c_int(0)
        _check(lib.sparsh_level_placement(self._h, level, C.byref(nt), C.byref(remap)))
        return bool(nt.value), remap.value

    # -- setup -----------------------------------------------------------------------------
    def setup(self, params: Params | None = None, host_only: bool = False):
        """AMG_solver_setup_jacobi + GPU_Allocations (host_only: hierarchy only, no GPU)."""
        self.params = params if params is not None else default_params()
        fn = lib.sparsh_setup_host if host_only else lib.sparsh_setup
        _check(fn(self._h, C.byref(self.params)))
        return self

    def set_stopping(self, tol, max_iter=0, check_every=0):
        _check(lib.sparsh_set_stopping(self._h, tol, max_iter, check_every))

    @property
    def nlevels(self):
        return lib.sparsh_num_levels(self._h)

    @property
    def setup_seconds(self):
        return lib.sparsh_setup_seconds(self._h)

    def level_info(self, level):
        a = [C.c_int() for _ in range(4)]
        _check(lib.sparsh_level_info(self._h, level, *[C.byref(v) for v in a]))
        return dict(nrow=a[0].value, nnz=a[1].value, p_ncol=a[2].value, p_nnz=a[3].value)

    def level_format(self, level):
        """(kind, stored entries) of the layout the SpMV-type kernels use on this level."""
        k, e = C.c_int(), C.c_long()
        _check(lib.sparsh_level_format(self._h, level, C.byref(k), C.byref(e)))
        return k.value, e.value

    def bench_comm(self, what, level=0, reps=50):
        """Average seconds of one communication step alone (collective); -1 when there is none."""
        sec = C.c_double()
        _check(lib.sparsh_bench_comm(self._h, {"halo": 0, "allreduce": 1, "allgather": 2}[what], level, reps, C.byref(sec)))
        return sec.value

    def level_kernel(self, level):
        """Name of the kernel the SpMV-type operations of this level launch under the current config."""
        return lib.sparsh_level_kernel(self._h, level).decode()

    def level_layout(self, level):
        """(slots, value blocks, descriptor bytes) of the sliced-diagonal layout of this level; constant
        slots own no value block."""
        sl, vb, mb = C.c_long(), C.c_long(), C.c_long()
        _check(lib.sparsh_level_layout(self._h, level, C.byref(sl), C.byref(vb), C.byref(mb)))
        return sl.value, vb.value, mb.value

    def level_csr(self, level, which="A"):
        info = self.level_info(level)
        nrow = info["nrow"]
        nnz = info["nnz"] if which == "A" else info["p_nnz"]
        rp = np.zeros(nrow + 1, dtype=np.int32)
        ci = np.zeros(max(nnz, 1), dtype=np.int32)
        v = np.zeros(max(nnz, 1), dtype=np.float64)
        _check(lib.sparsh_level_csr(self._h, level, 0 if which == "A" else 1, _ip(rp), _ip(ci), _dp(v)))
        ncol = nrow if which == "A" else info["p_ncol"]
        return rp, ci[:nnz], v[:nnz], ncol

    def level_scipy(self, level, which="A"):
        import scipy.sparse as sp

        rp, ci, v, ncol = self.level_csr(level, which)
        return sp.csr_matrix((v, ci, rp), shape=(len(rp) - 1, ncol))

    def coarse_inverse(self):
        n = self.level_info(self.nlevels - 1)["nrow"]
        inv = np.zeros((n, n))
        _check(lib.sparsh_coarse_inverse(self._h, _dp(inv)))
        return inv

    def coarse_info(self):
        """Form of the coarsest-level direct solver (dense inverse, nested-dissection or block-tridiagonal factors)."""
        info = (C.c_int * 6)()
        nbytes = C.c_long(0)
        _check(lib.sparsh_coarse_info(self._h, info, C.byref(nbytes)))
        win = C.c_int(0)
        _check(lib.sparsh_coarse_window(self._h, C.byref(win)))
        nd = (C.c_int * 6)()
        _check(lib.sparsh_coarse_nd_info(self._h, nd))
        form = "dense" if info[1] else ("nested_dissection" if nd[0] else ("block_tridiagonal" if info[3] else "not factored yet"))
        return {"rows": info[0], "dense": bool(info[1]), "form": form, "block": info[2], "nblocks": info[3], "bandwidth": info[4],
                "extended": bool(info[5]), "bytes": nbytes.value, "window": win.value,
                "nd_nodes": nd[1], "nd_levels": nd[2], "nd_max_pivot": nd[3], "nd_launches_per_solve": nd[4], "nd_leaf": nd[5]}

    def set_coarse_form(self, form="nd", leaf=0, merge_rows=-1, top_merge_rows=-1):
        """Direct solver of a coarsest level above dense_limit rows: "nd" (nested-dissection multifrontal, default) or "bt"
        (block tridiagonal, round 2's); leaf / merge_rows / top_merge_rows tune the dissection (0 / -1 = defaults).  Call before setup."""
        _check(lib.sparsh_set_coarse_form(self._h, {"nd": 0, "bt": 1}[form], int(leaf), int(merge_rows)))
        if top_merge_rows >= 0:
            _check(lib.sparsh_set_coarse_top_merge(self._h, int(top_merge_rows)))
        return self

    def set_coarse_block(self, rows=0):
        """Block size of the block-tridiagonal coarse factorisation (0 = built-in rule); call before setup."""
        _check(lib.sparsh_set_coarse_block(self._h, int(rows)))
        return self

    def set_coarse_interface(self, enable=True):
        """Interface (window) form of the block-tridiagonal coarse solve, default on; call before setup."""
        _check(lib.sparsh_set_coarse_interface(self._h, int(enable)))
        return self

    def set_comm_tuning(self, enable=True):
        """Multi-rank setups: choose the partitioned levels and the smoothing schedule from the measured transport (default on when
        replicate_rows <= 0); call before setup."""
        _check(lib.sparsh_set_comm_tuning(self._h, int(bool(enable))))
        return self

    def comm_schedule(self):
        """Per-level decision table of the measured multi-rank schedule, or None when none was made."""
        out = []
        for l in range(self.nlevels):
            info = (C.c_int * 4)()
            cost = (C.c_double * 3)()
            rc = lib.sparsh_comm_schedule(self._h, l, info, cost)
            if rc != 0:
                return None
            out.append({"level": l, "rows": info[0], "boundary_rows": info[1], "partitioned": bool(info[2]), "deep_halo": bool(info[3]),
                        "model_us_deep_halo": round(cost[0], 2), "model_us_exchange_per_sweep": round(cost[1], 2), "model_us_replicated": round(cost[2], 2)})
        return out

    def plan_comm_schedule(self, nranks, exchange_us, exchange_us_per_MB, allreduce_us, allgather_us, allgather_us_per_MB, sweep_floor_us, sweep_us_per_MB):
        """Host-only what-if: the schedule the tuner would choose for `nranks` ranks from these measurements (no device, no transport)."""
        m = (C.c_double * 7)(exchange_us, exchange_us_per_MB, allreduce_us, allgather_us, allgather_us_per_MB, sweep_floor_us, sweep_us_per_MB)
        _check(lib.sparsh_plan_comm_schedule(self._h, int(nranks), m))
        return self.comm_schedule()

    def comm_measured(self):
        m = (C.c_double * 7)()
        if lib.sparsh_comm_measured(self._h, m) != 0:
            return None
        keys = ("exchange_us", "exchange_us_per_MB", "allreduce_us", "allgather_us", "allgather_us_per_MB", "sweep_floor_us", "sweep_us_per_MB")
        return {k: round(m[i], 3) for i, k in enumerate(keys)}

    def set_deep_halo(self, enable=True):
        """Deep-halo smoothing on partitioned levels (default on; call before setup)."""
        _check(lib.sparsh_set_deep_halo(self._h, int(bool(enable))))
        return self

    def deep_info(self, level):
        info = (C.c_int * 4)()
        _check(lib.sparsh_deep_info(self._h, level, info))
        d = {"K": info[0], "rows": info[1], "cols": info[2], "npad": info[3]}
        d["layer_end"] = []
        for q in range(d["K"] + 1 if d["K"] else 0):
            e = C.c_int()
            _check(lib.sparsh_deep_layer_end(self._h, level, q, C.byref(e)))
            d["layer_end"].append(e.value)
        return d

    def deep_prefix_spmv(self, level, rows, x_ext):
        x_ext = np.ascontiguousarray(x_ext, dtype=np.float64)
        y = np.zeros(rows)
        _check(lib.sparsh_deep_prefix_spmv(self._h, level, rows, _dp(x_ext), _dp(y)))
        return y

    def exchanges_issued(self):
        return int(lib.sparsh_exchanges_issued(self._h))

    # -- multi-GPU -------------------------------------------------------------------------
    def comm_init_rccl(self, unique_id: bytes, rank: int, nranks: int):
        """Install the RCCL transport (call before setup)."""
        assert len(unique_id) == 128
        _check(lib.sparsh_comm_init_rccl(self._h, unique_id, rank, nranks))

    def comm_init_group(self, group, rank: int):
        """Install the in-process test transport (call before setup)."""
        _check(lib.sparsh_comm_init_group(self._h, group, rank))

    def set_overlap(self, enable=True):
        """Multi-GPU: overlap halo exchange with the interior slices (same results)."""
        _check(lib.sparsh_set_overlap(self._h, 1 if enable else 0))

    def local_range(self, level=0):
        lo, hi, rep = C.c_int(), C.c_int(), C.c_int()
        _check(lib.sparsh_local_range(self._h, level, C.byref(lo), C.byref(hi), C.byref(rep)))
        return lo.value, hi.value, bool(rep.value)

    def dist_local_op(self, level, which, rank, nranks):
        """Host-only planning query: (scipy local matrix, plan dict) of operator A/P/R."""
        import scipy.sparse as sp

        w = {"A": 0, "P": 1, "R": 2}[which]
        sz = (C.c_int * 8)()
        _check(lib.sparsh_dist_local_op(self._h, level, w, rank, nranks, sz))
        nrow, nnz, nloc, nhalo, nss, nrs, nsend, row0 = list(sz)
        rp = np.zeros(nrow + 1, dtype=np.int32)
        ci = np.zeros(max(nnz, 1), dtype=np.int32)
        v = np.zeros(max(nnz, 1))
        hg = np.zeros(max(nhalo, 1), dtype=np.int32)
        si = np.zeros(max(nsend, 1), dtype=np.int32)
        ss = np.zeros(max(3 * nss, 1), dtype=np.int32)
        rs = np.zeros(max(3 * nrs, 1), dtype=np.int32)
        _check(lib.sparsh_dist_local_op_get(self._h, _ip(rp), _ip(ci), _dp(v), _ip(hg), _ip(si), _ip(ss), _ip(rs)))
        M = sp.csr_matrix((v[:nnz], ci[:nnz], rp), shape=(nrow, nloc + nhalo))
        plan = dict(nloc=nloc, nhalo=nhalo, row0=row0, halo_global=hg[:nhalo], send_idx=si[:nsend],
                    send=ss[: 3 * nss].reshape(-1, 3), recv=rs[: 3 * nrs].reshape(-1, 3))
        return M, plan

    def dist_deep_op(self, level, rank, nranks, K, depth):
        """Host-only planning query of the deep-halo layout: (scipy local matrix, plan dict)."""
        import scipy.sparse as sp

        sz = (C.c_int * 8)()
        _check(lib.sparsh_dist_deep_op(self._h, level, rank, nranks, K, depth, sz))
        nrow, nnz, nloc, npad, nall, nss, nrs, nsend = list(sz)
        rp = np.zeros(nrow + 1, dtype=np.int32)
        ci = np.zeros(max(nnz, 1), dtype=np.int32)
        v = np.zeros(max(nnz, 1))
        gof = np.zeros(max(nall, 1), dtype=np.int32)
        le = np.zeros(K + 1, dtype=np.int32)
        si = np.zeros(max(nsend, 1), dtype=np.int32)
        ss = np.zeros(max(3 * nss, 1), dtype=np.int32)
        rs = np.zeros(max(3 * nrs, 1), dtype=np.int32)
        nrecv_max = max(nall - npad, 1)
        rpos = np.zeros(nrecv_max, dtype=np.int32)
        _check(lib.sparsh_dist_deep_op_get(self._h, _ip(rp), _ip(ci), _dp(v), _ip(gof), _ip(le), _ip(si), _ip(ss), _ip(rpos), _ip(rs)))
        M = sp.csr_matrix((v[:nnz], ci[:nnz], rp), shape=(nrow, nall))
        recv = rs[: 3 * nrs].reshape(-1, 3)
        nrecv = int(recv[:, 2].sum()) if len(recv) else 0
        plan = dict(nloc=nloc, npad=npad, nall=nall, global_of=gof[:nall], layer_end=le, send_idx=si[:nsend],
                    send=ss[: 3 * nss].reshape(-1, 3), recv=recv, recv_pos=rpos[:nrecv])
        return M, plan

    # -- solvers (host vectors) ------------------------------------------------------------
    def vcycle(self, b, x, iterations=-1, hist_cap=8192):
        b = np.ascontiguousarray(b, dtype=np.float64)
        hist = np.zeros(hist_cap)
        n = C.c_int()
        rc = _check(lib.sparsh_vcycle(self._h, _dp(b), _dp(x), iterations, _dp(hist), hist_cap, C.byref(n)), allow=(SPARSH_ENOCONV,))
        return hist[: min(n.value, hist_cap)].copy(), rc

    def solve(self, method, b, x, hist_cap=8192, allow=()):
        """allow: extra return codes handed back instead of raised (SPARSH_ENOCONV always is)."""
        b = np.ascontiguousarray(b, dtype=np.float64)
        assert x.dtype == np.float64 and x.flags.c_contiguous
        hist = np.zeros(hist_cap)
        n = C.c_int()
        m = METHODS[method] if isinstance(method, str) else method
        rc = _check(lib.sparsh_solve(self._h, m, _dp(b), _dp(x), _dp(hist), hist_cap, C.byref(n)), allow=(SPARSH_ENOCONV,) + tuple(allow))
        return hist[: min(n.value, hist_cap)].copy(), rc

    # -- device-resident path (bench) ------------------------------------------------------
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        _check(lib.sparsh_dev_alloc(self._h, nbytes, C.byref(p)))
        return p

    def dev_free(self, p):
        _check(lib.sparsh_dev_free(self._h, p))

    def dev_fill(self, dptr, n, value=0.0):
        _check(lib.sparsh_dev_fill(self._h, dptr, n, value))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib.sparsh_h2d(self._h, dptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def d2h(self, arr, dptr):
        _check(lib.sparsh_d2h(self._h, arr.ctypes.data_as(C.c_void_p), dptr, arr.nbytes))

    def sync(self):
        _check(lib.sparsh_sync(self._h))

    def solve_dev(self, method, b_dev, x_dev, max_iters=0, hist_cap=8192):
        hist = np.zeros(hist_cap)
        n = C.c_int()
        sec = C.c_double()
        m = METHODS[method] if isinstance(method, str) else method
        rc = _check(
            lib.sparsh_solve_dev(self._h, m, b_dev, x_dev, max_iters, _dp(hist), hist_cap, C.byref(n), C.byref(sec)),
            allow=(SPARSH_ENOCONV,),
        )
        return hist[: min(n.value, hist_cap)].copy(), n.value, sec.value, rc

    def krylov_init_dev(self, method, b_dev, x_dev):
        m = METHODS[method] if isinstance(method, str) else method
        _check(lib.sparsh_krylov_init_dev(self._h, m, b_dev, x_dev))

    def krylov_step_dev(self, nsteps):
        done = C.c_int()
        res = C.c_double()
        _check(lib.sparsh_krylov_step_dev(self._h, nsteps, C.byref(done), C.byref(res)))
        return done.value, res.value

    def krylov_history(self, hist_cap=8192):
        hist = np.zeros(hist_cap)
        n = C.c_int()
        _check(lib.sparsh_krylov_history(self._h, _dp(hist), hist_cap, C.byref(n)))
        return hist[: min(n.value, hist_cap)].copy()

    def profile(self, enable=True):
        _check(lib.sparsh_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        out = np.zeros(4)
        _check(lib.sparsh_profile_read(self._h, _dp(out)))
        return dict(launches=int(out[0]), seconds=out[1], nrow=int(out[2]), nnz=int(out[3]))

    def bench_op(self, op, level=0, reps=20):
        ops = {"spmv": 0, "jacobi": 1, "residual": 2, "restrict": 3, "prolong": 4, "coarse": 5, "dot": 6, "axpby": 7, "copy_int": 8,
               "jacobi_pingpong": 9, "jacobi_pingpong_resident": 10, "jacobi_double": 11}
        sec = C.c_double()
        _check(lib.sparsh_bench_op(self._h, ops[op] if isinstance(op, str) else op, level, reps, C.byref(sec)))
        return sec.value

    # -- operators (host vectors; kernel parity tests) -------------------------------------
    def op_spmv(self, level, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.level_info(level)["nrow"])
        _check(lib.sparsh_op_spmv(self._h, level, _dp(x), _dp(y)))
        return y

    def op_jacobi(self, level, b, x, sweeps, x_is_zero=False):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.array(x, dtype=np.float64)
        _check(lib.sparsh_op_jacobi(self._h, level, _dp(b), _dp(x), sweeps, 1 if x_is_zero else 0))
        return x

    def op_residual(self, level, b, x):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        r = np.zeros_like(b)
        _check(lib.sparsh_op_residual(self._h, level, _dp(b), _dp(x), _dp(r)))
        return r

    def op_resnorm(self, level, b, x):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = C.c_double()
        _check(lib.sparsh_op_resnorm(self._h, level, _dp(b), _dp(x), C.byref(out)))
        return out.value

    def op_restrict(self, level, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        bc = np.zeros(self.level_info(level + 1)["nrow"])
        _check(lib.sparsh_op_restrict(self._h, level, _dp(r), _dp(bc)))
        return bc

    def op_residual_restrict(self, level, b, x):
        """(b_{l+1}, x_{l+1}) of the fused residual + restriction launch (levels where level_paired() is true)."""
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float64)
        nc = self.level_info(level + 1)["nrow"]
        bc, xc = np.zeros(nc), np.zeros(nc)
        _check(lib.sparsh_op_residual_restrict(self._h, level, _dp(b), _dp(x), _dp(bc), _dp(xc)))
        return bc, xc

    def op_prolong(self, level, xc, xf):
        xc = np.ascontiguousarray(xc, dtype=np.float64)
        xf = np.array(xf, dtype=np.float64)
        _check(lib.sparsh_op_prolong(self._h, level, _dp(xc), _dp(xf)))
        return xf

    def op_coarse(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b)
        _check(lib.sparsh_op_coarse(self._h, _dp(b), _dp(x)))
        return x

    def op_precond_f32(self, r):
        """z = V32(r): one application of the opt-in fp32 preconditioner (needs precond_fp32=1)."""
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        _check(lib.sparsh_op_precond_f32(self._h, _dp(r), _dp(z)))
        return z

    def op_dot(self, x, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        out = C.c_double()
        _check(lib.sparsh_op_dot(self._h, len(x), _dp(x), _dp(y), C.byref(out)))
        return out.value

    def op_nrm2(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = C.c_double()
        _check(lib.sparsh_op_nrm2(self._h, len(x), _dp(x), C.byref(out)))
        return out.value

    def op_axpby(self, a, x, b, y):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.array(y, dtype=np.float64)
        _check(lib.sparsh_op_axpby(self._h, len(x), a, _dp(x), b, _dp(y)))
        return y


# ---- entry points named as in the reference's include/AMG.hpp:40-85 --------------------------
# Each call redoes the setup, as the reference does (solver objects live inside one call).

def _entry(method):
    def run(A: sp_matrix_mg, b, x, params: Params | None = None):
        A.setup(params)
        hist, _ = A.solve(method, b, x)
        return hist

    return run


AMG_Solver_CPU_baseline = _entry("amg")
AMG_Solver_1 = AMG_Solver_CPU_baseline
AMG_Solver_CPU_GPU_CI = _entry("amg")
AMG_Solver_CPU_GPU_MI = _entry("amg")
Solver_CG_1 = Solver_CG_2 = _entry("cg")
Solver_PCG_1 = Solver_PCG_2 = Solver_PCG_3 = Solver_PCG_4 = _entry("pcg")
Solver_BiCG_1 = _entry("bicg")
Solver_PBiCG_1 = Solver_PBiCG_2 = Solver_PBiCG_3 = Solver_PBiCG_4 = _entry("pbicg")


def readcoo(matrixfile: str, rhsfile: str):
    """Native text format of the reference (src/AMG_file_read.cpp:39-72) -> (sp_matrix_mg, b)."""
    with open(matrixfile) as f:
        nrow, ncol, nnz = (int(t) for t in f.readline().split())
        data = np.loadtxt(f, dtype=np.float64, ndmin=2)
    if data.shape[0] != nnz:
        raise IOError(f"{matrixfile}: header says {nnz} entries, file holds {data.shape[0]}")
    rows = data[:, 0].astype(np.int64)
    if np.any(np.diff(rows) < 0):
        raise IOError("readcoo requires entries sorted by row (as the reference does)")
    rowptr = np.zeros(nrow + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=nrow), out=rowptr[1:])
    with open(rhsfile) as f:
        n = int(f.readline().split()[0])
        b = np.loadtxt(f, dtype=np.float64)
    assert n == nrow
    return sp_matrix_mg(rowptr, data[:, 1].astype(np.int32), data[:, 2].copy(), ncol=ncol), b[:nrow].copy()
