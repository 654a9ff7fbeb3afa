/*
 * oracle/amg_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's CPU/OpenMP AMG path
 * (cmgcds/SParSH-AMG).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product
 * (sparsh_amg_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED (by this project's own rule): the reference holds no tests,
 * fixtures or golden vectors for this path, and the reference itself cannot be
 * built here (its CPU path needs mkl.h / PARDISO, which this image lacks; a
 * build would need stand-in headers, which is not allowed).  What this
 * restatement is held to are the residual histories the survey session
 * captured from the reference's own CPU sources (SURVEY.md Appendix A,
 * committed as tests/golden/appendix_a.json) -- but that session's build used
 * a declarations-only mkl.h, so those numbers do not count as a pin either
 * (see DESIGN.md section 2).
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it follows.
 */
#ifndef AMG_ORACLE_H_
#define AMG_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* CSR container; mirrors sp_matrix / sp_matrix_mg (include/AMG_matrix.hpp:6-32,
 * include/AMG_cpu_matrix.hpp:12-51) without the MKL handle. */
typedef struct ocsr {
    int nrow, ncol, nnz;
    int *rowptr;
    int *col;
    double *val;
    double *diag;   /* sp_matrix_fill_diagonal (src/AMG_cpu_matrix.cpp:35-51) */
    double *helper; /* n-vector scratch, as sp_matrix_mg::helper */
} ocsr;

/* Runtime form of the compile-time macros in include/AMG.hpp:15-27. */
typedef struct oparams {
    int threads;      /* th            = 2       */
    double omega;     /* omega         = 0.66667 */
    double tol;       /* tol1          = 1e-8    */
    int limit_upper;  /* limit_upper   = 4000    */
    int limit_lower;  /* limit_lower   = 2000    */
    int max_levels;   /* level1        = 6       */
    int smooth_iter;  /* smooth_iter   = 6  (CPU path does smooth_iter+1 sweeps) */
    int coarsening;   /* 0 = HEM (default, src/AMG_phases.cpp:60), 1 = Beck (:63) */
    int max_iter;     /* guard the reference lacks: cap on cycles / iterations  */
} oparams;

void oracle_default_params(oparams *p);

typedef struct oamg oamg;

/* ---- containers ---- */
ocsr *oracle_csr_new(int nrow, int ncol, int nnz);
ocsr *oracle_csr_from(int nrow, int ncol, const int *rowptr, const int *col, const double *val);
void oracle_csr_free(ocsr *A);
void oracle_fill_diagonal(ocsr *A);
void oracle_sort_columns(ocsr *A);
int oracle_readcoo(const char *matrixfile, const char *rhsfile, ocsr **A, double **b);

/* ---- operators ---- */
void oracle_set_threads(int t);
void oracle_spmv(const ocsr *A, const double *x, double *y);
void oracle_spmv_t(const ocsr *A, const double *x, double *y);
void oracle_jacobi(ocsr *A, const double *b, double *x, int iteration, double omega);
double oracle_residual(ocsr *A, const double *b, const double *x);
void oracle_store_residual(const ocsr *A, const double *b, const double *x, double *r);
void oracle_transfer_residual(const ocsr *P, const double *r, double *b);
void oracle_transfer_solution(const ocsr *P, const double *xc, double *xf);
double oracle_dot(int n, const double *x, const double *y);
double oracle_nrm2(int n, const double *x);

/* ---- setup ---- */
ocsr *oracle_hem_prolongator(const ocsr *A, int level);
ocsr *oracle_beck_prolongator(const ocsr *A);
ocsr *oracle_transpose(const ocsr *A);
ocsr *oracle_spgemm(const ocsr *A, const ocsr *B);
ocsr *oracle_coarsen_matrix(const ocsr *A, const ocsr *P);

oamg *oracle_amg_setup(ocsr *A, const oparams *prm);
void oracle_amg_free(oamg *S);
int oracle_amg_levels(const oamg *S);                 /* l + 1 */
const ocsr *oracle_amg_A(const oamg *S, int level);
const ocsr *oracle_amg_P(const oamg *S, int level);
void oracle_coarse_solve(const oamg *S, const double *b, double *x);

/* ---- solvers: return number of cycles/iterations; hist[k] = residual printed at step k ---- */
int oracle_amg_solve(oamg *S, const double *b, double *x, int iterations, double *hist, int hist_cap);
int oracle_solver_amg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap);
int oracle_solver_cg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap);
int oracle_solver_pcg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap);
int oracle_solver_bicg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap);
int oracle_solver_pbicg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap);

/* PCG on a pre-built hierarchy (used for the timed cpu_baseline: setup excluded,
 * at most max_it iterations). Returns iterations done; *seconds = solve-loop time. */
int oracle_pcg_presetup(oamg *S, const double *b, double *x, int max_it, double *hist, int hist_cap, double *seconds);

/* float restatement of one V(nu,nu) cycle from a zero guess, z = V32(r): checker of the product's opt-in
 * fp32 preconditioner mode (not a reference feature) */
void oracle_vcycle_f32(oamg *S, const double *r, double *z);

/* ---- cpu_baseline helpers of bench.py (timing only) ----
 * oracle_time_spmv: average seconds of one y = A x (the reference's mkl_sparse_d_mv call,
 * src/AMG_cycle_utilities.cpp:87) over `reps` repetitions after one warm-up, `threads` OpenMP threads.
 * oracle_stream_triad: best GB/s of a[i] = b[i] + s*c[i] over n doubles (24 n bytes per pass,
 * first-touch initialised by the same static schedule), `reps` passes: the host's STREAM-triad rate. */
double oracle_time_spmv(const ocsr *A, const double *x, double *y, int reps, int threads);
double oracle_stream_triad(long n, int reps, int threads);

#ifdef __cplusplus
}
#endif
#endif
