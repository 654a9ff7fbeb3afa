/*
 * oracle/amg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see amg_oracle.h).
 *
 * CPU restatement of the reference's CPU/OpenMP AMG solve + setup path.
 * Where the reference calls MKL (sparse mv / spmm / cblas / PARDISO) the
 * published semantics of those calls are restated in plain C:
 *   mkl_sparse_d_mv (non-transpose)  -> row-wise sum in stored column order
 *   mkl_sparse_d_mv (transpose)      -> sequential scatter order (row-major walk)
 *   mkl_sparse_spmm + order          -> Gustavson SpGEMM, structural entries kept, columns sorted
 *   cblas_daxpy/daxpby/ddot/dnrm2    -> element loops; sums with a fixed, thread-count
 *                                       independent chunked order (deterministic)
 *   PARDISO (mtype 11, LU)           -> RCM ordering + banded LU with partial pivoting
 * Compile with -ffp-contract=off -mno-fma so every a*b+c is two IEEE roundings,
 * as in the reference's scalar loops.
 */
#include "amg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ params */

void oracle_default_params(oparams *p)
{
    /* include/AMG.hpp:15-27 */
    p->threads = 2;
    p->omega = 0.66667;
    p->tol = 1e-8;
    p->limit_upper = 4000;
    p->limit_lower = 2000;
    p->max_levels = 6;
    p->smooth_iter = 6;
    p->coarsening = 0;
    p->max_iter = 100000;
}

void oracle_set_threads(int t)
{
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

static double now_sec(void)
{
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

/* --------------------------------------------------------------- containers */

/* sp_matrix::sp_matrix(r,c,n): zero-filled arrays (src/AMG_matrix.cpp:15-36) */
ocsr *oracle_csr_new(int nrow, int ncol, int nnz)
{
    ocsr *A = (ocsr *)calloc(1, sizeof(ocsr));
    A->nrow = nrow;
    A->ncol = ncol;
    A->nnz = nnz;
    A->rowptr = (int *)calloc((size_t)nrow + 1, sizeof(int));
    A->col = (int *)calloc((size_t)(nnz > 0 ? nnz : 1), sizeof(int));
    A->val = (double *)calloc((size_t)(nnz > 0 ? nnz : 1), sizeof(double));
    return A;
}

ocsr *oracle_csr_from(int nrow, int ncol, const int *rowptr, const int *col, const double *val)
{
    int nnz = rowptr[nrow];
    ocsr *A = oracle_csr_new(nrow, ncol, nnz);
    memcpy(A->rowptr, rowptr, sizeof(int) * ((size_t)nrow + 1));
    memcpy(A->col, col, sizeof(int) * (size_t)nnz);
    memcpy(A->val, val, sizeof(double) * (size_t)nnz);
    return A;
}

void oracle_csr_free(ocsr *A)
{
    if (!A) return;
    free(A->rowptr);
    free(A->col);
    free(A->val);
    free(A->diag);
    free(A->helper);
    free(A);
}

/* sp_matrix_mg::sp_matrix_fill_diagonal (src/AMG_cpu_matrix.cpp:35-51) */
void oracle_fill_diagonal(ocsr *A)
{
    free(A->diag);
    free(A->helper);
    A->diag = (double *)calloc((size_t)A->nrow, sizeof(double));
    A->helper = (double *)calloc((size_t)A->nrow, sizeof(double));
    for (int i = 0; i < A->nrow; i++) {
        for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) {
            if (A->col[j] == i) {
                A->diag[i] = A->val[j];
                break;
            }
        }
    }
}

/* mkl_sparse_order in sp_matrix_fill (src/AMG_cpu_matrix.cpp:17-31): sort columns within each row */
void oracle_sort_columns(ocsr *A)
{
    for (int i = 0; i < A->nrow; i++) {
        int s = A->rowptr[i], e = A->rowptr[i + 1];
        for (int j = s + 1; j < e; j++) { /* insertion sort: rows are short */
            int c = A->col[j];
            double v = A->val[j];
            int k = j - 1;
            while (k >= s && A->col[k] > c) {
                A->col[k + 1] = A->col[k];
                A->val[k + 1] = A->val[k];
                k--;
            }
            A->col[k + 1] = c;
            A->val[k + 1] = v;
        }
    }
}

/* readcoo (src/AMG_file_read.cpp:39-72): "nrow ncol nnz" then row-sorted 0-based triplets;
 * rhs file: count then one value per line. */
int oracle_readcoo(const char *matrixfile, const char *rhsfile, ocsr **Aout, double **bout)
{
    FILE *f = fopen(matrixfile, "r");
    if (!f) return -1;
    int nrow, ncol, nnz;
    if (fscanf(f, "%d %d %d", &nrow, &ncol, &nnz) != 3) {
        fclose(f);
        return -2;
    }
    ocsr *A = oracle_csr_new(nrow, ncol, nnz);
    double *b = (double *)calloc((size_t)nrow, sizeof(double));
    for (int i = 0; i < nnz; i++) {
        int k;
        if (fscanf(f, "%d %d %lf", &k, &A->col[i], &A->val[i]) != 3) {
            fclose(f);
            return -3;
        }
        A->rowptr[k + 1]++;
    }
    fclose(f);
    f = fopen(rhsfile, "r");
    if (!f) return -4;
    int k;
    if (fscanf(f, "%d", &k) != 1) {
        fclose(f);
        return -5;
    }
    for (int i = 0; i < nrow; i++) {
        if (fscanf(f, "%lf", &b[i]) != 1) {
            fclose(f);
            return -6;
        }
        A->rowptr[i + 1] += A->rowptr[i];
    }
    fclose(f);
    *Aout = A;
    *bout = b;
    return 0;
}

/* ---------------------------------------------------------------- BLAS-1 */

/* Deterministic reduction: fixed 4096-element chunks summed with 4 interleaved
 * partials, chunk results combined in index order.  Independent of thread count. */
#define OCHUNK 4096

static double chunk_dot(const double *x, const double *y, int n)
{
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = 0;
    for (; i + 3 < n; i += 4) {
        s0 += x[i] * y[i];
        s1 += x[i + 1] * y[i + 1];
        s2 += x[i + 2] * y[i + 2];
        s3 += x[i + 3] * y[i + 3];
    }
    for (; i < n; i++) s0 += x[i] * y[i];
    return (s0 + s1) + (s2 + s3);
}

/* cblas_ddot */
double oracle_dot(int n, const double *x, const double *y)
{
    int nchunk = (n + OCHUNK - 1) / OCHUNK;
    if (nchunk <= 1) return chunk_dot(x, y, n);
    double *part = (double *)malloc(sizeof(double) * (size_t)nchunk);
#pragma omp parallel for schedule(static)
    for (int c = 0; c < nchunk; c++) {
        int s = c * OCHUNK;
        int len = n - s < OCHUNK ? n - s : OCHUNK;
        part[c] = chunk_dot(x + s, y + s, len);
    }
    double sum = 0.0;
    for (int c = 0; c < nchunk; c++) sum += part[c];
    free(part);
    return sum;
}

/* cblas_dnrm2 */
double oracle_nrm2(int n, const double *x) { return sqrt(oracle_dot(n, x, x)); }

/* ------------------------------------------------------------- operators */

/* mkl_sparse_d_mv(NON_TRANSPOSE, 1.0, A, x, 0.0, y): y_i = sum_j a_ij x_j in stored order */
void oracle_spmv(const ocsr *A, const double *x, double *y)
{
    const int *rp = A->rowptr, *ci = A->col;
    const double *v = A->val;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < A->nrow; i++) {
        double sum = 0.0;
        for (int j = rp[i]; j < rp[i + 1]; j++) sum += v[j] * x[ci[j]];
        y[i] = sum;
    }
}

double oracle_time_spmv(const ocsr *A, const double *x, double *y, int reps, int threads)
{
    oracle_set_threads(threads);
    oracle_spmv(A, x, y);
    const double t0 = now_sec();
    for (int r = 0; r < reps; r++) oracle_spmv(A, x, y);
    return (now_sec() - t0) / (reps > 0 ? reps : 1);
}

double oracle_stream_triad(long n, int reps, int threads)
{
    oracle_set_threads(threads);
    double *a = (double *)malloc(sizeof(double) * (size_t)n);
    double *b = (double *)malloc(sizeof(double) * (size_t)n);
    double *c = (double *)malloc(sizeof(double) * (size_t)n);
    if (!a || !b || !c) {
        free(a);
        free(b);
        free(c);
        return 0.0;
    }
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; i++) {
        a[i] = 0.0;
        b[i] = 1.0;
        c[i] = 2.0;
    }
    double best = 0.0;
    for (int r = 0; r < reps; r++) {
        const double s = 3.0 + r;
        const double t0 = now_sec();
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; i++) a[i] = b[i] + s * c[i];
        const double dt = now_sec() - t0;
        const double gbs = 24.0 * (double)n / dt / 1e9;
        if (gbs > best) best = gbs;
    }
    volatile double sink = a[n / 2];
    (void)sink;
    free(a);
    free(b);
    free(c);
    return best;
}

/* mkl_sparse_d_mv(TRANSPOSE, 1.0, A, x, 0.0, y): y = A^T x, sequential scatter */
void oracle_spmv_t(const ocsr *A, const double *x, double *y)
{
    for (int j = 0; j < A->ncol; j++) y[j] = 0.0;
    for (int i = 0; i < A->nrow; i++)
        for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) y[A->col[j]] += A->val[j] * x[i];
}

/* parallel::jacobi_smoother (src/AMG_smoothers.cpp:53-76):
 * (iteration+1) sweeps of { h = A x ; h = b - h ; x += omega*h/d } */
void oracle_jacobi(ocsr *A, const double *b, double *x, int iteration, double omega)
{
    int n = A->nrow;
    double *h = A->helper;
    const double *d = A->diag;
    int count = 0;
    while (count++ <= iteration) {
        oracle_spmv(A, x, h);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) h[i] = 1.0 * b[i] + (-1.0) * h[i]; /* cblas_daxpby(n,1,b,-1,h) */
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) x[i] += omega * h[i] / d[i];
    }
}

/* parallel::residual (src/AMG_cycle_utilities.cpp:83-94): || A x - b ||_2 */
double oracle_residual(ocsr *A, const double *b, const double *x)
{
    int n = A->nrow;
    double *h = A->helper;
    oracle_spmv(A, x, h);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) h[i] += -1.0 * b[i]; /* cblas_daxpy(n,-1,b,h) */
    return oracle_nrm2(n, h);
}

/* parallel::store_residual (src/AMG_cycle_utilities.cpp:115-123): r = b - A x */
void oracle_store_residual(const ocsr *A, const double *b, const double *x, double *r)
{
    int n = A->nrow;
    oracle_spmv(A, x, r);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) r[i] = 1.0 * b[i] + (-1.0) * r[i];
}

/* parallel::transfer_residual (src/AMG_cycle_utilities.cpp:97-104): b_c = P^T r */
void oracle_transfer_residual(const ocsr *P, const double *r, double *b) { oracle_spmv_t(P, r, b); }

/* parallel::transfer_solution (src/AMG_cycle_utilities.cpp:107-112): x_f = P x_c + x_f */
void oracle_transfer_solution(const ocsr *P, const double *xc, double *xf)
{
    const int *rp = P->rowptr, *ci = P->col;
    const double *v = P->val;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P->nrow; i++) {
        double sum = 0.0;
        for (int j = rp[i]; j < rp[i + 1]; j++) sum += v[j] * xc[ci[j]];
        xf[i] = sum + xf[i];
    }
}

/* ------------------------------------------------------------ coarsening */

/* sequential::HEM_Prolongator (src/AMG_coarsening.cpp:14-97): pairwise heavy-edge
 * matching; forward row sweep on even levels, backward on odd; unmatched rows become
 * singletons numbered after all pairs.  P is nrow x ncoarse, one 1.0 per row. */
ocsr *oracle_hem_prolongator(const ocsr *A, int level)
{
    int n = A->nrow;
    ocsr *P = oracle_csr_new(n, 1, n);
    int newnum = 0;
    for (int i = 0; i < n; i++) {
        P->col[i] = -1;
        P->val[i] = 1.0;
    }
    int start = (level % 2 == 0) ? 0 : n - 1;
    int step = (level % 2 == 0) ? 1 : -1;
    for (int t = 0, i = start; t < n; t++, i += step) {
        P->rowptr[i] = i;
        if (P->col[i] == -1) {
            int id = -1;
            double max1 = 0.0;
            for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) {
                int c = A->col[j];
                if (P->col[c] == -1 && fabs(A->val[j]) > max1 && c != i) {
                    max1 = fabs(A->val[j]);
                    id = c;
                }
            }
            if (id != -1) {
                P->col[i] = newnum;
                P->col[id] = newnum;
                newnum++;
            }
        }
    }
    for (int i = 0; i < n; i++) {
        if (P->col[i] == -1) {
            P->col[i] = newnum;
            newnum++;
        }
    }
    P->rowptr[n] = n;
    P->ncol = newnum;
    return P;
}

/* sequential::beck_prolongator (src/AMG_coarsening.cpp:269-339): greedy C/F split in row
 * order; C rows inject; F rows average their C neighbours with weight 1/|c_f[i]|, where
 * |c_f[i]| counts how many C points listed i as a neighbour. */
ocsr *oracle_beck_prolongator(const ocsr *A)
{
    int n = A->nrow;
    int *c_f = (int *)calloc((size_t)n, sizeof(int));
    int c_count = 0;
    for (int i = 0; i < n; i++) {
        if (c_f[i] == 0) {
            for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) c_f[A->col[j]] -= 1;
            c_f[i] = c_count + 1;
            c_count++;
        }
    }
    int *rowp = (int *)calloc((size_t)n + 1, sizeof(int));
    for (int i = 0; i < n; i++) {
        if (c_f[i] > 0) {
            rowp[i + 1] = 1;
        } else if (c_f[i] < 0) {
            for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++)
                if (c_f[A->col[j]] > 0) rowp[i + 1]++;
        }
    }
    for (int i = 0; i < n; i++) rowp[i + 1] += rowp[i];
    ocsr *P = oracle_csr_new(n, c_count, rowp[n]);
    memcpy(P->rowptr, rowp, sizeof(int) * ((size_t)n + 1));
    for (int i = 0; i < n; i++) {
        int q = rowp[i];
        if (c_f[i] > 0) {
            P->col[q] = c_f[i] - 1;
            P->val[q] = 1.0;
        } else if (c_f[i] < 0) {
            double p1 = 1 / fabs((double)c_f[i]);
            for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) {
                int k = A->col[j];
                if (c_f[k] > 0) {
                    P->col[q] = c_f[k] - 1;
                    P->val[q] = p1;
                    q++;
                }
            }
        }
    }
    free(rowp);
    free(c_f);
    /* P1->sp_matrix_fill() -> mkl_sparse_order */
    oracle_sort_columns(P);
    return P;
}

/* explicit transpose; entries of each output row appear in increasing source-row order,
 * i.e. the order in which oracle_spmv_t scatters into that output element. */
ocsr *oracle_transpose(const ocsr *A)
{
    int nnz = A->rowptr[A->nrow];
    ocsr *T = oracle_csr_new(A->ncol, A->nrow, nnz);
    for (int j = 0; j < nnz; j++) T->rowptr[A->col[j] + 1]++;
    for (int i = 0; i < A->ncol; i++) T->rowptr[i + 1] += T->rowptr[i];
    int *next = (int *)malloc(sizeof(int) * ((size_t)A->ncol + 1));
    memcpy(next, T->rowptr, sizeof(int) * ((size_t)A->ncol + 1));
    for (int i = 0; i < A->nrow; i++) {
        for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) {
            int q = next[A->col[j]]++;
            T->col[q] = i;
            T->val[q] = A->val[j];
        }
    }
    free(next);
    return T;
}

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* mkl_sparse_spmm + mkl_sparse_order: C = A*B, Gustavson row-by-row; every structural
 * entry is kept (also exact numerical zeros); columns sorted. */
ocsr *oracle_spgemm(const ocsr *A, const ocsr *B)
{
    int n = A->nrow, m = B->ncol;
    int *rowp = (int *)calloc((size_t)n + 1, sizeof(int));
    /* symbolic */
#pragma omp parallel
    {
        int *mark = (int *)malloc(sizeof(int) * (size_t)m);
        for (int j = 0; j < m; j++) mark[j] = -1;
#pragma omp for schedule(static)
        for (int i = 0; i < n; i++) {
            int cnt = 0;
            for (int ja = A->rowptr[i]; ja < A->rowptr[i + 1]; ja++) {
                int k = A->col[ja];
                for (int jb = B->rowptr[k]; jb < B->rowptr[k + 1]; jb++) {
                    int c = B->col[jb];
                    if (mark[c] != i) {
                        mark[c] = i;
                        cnt++;
                    }
                }
            }
            rowp[i + 1] = cnt;
        }
        free(mark);
    }
    for (int i = 0; i < n; i++) rowp[i + 1] += rowp[i];
    ocsr *C = oracle_csr_new(n, m, rowp[n]);
    memcpy(C->rowptr, rowp, sizeof(int) * ((size_t)n + 1));
    free(rowp);
    /* numeric */
#pragma omp parallel
    {
        int *mark = (int *)malloc(sizeof(int) * (size_t)m);
        double *acc = (double *)malloc(sizeof(double) * (size_t)m);
        for (int j = 0; j < m; j++) mark[j] = -1;
#pragma omp for schedule(static)
        for (int i = 0; i < n; i++) {
            int s = C->rowptr[i], q = s;
            for (int ja = A->rowptr[i]; ja < A->rowptr[i + 1]; ja++) {
                int k = A->col[ja];
                double av = A->val[ja];
                for (int jb = B->rowptr[k]; jb < B->rowptr[k + 1]; jb++) {
                    int c = B->col[jb];
                    if (mark[c] != i) {
                        mark[c] = i;
                        C->col[q++] = c;
                        acc[c] = av * B->val[jb];
                    } else {
                        acc[c] += av * B->val[jb];
                    }
                }
            }
            qsort(C->col + s, (size_t)(q - s), sizeof(int), cmp_int);
            for (int j = s; j < q; j++) C->val[j] = acc[C->col[j]];
        }
        free(mark);
        free(acc);
    }
    return C;
}

/* parallel::coarsen_matrix (src/AMG_cycle_utilities.cpp:126-146): Ac = P^T (A P) */
ocsr *oracle_coarsen_matrix(const ocsr *A, const ocsr *P)
{
    ocsr *AP = oracle_spgemm(A, P);
    ocsr *R = oracle_transpose(P);
    ocsr *Ac = oracle_spgemm(R, AP);
    oracle_csr_free(AP);
    oracle_csr_free(R);
    oracle_fill_diagonal(Ac);
    return Ac;
}

/* ---------------------------------------------------------- coarse direct */

/* Direct_Solver_Pardiso (src/AMG_coarse_level_solver.cpp:9-76): mtype 11 (real
 * unsymmetric) sparse LU with a fill-reducing ordering, factor once (phase 12), solve per
 * V-cycle (phase 33).  Restated as: reverse Cuthill-McKee ordering of the symmetrised
 * pattern, then banded LU with partial pivoting (row interchanges inside the band, the
 * classic dgbtf2/dgbtrs scheme).  Any exact direct solver returns the same x up to
 * rounding; PARDISO's own pivot order is not reproducible from outside MKL. */

typedef struct oband {
    int n, kl, ku, W; /* row i holds columns i-kl .. i+ku+kl at ab[i*W + (j-i+kl)] */
    double *ab;
    int *piv;
    int *perm; /* perm[new] = old */
    double *work;
} oband;

static void rcm_order(const ocsr *A, int *perm)
{
    int n = A->nrow;
    /* symmetrised adjacency */
    ocsr *T = oracle_transpose(A);
    int *deg = (int *)calloc((size_t)n, sizeof(int));
    int *ap = (int *)calloc((size_t)n + 1, sizeof(int));
    for (int i = 0; i < n; i++) ap[i + 1] = ap[i] + (A->rowptr[i + 1] - A->rowptr[i]) + (T->rowptr[i + 1] - T->rowptr[i]);
    int *adj = (int *)malloc(sizeof(int) * (size_t)(ap[n] > 0 ? ap[n] : 1));
    for (int i = 0; i < n; i++) {
        int q = ap[i];
        for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) adj[q++] = A->col[j];
        for (int j = T->rowptr[i]; j < T->rowptr[i + 1]; j++) adj[q++] = T->col[j];
        qsort(adj + ap[i], (size_t)(q - ap[i]), sizeof(int), cmp_int);
        int u = ap[i];
        for (int j = ap[i]; j < q; j++)
            if (adj[j] != i && (u == ap[i] || adj[u - 1] != adj[j])) adj[u++] = adj[j];
        deg[i] = u - ap[i];
    }
    oracle_csr_free(T);
    int *visited = (int *)calloc((size_t)n, sizeof(int));
    int *order = (int *)malloc(sizeof(int) * (size_t)n);
    int *level = (int *)malloc(sizeof(int) * (size_t)n);
    int head = 0, tail = 0;
    for (int seed = 0; seed < n; seed++) {
        if (visited[seed]) continue;
        /* pseudo-peripheral start: repeat BFS from the last, lowest-degree node of the deepest level */
        int start = seed;
        for (int pass = 0; pass < 4; pass++) {
            int h = tail, t = tail;
            order[t++] = start;
            level[start] = 0;
            visited[start] = 2;
            while (h < t) {
                int v = order[h++];
                for (int j = ap[v]; j < ap[v] + deg[v]; j++) {
                    int w = adj[j];
                    if (visited[w] == 0) {
                        visited[w] = 2;
                        level[w] = level[v] + 1;
                        order[t++] = w;
                    }
                }
            }
            int last = order[t - 1], depth = level[last], best = last;
            for (int q = t - 1; q >= tail && level[order[q]] == depth; q--)
                if (deg[order[q]] < deg[best]) best = order[q];
            for (int q = tail; q < t; q++) visited[order[q]] = 0;
            if (best == start) break;
            start = best;
        }
        /* Cuthill-McKee BFS, neighbours by increasing degree */
        head = tail;
        order[tail++] = start;
        visited[start] = 1;
        while (head < tail) {
            int v = order[head++];
            int s = tail;
            for (int j = ap[v]; j < ap[v] + deg[v]; j++) {
                int w = adj[j];
                if (!visited[w]) {
                    visited[w] = 1;
                    order[tail++] = w;
                }
            }
            for (int a = s + 1; a < tail; a++) { /* insertion sort by degree */
                int w = order[a], k = a - 1;
                while (k >= s && deg[order[k]] > deg[w]) {
                    order[k + 1] = order[k];
                    k--;
                }
                order[k + 1] = w;
            }
        }
    }
    for (int i = 0; i < n; i++) perm[i] = order[n - 1 - i];
    free(deg);
    free(ap);
    free(adj);
    free(visited);
    free(order);
    free(level);
}

static oband *band_factor(const ocsr *A)
{
    int n = A->nrow;
    oband *F = (oband *)calloc(1, sizeof(oband));
    F->n = n;
    F->perm = (int *)malloc(sizeof(int) * (size_t)n);
    F->piv = (int *)malloc(sizeof(int) * (size_t)n);
    F->work = (double *)malloc(sizeof(double) * (size_t)n);
    rcm_order(A, F->perm);
    int *inv = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) inv[F->perm[i]] = i;
    int kl = 0, ku = 0;
    for (int i = 0; i < n; i++) {
        int ni = inv[i];
        for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) {
            int nj = inv[A->col[j]];
            if (ni - nj > kl) kl = ni - nj;
            if (nj - ni > ku) ku = nj - ni;
        }
    }
    F->kl = kl;
    F->ku = ku;
    int W = 2 * kl + ku + 1;
    F->W = W;
    F->ab = (double *)calloc((size_t)n * (size_t)W, sizeof(double));
    double *ab = F->ab;
    for (int i = 0; i < n; i++) {
        int ni = inv[i];
        for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) {
            int nj = inv[A->col[j]];
            ab[(size_t)ni * W + (nj - ni + kl)] += A->val[j];
        }
    }
    free(inv);
#define AB(i, j) ab[(size_t)(i) * W + ((j) - (i) + kl)]
    int uw = ku + kl; /* upper bandwidth of U after interchanges */
    for (int k = 0; k < n; k++) {
        int iend = k + kl < n - 1 ? k + kl : n - 1;
        int jend = k + uw < n - 1 ? k + uw : n - 1;
        int p = k;
        double amax = fabs(AB(k, k));
        for (int i = k + 1; i <= iend; i++) {
            double v = fabs(AB(i, k));
            if (v > amax) {
                amax = v;
                p = i;
            }
        }
        F->piv[k] = p;
        if (amax == 0.0) {
            fprintf(stderr, "oracle: singular coarse matrix at pivot %d\n", k);
            continue;
        }
        if (p != k) {
            for (int j = k; j <= jend; j++) {
                double t = AB(k, j);
                AB(k, j) = AB(p, j);
                AB(p, j) = t;
            }
        }
        double pv = AB(k, k);
        const double *rk = &AB(k, k);
        int len = jend - k;
#pragma omp parallel for schedule(static) if ((iend - k) * len > 20000)
        for (int i = k + 1; i <= iend; i++) {
            double *ri = &AB(i, k);
            double lik = ri[0] / pv;
            ri[0] = lik;
            for (int j = 1; j <= len; j++) ri[j] -= lik * rk[j];
        }
    }
    return F;
}

static void band_solve(const oband *F, const double *b, double *x)
{
    int n = F->n, kl = F->kl, W = F->W, uw = F->ku + F->kl;
    const double *ab = F->ab;
    double *y = F->work;
    for (int i = 0; i < n; i++) y[i] = b[F->perm[i]];
    for (int k = 0; k < n; k++) {
        int p = F->piv[k];
        if (p != k) {
            double t = y[k];
            y[k] = y[p];
            y[p] = t;
        }
        int iend = k + kl < n - 1 ? k + kl : n - 1;
        double yk = y[k];
        for (int i = k + 1; i <= iend; i++) y[i] -= AB(i, k) * yk;
    }
    for (int i = n - 1; i >= 0; i--) {
        int jend = i + uw < n - 1 ? i + uw : n - 1;
        double s = y[i];
        const double *ri = &AB(i, i);
        for (int j = 1; j <= jend - i; j++) s -= ri[j] * y[i + j];
        y[i] = s / ri[0];
    }
    for (int i = 0; i < n; i++) x[F->perm[i]] = y[i];
#undef AB
}

static void band_free(oband *F)
{
    if (!F) return;
    free(F->ab);
    free(F->piv);
    free(F->perm);
    free(F->work);
    free(F);
}

/* ------------------------------------------------------------- hierarchy */

struct oamg {
    int l; /* index of the coarsest level (AMG_solver::l, include/AMG_phases.hpp:13) */
    int cap;
    ocsr **A;
    ocsr **P;
    double **X, **B, **R;
    oband *direct; /* Directsolve */
    oparams prm;
};

/* AMG_solver::AMG_solver_setup_jacobi (src/AMG_phases.cpp:35-90) */
oamg *oracle_amg_setup(ocsr *A0, const oparams *prm)
{
    oamg *S = (oamg *)calloc(1, sizeof(oamg));
    S->prm = *prm;
    int cap = prm->max_levels;
    S->cap = cap;
    S->A = (ocsr **)calloc((size_t)cap, sizeof(ocsr *));
    S->P = (ocsr **)calloc((size_t)cap, sizeof(ocsr *));
    S->X = (double **)calloc((size_t)cap, sizeof(double *));
    S->B = (double **)calloc((size_t)cap, sizeof(double *));
    S->R = (double **)calloc((size_t)cap, sizeof(double *));
    oracle_set_threads(prm->threads);
    int l = 0;
    S->A[0] = A0;
    if (!A0->diag) oracle_fill_diagonal(A0);
    S->X[0] = (double *)calloc((size_t)A0->nrow, sizeof(double));
    S->B[0] = (double *)calloc((size_t)A0->nrow, sizeof(double));
    S->R[0] = (double *)calloc((size_t)A0->nrow, sizeof(double));
    while (S->A[l]->nrow > prm->limit_upper && l < cap - 1) {
        if (prm->coarsening == 1)
            S->P[l] = oracle_beck_prolongator(S->A[l]);
        else
            S->P[l] = oracle_hem_prolongator(S->A[l], l);
        S->A[l + 1] = oracle_coarsen_matrix(S->A[l], S->P[l]);
        l = l + 1;
        int n = S->A[l]->nrow;
        S->X[l] = (double *)calloc((size_t)n, sizeof(double));
        S->B[l] = (double *)calloc((size_t)n, sizeof(double));
        S->R[l] = (double *)calloc((size_t)n, sizeof(double));
        if (n < prm->limit_lower) break;
    }
    S->l = l;
    /* Directsolve = new Direct_Solver_Pardiso(*Av[l])  (src/AMG_phases.cpp:89) */
    S->direct = band_factor(S->A[l]);
    return S;
}

void oracle_amg_free(oamg *S)
{
    if (!S) return;
    for (int q = S->l; q >= 0; q--) {
        free(S->X[q]);
        free(S->B[q]);
        free(S->R[q]);
        if (q > 0) {
            oracle_csr_free(S->A[q]);
            oracle_csr_free(S->P[q - 1]);
        }
    }
    free(S->A);
    free(S->P);
    free(S->X);
    free(S->B);
    free(S->R);
    band_free(S->direct);
    free(S);
}

int oracle_amg_levels(const oamg *S) { return S->l + 1; }
const ocsr *oracle_amg_A(const oamg *S, int level) { return S->A[level]; }
const ocsr *oracle_amg_P(const oamg *S, int level) { return S->P[level]; }

void oracle_coarse_solve(const oamg *S, const double *b, double *x) { band_solve(S->direct, b, x); }

/* one V(nu,nu) cycle: body of the loops in AMG_solve_jacobi (src/AMG_phases.cpp:198-216) */
static void vcycle(oamg *S)
{
    int l = S->l;
    int it = S->prm.smooth_iter;
    double w = S->prm.omega;
    for (int l1 = 0; l1 < l; l1++) {
        oracle_jacobi(S->A[l1], S->B[l1], S->X[l1], it, w);
        oracle_store_residual(S->A[l1], S->B[l1], S->X[l1], S->R[l1]);
        oracle_transfer_residual(S->P[l1], S->R[l1], S->B[l1 + 1]);
        memset(S->X[l1 + 1], 0, sizeof(double) * (size_t)S->A[l1 + 1]->nrow);
    }
    oracle_coarse_solve(S, S->B[l], S->X[l]);
    for (int l1 = l; l1 > 0; l1--) {
        oracle_transfer_solution(S->P[l1 - 1], S->X[l1], S->X[l1 - 1]);
        oracle_jacobi(S->A[l1 - 1], S->B[l1 - 1], S->X[l1 - 1], it, w);
    }
}

/* ---- float restatement of the V-cycle (checker of the product's OPT-IN fp32 preconditioner; the
 * reference has no such mode: SURVEY section 8f-4 "mixed precision").  Same order of operations as
 * vcycle() above (src/AMG_phases.cpp:198-216, src/AMG_smoothers.cpp:62-72), every operand and every
 * intermediate rounded to float: values (float)a_ij, (float)omega, vectors float, zero initial guess on
 * every level (first sweep x = omega*b/d), row sums in stored order, restriction as the gather over
 * P^T in ascending fine-row order.  The coarsest solve is done in double and rounded (the product applies
 * a float inverse or its fp64 block factors: agreement to float accuracy times the coarse condition number). */
static void spmv_f32(const ocsr *A, const float *x, float *y)
{
    const int *rp = A->rowptr, *ci = A->col;
    const double *v = A->val;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < A->nrow; i++) {
        float sum = 0.f;
        for (int j = rp[i]; j < rp[i + 1]; j++) sum += (float)v[j] * x[ci[j]];
        y[i] = sum;
    }
}

/* z = V32(r): one V(nu,nu) cycle in float from a zero guess; sweeps = smooth_iter + 1 */
void oracle_vcycle_f32(oamg *S, const double *r, double *z)
{
    const int l = S->l;
    const int nu = S->prm.smooth_iter + 1;
    const float w = (float)S->prm.omega;
    oracle_set_threads(S->prm.threads);
    float **X = (float **)calloc((size_t)l + 1, sizeof(float *));
    float **B = (float **)calloc((size_t)l + 1, sizeof(float *));
    ocsr **Rt = (ocsr **)calloc((size_t)l + 1, sizeof(ocsr *));
    int nmax = 0;
    for (int k = 0; k <= l; k++) {
        const int n = S->A[k]->nrow;
        X[k] = (float *)calloc((size_t)n, sizeof(float));
        B[k] = (float *)calloc((size_t)n, sizeof(float));
        if (n > nmax) nmax = n;
        if (k < l) Rt[k] = oracle_transpose(S->P[k]);
    }
    float *h = (float *)calloc((size_t)nmax, sizeof(float));
    for (int i = 0; i < S->A[0]->nrow; i++) B[0][i] = (float)r[i];
    for (int k = 0; k < l; k++) {
        const ocsr *A = S->A[k];
        const int n = A->nrow;
        float *x = X[k], *b = B[k];
        for (int i = 0; i < n; i++) x[i] = w * b[i] / (float)A->diag[i]; /* first sweep from x = 0 */
        for (int sw = 1; sw < nu; sw++) {
            spmv_f32(A, x, h);
            for (int i = 0; i < n; i++) x[i] = x[i] + w * (b[i] - h[i]) / (float)A->diag[i];
        }
        spmv_f32(A, x, h);
        for (int i = 0; i < n; i++) h[i] = b[i] - h[i];
        spmv_f32(Rt[k], h, B[k + 1]); /* b_c = P^T r, gathered in ascending fine-row order */
    }
    {
        const int n = S->A[l]->nrow;
        double *bd = (double *)calloc((size_t)n, sizeof(double)), *xd = (double *)calloc((size_t)n, sizeof(double));
        for (int i = 0; i < n; i++) bd[i] = (double)B[l][i];
        oracle_coarse_solve(S, bd, xd);
        for (int i = 0; i < n; i++) X[l][i] = (float)xd[i];
        free(bd);
        free(xd);
    }
    for (int k = l; k > 0; k--) {
        const ocsr *A = S->A[k - 1], *P = S->P[k - 1];
        const int n = A->nrow;
        float *x = X[k - 1], *b = B[k - 1];
        for (int i = 0; i < n; i++) {
            float sum = 0.f;
            for (int j = P->rowptr[i]; j < P->rowptr[i + 1]; j++) sum += (float)P->val[j] * X[k][P->col[j]];
            x[i] = sum + x[i];
        }
        for (int sw = 0; sw < nu; sw++) {
            spmv_f32(A, x, h);
            for (int i = 0; i < n; i++) x[i] = x[i] + w * (b[i] - h[i]) / (float)A->diag[i];
        }
    }
    for (int i = 0; i < S->A[0]->nrow; i++) z[i] = (double)X[0][i];
    for (int k = 0; k <= l; k++) {
        free(X[k]);
        free(B[k]);
        if (Rt[k]) oracle_csr_free(Rt[k]);
    }
    free(X);
    free(B);
    free(Rt);
    free(h);
}

/* AMG_solver::AMG_solve_jacobi (src/AMG_phases.cpp:151-230).
 * iterations > 0: exactly that many V-cycles; iterations == -1: until ||Ax-b|| <= tol. */
int oracle_amg_solve(oamg *S, const double *b, double *x, int iterations, double *hist, int hist_cap)
{
    int n = S->A[0]->nrow;
    int cycles = 0;
    oracle_set_threads(S->prm.threads);
    memcpy(S->B[0], b, sizeof(double) * (size_t)n);
    memcpy(S->X[0], x, sizeof(double) * (size_t)n);
    double r1 = oracle_residual(S->A[0], S->B[0], S->X[0]);
    if (iterations > 0) {
        while (cycles < iterations) {
            vcycle(S);
            cycles++;
            r1 = oracle_residual(S->A[0], S->B[0], S->X[0]);
            if (hist && cycles - 1 < hist_cap) hist[cycles - 1] = r1;
        }
    }
    if (iterations == -1) {
        while (r1 > S->prm.tol && cycles < S->prm.max_iter) {
            vcycle(S);
            cycles++;
            r1 = oracle_residual(S->A[0], S->B[0], S->X[0]);
            if (hist && cycles - 1 < hist_cap) hist[cycles - 1] = r1;
        }
    }
    memcpy(x, S->X[0], sizeof(double) * (size_t)n);
    return cycles;
}

/* AMG_Solver_CPU_baseline (src/AMG_main_solvers.cpp:14-26) */
int oracle_solver_amg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap)
{
    oamg *S = oracle_amg_setup(A, prm);
    int c = oracle_amg_solve(S, b, x, -1, hist, hist_cap);
    oracle_amg_free(S);
    return c;
}

/* Solver_CG_1 (src/AMG_main_solvers.cpp:47-103).  The reference overwrites r with b
 * (assumes x0 = 0); here r0 = b - A x0, identical for x0 = 0. */
int oracle_solver_cg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap)
{
    int n = A->nrow;
    oracle_set_threads(prm->threads);
    if (!A->diag) oracle_fill_diagonal(A);
    double *Ap = (double *)calloc((size_t)n, sizeof(double));
    double *p = (double *)calloc((size_t)n, sizeof(double));
    double *r = A->helper;
    oracle_store_residual(A, b, x, r);
    memcpy(p, r, sizeof(double) * (size_t)n);
    double r1 = oracle_nrm2(n, r);
    int count = 0;
    while (count++ < n && r1 > 1e-8) {
        oracle_spmv(A, p, Ap);
        double alpha = oracle_dot(n, p, Ap);
        double s = oracle_dot(n, r, r);
        alpha = s / alpha;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) x[i] += alpha * p[i];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) r[i] += (-alpha) * Ap[i];
        double beta = oracle_dot(n, r, r) / s;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) p[i] = 1.0 * r[i] + beta * p[i];
        r1 = sqrt(s * beta);
        if (hist && count - 1 < hist_cap) hist[count - 1] = r1;
    }
    free(Ap);
    free(p);
    return count - 1;
}

/* loop of Solver_PCG_1 (src/AMG_main_solvers.cpp:107-167) on an existing hierarchy */
static int pcg_loop(oamg *S, const double *b, double *x, int max_it, double *hist, int hist_cap, double *seconds)
{
    ocsr *A = S->A[0];
    int n = A->nrow;
    double *Ap = (double *)calloc((size_t)n, sizeof(double));
    double *p = (double *)calloc((size_t)n, sizeof(double));
    double *z0 = (double *)calloc((size_t)n, sizeof(double)); /* zeroed: SURVEY Q2 */
    double *r0 = (double *)calloc((size_t)n, sizeof(double));
    double t0 = now_sec();
    oracle_store_residual(A, b, x, r0);
    double r1 = oracle_nrm2(n, r0);
    oracle_amg_solve(S, r0, z0, 1, NULL, 0);
    memcpy(p, z0, sizeof(double) * (size_t)n);
    int count = 0;
    while (count++ < n && r1 > S->prm.tol && count <= max_it) {
        oracle_spmv(A, p, Ap);
        double alpha = oracle_dot(n, p, Ap);
        double s = oracle_dot(n, r0, z0);
        alpha = s / alpha;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) x[i] += alpha * p[i];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) r0[i] += (-alpha) * Ap[i];
        memset(z0, 0, sizeof(double) * (size_t)n);
        oracle_amg_solve(S, r0, z0, 1, NULL, 0);
        double beta = oracle_dot(n, z0, r0) / s;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) p[i] = 1.0 * z0[i] + beta * p[i];
        r1 = oracle_nrm2(n, r0);
        if (hist && count - 1 < hist_cap) hist[count - 1] = r1;
    }
    if (seconds) *seconds = now_sec() - t0;
    free(Ap);
    free(p);
    free(z0);
    free(r0);
    return count - 1;
}

int oracle_solver_pcg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap)
{
    oamg *S = oracle_amg_setup(A, prm);
    int c = pcg_loop(S, b, x, prm->max_iter, hist, hist_cap, NULL);
    oracle_amg_free(S);
    return c;
}

int oracle_pcg_presetup(oamg *S, const double *b, double *x, int max_it, double *hist, int hist_cap, double *seconds)
{
    oracle_set_threads(S->prm.threads);
    return pcg_loop(S, b, x, max_it, hist, hist_cap, seconds);
}

/* Solver_BiCG_1 / Solver_PBiCG_1 (src/AMG_main_solvers.cpp:271-355, 358-458).
 * S == NULL: un-preconditioned (p1 = p, s1 = s). */
static int bicg_loop(ocsr *A, oamg *S, const double *b, double *x, const oparams *prm, double *hist, int hist_cap)
{
    int n = A->nrow;
    double *r0 = (double *)calloc((size_t)n, sizeof(double));
    double *r = (double *)calloc((size_t)n, sizeof(double));
    double *p = (double *)calloc((size_t)n, sizeof(double));
    double *Ap = (double *)calloc((size_t)n, sizeof(double));
    double *s = (double *)calloc((size_t)n, sizeof(double));
    double *As = (double *)calloc((size_t)n, sizeof(double));
    double *p1 = S ? (double *)calloc((size_t)n, sizeof(double)) : p;
    double *s1 = S ? (double *)calloc((size_t)n, sizeof(double)) : s;
    oracle_store_residual(A, b, x, r0);
    memcpy(r, r0, sizeof(double) * (size_t)n);
    memcpy(p, r0, sizeof(double) * (size_t)n);
    double res = oracle_nrm2(n, r0);
    int count = 0;
    while (res > prm->tol && count < prm->max_iter) {
        if (S) {
            memset(p1, 0, sizeof(double) * (size_t)n);
            oracle_amg_solve(S, p, p1, 1, NULL, 0);
        }
        double alpha1 = oracle_dot(n, r, r0);
        oracle_spmv(A, p1, Ap);
        double alpha = oracle_dot(n, Ap, r0);
        alpha = alpha1 / alpha;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) s[i] = r[i] - alpha * Ap[i];
        if (S) {
            memset(s1, 0, sizeof(double) * (size_t)n);
            oracle_amg_solve(S, s, s1, 1, NULL, 0);
        }
        oracle_spmv(A, s1, As);
        double omega1 = oracle_dot(n, As, s);
        omega1 /= oracle_dot(n, As, As);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) {
            x[i] = x[i] + alpha * p1[i] + omega1 * s1[i];
            r[i] = s[i] - omega1 * As[i];
        }
        double beta = oracle_dot(n, r, r0) / alpha1;
        beta = beta * (alpha / omega1);
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; i++) p[i] = r[i] + beta * (p[i] - omega1 * Ap[i]);
        res = oracle_nrm2(n, r);
        if (hist && count < hist_cap) hist[count] = res;
        count++;
    }
    free(r0);
    free(r);
    free(Ap);
    free(As);
    if (S) {
        free(p1);
        free(s1);
    }
    free(p);
    free(s);
    return count;
}

int oracle_solver_bicg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap)
{
    oracle_set_threads(prm->threads);
    if (!A->diag) oracle_fill_diagonal(A);
    return bicg_loop(A, NULL, b, x, prm, hist, hist_cap);
}

int oracle_solver_pbicg(ocsr *A, const double *b, double *x, const oparams *prm, double *hist, int hist_cap)
{
    oamg *S = oracle_amg_setup(A, prm);
    int c = bicg_loop(A, S, b, x, prm, hist, hist_cap);
    oracle_amg_free(S);
    return c;
}
