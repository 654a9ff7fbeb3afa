/* AddressSanitizer / UBSan driver for the oracle (test infrastructure). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "amg_oracle.h"

int main(void)
{
    const int n = 60, N = n * n;
    ocsr *A = oracle_csr_new(N, N, 5 * N);
    int q = 0;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            const int r = i + n * j;
            if (j > 0) { A->col[q] = r - n; A->val[q++] = -1; }
            if (i > 0) { A->col[q] = r - 1; A->val[q++] = -1; }
            A->col[q] = r; A->val[q++] = 4;
            if (i < n - 1) { A->col[q] = r + 1; A->val[q++] = -1; }
            if (j < n - 1) { A->col[q] = r + n; A->val[q++] = -1; }
            A->rowptr[r + 1] = q;
        }
    A->nnz = q;
    oracle_fill_diagonal(A);
    double *b = (double *)malloc(sizeof(double) * N), *x = (double *)calloc(N, sizeof(double));
    double hist[256];
    for (int i = 0; i < N; i++) b[i] = 1.0;
    for (int coarsening = 0; coarsening < 2; coarsening++) {
        oparams p;
        oracle_default_params(&p);
        p.limit_upper = 500;
        p.limit_lower = 250;
        p.coarsening = coarsening;
        const char *names[] = {"amg", "pcg", "pbicg", "cg", "bicg"};
        int (*fn[])(ocsr *, const double *, double *, const oparams *, double *, int) = {oracle_solver_amg, oracle_solver_pcg,
                                                                                          oracle_solver_pbicg, oracle_solver_cg,
                                                                                          oracle_solver_bicg};
        for (int m = 0; m < 5; m++) {
            for (int i = 0; i < N; i++) x[i] = 0.0;
            int it = fn[m](A, b, x, &p, hist, 256);
            double r = oracle_residual(A, b, x);
            if (!(r <= 1.0e-7)) {
                printf("oracle %s did not converge: %d its, r=%g\n", names[m], it, r);
                return 1;
            }
        }
    }
    free(b);
    free(x);
    oracle_csr_free(A);
    printf("ASAN_ORACLE_OK\n");
    return 0;
}
