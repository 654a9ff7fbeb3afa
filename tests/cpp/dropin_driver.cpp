// Drop-in check driver (own code, mirrors how the reference's main.cpp uses the API):
// readcoo -> sp_matrix_fill -> sp_matrix_fill_diagonal -> one solver entry point by name,
// then prints the true residual ||b - A x||_2 computed on the host and x[0].
#include "AMG.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    sp_matrix_mg *A = new sp_matrix_mg();
    double *b;
    readcoo(argv[1], argv[2], A, b);
    double *x = new double[A->nrow]();
    A->sp_matrix_fill();
    A->sp_matrix_fill_diagonal();
    const char *name = argv[3];
    if (!std::strcmp(name, "mi")) AMG_Solver_CPU_GPU_MI(*A, b, x);
    else if (!std::strcmp(name, "ci")) AMG_Solver_CPU_GPU_CI(*A, b, x);
    else if (!std::strcmp(name, "cpu")) AMG_Solver_CPU_baseline(*A, b, x);
    else if (!std::strcmp(name, "pcg1")) Solver_PCG_1(*A, b, x);
    else if (!std::strcmp(name, "pcg4")) Solver_PCG_4(*A, b, x);
    else if (!std::strcmp(name, "pbicg1")) Solver_PBiCG_1(*A, b, x);
    else if (!std::strcmp(name, "pbicg4")) Solver_PBiCG_4(*A, b, x);
    else if (!std::strcmp(name, "cg2")) Solver_CG_2(*A, b, x);
    else if (!std::strcmp(name, "bicg1")) Solver_BiCG_1(*A, b, x);
    else if (!std::strcmp(name, "sor")) AMG_Solver_2(*A, b, x);
    else return 3;
    double rr = 0.0;
    for (int i = 0; i < A->nrow; i++) {
        double s = 0.0;
        for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) s += A->val[j] * x[A->colindex[j]];
        rr += (b[i] - s) * (b[i] - s);
    }
    std::printf("RESULT %s residual %.6e x0 %.15e\n", name, std::sqrt(rr), x[0]);
    A->~sp_matrix_mg();
    delete[] x;
    delete[] b;
    return 0;
}
