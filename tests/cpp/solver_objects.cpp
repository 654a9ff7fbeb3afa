// Uses the solver-object layer of the drop-in API the way the reference's own Krylov solvers do
// (src/AMG_main_solvers.cpp:129-147, src/AMG_main_solvers.cu:283-345): set the hierarchy up once,
// then call the V-cycle driver with host vectors (AMG_solver / helper) and with device vectors
// (AMG_GPU1_solver::AMG_Solve).
#include "AMG.hpp"
#include "AMG_gpu_phases.hpp"
#include "AMG_gpu_phases_2.hpp"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <vector>

static double nrm(const double *x, int n)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += x[i] * x[i];
    return std::sqrt(s);
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    sp_matrix_mg *A = new sp_matrix_mg();
    double *b;
    readcoo(argv[1], argv[2], A, b);
    A->sp_matrix_fill();
    A->sp_matrix_fill_diagonal();
    const int n = A->nrow;

    AMG_solver *S = new AMG_solver();
    S->AMG_solver_setup_jacobi(*A);
    std::printf("LEVELS %d", S->l + 1);
    for (int q = 0; q <= S->l; q++) std::printf(" %d:%d", S->Av[q]->nrow, S->Av[q]->rowptr[S->Av[q]->nrow]);
    std::printf(" P0 %dx%d\n", S->Pv[0]->nrow, S->Pv[0]->ncol);
    double *x = new double[n]();
    S->AMG_solve_jacobi(b, x, 3);  // exactly three V-cycles
    std::printf("HOST3 %.17g %.17g\n", x[0], nrm(x, n));
    S->~AMG_solver();

    AMG_GPU1_solver *G = new AMG_GPU1_solver();
    G->AMG_solver_setup_jacobi(*A);
    G->GPU_Allocations();
    std::vector<double> x2((size_t)n, 0.0);
    G->helper(b, x2.data(), 3);
    std::printf("HELPER3 %.17g %.17g\n", x2[0], nrm(x2.data(), n));
    double *bd = nullptr, *xd = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&bd), sizeof(double) * n) != hipSuccess) return 3;
    if (hipMalloc(reinterpret_cast<void **>(&xd), sizeof(double) * n) != hipSuccess) return 3;
    std::vector<double> x3((size_t)n, 0.0);
    hipMemcpy(bd, b, sizeof(double) * n, hipMemcpyHostToDevice);
    hipMemcpy(xd, x3.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    G->AMG_Solve(bd, xd, 3);
    hipMemcpy(x3.data(), xd, sizeof(double) * n, hipMemcpyDeviceToHost);
    std::printf("DEVICE3 %.17g %.17g\n", x3[0], nrm(x3.data(), n));
    hipFree(bd);
    hipFree(xd);
    G->~AMG_GPU1_solver();

    AMG_GPU_solver *C = new AMG_GPU_solver();
    C->AMG_solver_setup_jacobi(*A);
    C->GPU_Allocations();
    std::vector<double> x4((size_t)n, 0.0);
    C->AMG_GPU_solve(b, x4.data(), -1);  // until ||Ax-b|| <= tol1
    double rr = 0;
    for (int i = 0; i < n; i++) {
        double s = 0;
        for (int j = A->rowptr[i]; j < A->rowptr[i + 1]; j++) s += A->val[j] * x4[(size_t)A->colindex[j]];
        rr += (b[i] - s) * (b[i] - s);
    }
    std::printf("CI_SOLVE %.6e\n", std::sqrt(rr));
    C->~AMG_GPU_solver();

    A->~sp_matrix_mg();
    delete[] x;
    delete[] b;
    return 0;
}
