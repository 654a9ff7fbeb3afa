// sp_matrix_gpu + residual() used the way the reference's device CG does (src/AMG_main_solvers.cu:
// 283-300): upload one operator, smooth and measure residuals on device vectors.
#include "AMG.hpp"
#include "AMG_gpu_matrix.hpp"
#include "AMG_gpu_phase_utilities.hpp"

#include <cstdio>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    sp_matrix_mg *A = new sp_matrix_mg();
    double *b;
    readcoo(argv[1], argv[2], A, b);
    A->sp_matrix_fill();
    A->sp_matrix_fill_diagonal();
    const int n = A->nrow;
    hipStream_t st;
    if (hipStreamCreate(&st) != hipSuccess) return 3;
    sp_matrix_gpu *G = new sp_matrix_gpu(*A);
    G->matrix_transfer_gpu(*A, st);
    double *bd, *xd, *hd;
    hipMalloc(reinterpret_cast<void **>(&bd), sizeof(double) * n);
    hipMalloc(reinterpret_cast<void **>(&xd), sizeof(double) * n);
    hipMalloc(reinterpret_cast<void **>(&hd), sizeof(double) * n);
    std::vector<double> x((size_t)n);
    for (int i = 0; i < n; i++) x[(size_t)i] = 0.001 * (i % 17) - 0.003;
    hipMemcpy(bd, b, sizeof(double) * n, hipMemcpyHostToDevice);
    hipMemcpy(xd, x.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    const double r0 = residual(*G, bd, xd, hd, st);
    G->smooth_jacobi(bd, xd, hd, st, 5);  // odd count: exercises the copy back into x
    const double r5 = residual(*G, bd, xd, hd, st);
    G->smooth_jacobi(bd, xd, hd, st, 6);
    const double r11 = residual(*G, bd, xd, hd, st);
    hipMemcpy(x.data(), xd, sizeof(double) * n, hipMemcpyDeviceToHost);
    std::printf("GPUMAT %.17g %.17g %.17g %.17g %.17g\n", r0, r5, r11, x[0], x[(size_t)n / 2]);
    sp_matrix_gpu *H = nullptr;
    gpu_swap_pointers(G, H);
    if (G != nullptr || H == nullptr) return 4;
    double *xh = x.data();
    pin_AMG_MG_matrix(*A, b, xh, *A);
    unpin_AMG_MG_matrix(*A, b, xh, *A);
    delete H;
    hipFree(bd);
    hipFree(xd);
    hipFree(hd);
    hipStreamDestroy(st);
    return 0;
}
