// AMG_gpu_matrix.hpp -- sp_matrix_gpu of the drop-in C++ API: one CSR operator resident in HBM.
// Member and method names follow the reference's include/AMG_gpu_matrix.hpp:10-48; the CUDA /
// cuSPARSE / cuBLAS handle members are gone (the kernels are this library's own HIP kernels) and
// streams are hipStream_t.
#ifndef AMG_GPU_MATRIX_HPP_
#define AMG_GPU_MATRIX_HPP_

#include <hip/hip_runtime_api.h>

#include "AMG_cpu_matrix.hpp"

class sp_matrix_gpu
{
  public:
    int nrow;  // rows
    int ncol;  // columns
    int nnz;   // stored entries

    int *rowptr = nullptr;    // device
    int *colindex = nullptr;  // device
    double *val = nullptr;    // device
    double *diag = nullptr;   // device (square matrices only)

  public:
    // allocates device storage sized for A
    sp_matrix_gpu(sp_matrix_mg &A);

    // copies A's CSR arrays (and diagonal, if square) to the device on the given stream
    void matrix_transfer_gpu(sp_matrix_mg &A, hipStream_t streams);

    // `steps` weighted-Jacobi sweeps x <- x - omega (A x - b)/d on device vectors; hgpu is scratch
    // of nrow doubles (src/AMG_gpu_matrix.cu:106-127; each sweep is one fused kernel here)
    void smooth_jacobi(double *bgpu, double *xgpu, double *hgpu, hipStream_t streams, int steps);

    ~sp_matrix_gpu();

    void *impl_ = nullptr;  // row-block schedule, reduction workspace
};

#endif
