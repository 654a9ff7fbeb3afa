// AMG_gpu_phase_utilities.hpp -- helpers of the reference's hybrid GPU path
// (include/AMG_gpu_phase_utilities.hpp:1-17) for the drop-in C++ API.
#ifndef AMG_GPU_PHASE_UTILITIES_HPP_
#define AMG_GPU_PHASE_UTILITIES_HPP_

#include "AMG_cpu_matrix.hpp"
#include "AMG_gpu_matrix.hpp"

// The reference page-locks nine host arrays per level so that it can stream levels over PCIe every
// cycle (src/AMG_gpu_phase_utilities.cu:11-126).  The hierarchy is resident in HBM here: no-ops.
void pin_AMG_MG_matrix(sp_matrix_mg &A, double *&b, double *&x, sp_matrix_mg &P);
void unpin_AMG_MG_matrix(sp_matrix_mg &A, double *&b, double *&x, sp_matrix_mg &P);

void gpu_swap_pointers(sp_matrix_gpu *&A1, sp_matrix_gpu *&A2);

// ||A x - b||_2 on device vectors, returned to the host; h is scratch of nrow doubles
double residual(sp_matrix_gpu &A, double *b, double *x, double *h, hipStream_t streams);

#endif
