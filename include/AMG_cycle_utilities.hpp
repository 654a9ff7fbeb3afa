// AMG_cycle_utilities.hpp -- V-cycle building blocks of the drop-in C++ API (names of the reference's
// include/AMG_cycle_utilities.hpp:6-38).  Host vectors in/out, device execution (operators cached
// behind sp_matrix_mg::A1); coarsen_matrix runs on the host (setup stays on the host).
#ifndef AMG_CYCLE_UTILITIES_HPP_
#define AMG_CYCLE_UTILITIES_HPP_

#include "AMG_cpu_matrix.hpp"

namespace sequential
{
double residual(sp_matrix_mg &A, double *&b, double *&x);                          // ||A x - b||_2
void transfer_residual(sp_matrix_mg &P1, double *&r, double *&b);                  // b = P1^T r
void transfer_solution(sp_matrix_mg &P1, double *&x, double *&x1);                 // x1 = x1 + P1 x
void store_residual(sp_matrix_mg &A, double *&b, double *&x, double *&r);          // r = b - A x
void coarsen_matrix(sp_matrix_mg &A, sp_matrix_mg *&Ac, sp_matrix_mg &P1);         // Ac = P1^T A P1
}

namespace parallel
{
double residual(sp_matrix_mg &A, double *&b, double *&x);
void transfer_residual(sp_matrix_mg &P1, double *&r, double *&b);
void transfer_solution(sp_matrix_mg &P1, double *&x, double *&x1);
void store_residual(sp_matrix_mg &A, double *&b, double *&x, double *&r);
void coarsen_matrix(sp_matrix_mg &A, sp_matrix_mg *&Ac, sp_matrix_mg &P1);
void reorder_rhs(sp_matrix_mg &A, double *&b);                   // multicolour SOR only: not in this build
void reorder_prolongator(sp_matrix_mg &A, sp_matrix_mg *&P);     // multicolour SOR only: not in this build
}

#endif
