// AMG_smoothers.hpp -- smoother building blocks of the drop-in C++ API (names of the reference's
// include/AMG_smoothers.hpp:6-21).  Host vectors in/out; the sweeps run on the MI355X (the matrix is
// uploaded once and cached behind sp_matrix_mg::A1).  `iteration` keeps the reference's meaning:
// iteration+1 sweeps are performed (src/AMG_smoothers.cpp:59-60).
#ifndef AMG_SMOOTHERS_HPP_
#define AMG_SMOOTHERS_HPP_

#include "AMG_cpu_matrix.hpp"

namespace sequential
{
void jacobi_smoother(sp_matrix_mg &A, double *&b, double *&x, int iteration = 2);
void sor_smoother(sp_matrix_mg &A, double *&b, double *&x, int iteration = 2);  // not part of the MI355X build
}

namespace parallel
{
void jacobi_smoother(sp_matrix_mg &A, double *&b, double *&x, int iteration = 2);
void sor_smoother(sp_matrix_mg &A, double *&b, double *&x, int iteration = 2);  // not part of the MI355X build
}

#endif
