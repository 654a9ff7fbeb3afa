// AMG_coarsening.hpp -- coarsening strategies of the drop-in C++ API (names of the reference's
// include/AMG_coarsening.hpp:6-27).  Host code.  HEM (the reference's default) and Beck are
// provided; the strategies the reference never enables (MIS -- non-deterministic --, compatible
// weighted matching, HEM-2) print a notice and return a null prolongator.
#ifndef AMG_COARSENING_HPP_
#define AMG_COARSENING_HPP_

#include "AMG_cpu_matrix.hpp"

namespace sequential
{
void HEM_Prolongator(sp_matrix_mg &A, sp_matrix_mg *&P, int l1);   // pairwise heavy-edge matching
void beck_prolongator(sp_matrix_mg &A, sp_matrix_mg *&P1);         // Beck's C/F interpolation
void mis_prolongator(sp_matrix_mg &A, sp_matrix_mg *&P1);
void C_W_prolongator(sp_matrix_mg &A, sp_matrix_mg *&P, int l1);
void HEM_Prolongator_2(sp_matrix_mg &A, sp_matrix_mg *&P);
}

#endif
