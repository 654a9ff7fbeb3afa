// AMG_coarse_level_solver.hpp -- coarsest-level direct solver of the drop-in C++ API.  The class
// keeps the reference's name (include/AMG_coarse_level_solver.hpp:7-28) although nothing of PARDISO
// is left: the constructor factors A on the host (reverse Cuthill-McKee + banded LU with partial
// pivoting) into an explicit inverse held in HBM, the solve is one device GEMV.
#ifndef AMG_COARSE_LEVEL_SOLVER_HPP_
#define AMG_COARSE_LEVEL_SOLVER_HPP_

#include "AMG_cpu_matrix.hpp"

class Direct_Solver_Pardiso
{
  public:
    int n = 0;      // order of the system
    int error = 0;  // 0 = factorisation succeeded

  public:
    Direct_Solver_Pardiso(sp_matrix_mg &A);                       // analyse + factor (exits on a singular matrix, as the reference does)
    void Direct_Solver_Pardiso_solve(double *&b, double *&x);     // host vectors
    ~Direct_Solver_Pardiso();

  private:
    void *impl_ = nullptr;
};

#endif
