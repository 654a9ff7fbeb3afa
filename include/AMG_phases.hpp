// AMG_phases.hpp -- AMG_solver of the drop-in C++ API: hierarchy setup + V-cycle driver.
// Public names follow the reference's include/AMG_phases.hpp:8-53.  Setup runs on the host and
// uploads the hierarchy to the MI355X; AMG_solve_jacobi runs the V(nu,nu) cycles on the device
// (same arithmetic as src/AMG_phases.cpp:151-230: iterations = k > 0 fixed count, -1 until
// ||Ax-b|| <= tol1).  Av[i] / Pv[i] expose the host copies of the level operators for inspection;
// the per-level vectors live in HBM, so Xv/Bv/Rv stay null.
#ifndef AMG_PHASES_HPP_
#define AMG_PHASES_HPP_

#include "AMG_cpu_matrix.hpp"

class AMG_solver
{
  public:
    int l = 0;  // index of the coarsest level

    sp_matrix_mg **Av = nullptr;  // l+1 level operators (Av[0] aliases the caller's matrix)
    sp_matrix_mg **Pv = nullptr;  // l prolongators

    double **Xv = nullptr;  // kept for source compatibility; vectors are device-resident
    double **Bv = nullptr;
    double **Rv = nullptr;

    void *Directsolve = nullptr;  // coarse direct solve lives inside the engine (explicit inverse on the device)

  public:
    AMG_solver();

    void AMG_solver_setup_jacobi(sp_matrix_mg &A);
    void AMG_solver_setup_SOR(sp_matrix_mg &A);  // SOR path: not part of the MI355X build (prints a notice)

    void AMG_solve_jacobi(double *&b, double *&x, int iterations);
    void AMG_solve_SOR(double *&b, double *&x, int iterations);  // prints a notice, leaves x unchanged

    ~AMG_solver();

  protected:
    void *engine_ = nullptr;  // sparsh_handle
    void release();
};

#endif
