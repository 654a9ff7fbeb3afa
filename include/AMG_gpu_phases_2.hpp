// AMG_gpu_phases_2.hpp -- AMG_GPU1_solver ("Hybrid AMG 2 / MI": whole hierarchy resident on the
// GPU) of the drop-in C++ API; names follow the reference's include/AMG_gpu_phases_2.hpp:11-42.
// On MI355X this IS the engine behind every entry point, so the class adds only the two calls
// the reference's GPU Krylov solvers use: helper (host b/x) and AMG_Solve (device b/x).
#ifndef AMG_GPU_PHASES_2_HPP_
#define AMG_GPU_PHASES_2_HPP_

#include "AMG_phases.hpp"

class AMG_GPU1_solver : public AMG_solver
{
  public:
    using AMG_solver::AMG_solver;

    // the hierarchy is already resident after AMG_solver_setup_jacobi; kept for source compatibility
    void GPU_Allocations();

    // b and x in host memory (src/AMG_gpu_phases_2.cu:242-263)
    void helper(double *b, double *x, int iterations);

    // b and x in device memory (hipMalloc'd, nrow doubles each) (src/AMG_gpu_phases_2.cu:96-240)
    void AMG_Solve(double *b, double *x, int iterations);
};

#endif
