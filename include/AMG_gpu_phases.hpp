// AMG_gpu_phases.hpp -- AMG_GPU_solver ("Hybrid AMG 1 / CI") of the drop-in C++ API; names follow
// the reference's include/AMG_gpu_phases.hpp:10-55.  The reference streams every level over PCIe
// each cycle so that small GPUs can hold it; with 288 GB of HBM the hierarchy is resident, so the
// class is the resident engine under its old name (the overlap helper methods have no meaning
// here and are not provided).
#ifndef AMG_GPU_PHASES_HPP_
#define AMG_GPU_PHASES_HPP_

#include "AMG_phases.hpp"

class AMG_GPU_solver : public AMG_solver
{
  public:
    using AMG_solver::AMG_solver;

    void GPU_Allocations();

    // rhs and solution in host memory (src/AMG_gpu_phases.cu:122-287)
    void AMG_GPU_solve(double *b, double *x, int iterations);

    // rhs and solution in device memory (src/AMG_gpu_phases.cu:290-459)
    void AMG_GPU_solve_1(double *b, double *x, int iterations);
};

#endif
