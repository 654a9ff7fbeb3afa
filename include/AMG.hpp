// AMG.hpp -- public header of the drop-in C++ API (MI355X build).
//
// Source-compatible with the reference's include/AMG.hpp:15-85: same macro names, same reader and
// solver signatures.  The macros are kept because user code may name them; the library itself
// reads the same defaults at run time (sparsh_params in include/sparsh_amg.h, env SPARSH_*).
//
// Every solver runs its solve phase on the MI355X (HIP kernels); there is no CPU solve path in
// this library.  Entry points that differed only in where the reference ran them (CPU / hybrid
// CI / hybrid MI) therefore share one device implementation with the CPU path's arithmetic
// (7 Jacobi sweeps, residual check before the first cycle).
#ifndef AMG_HPP_
#define AMG_HPP_

#include "AMG_matrix.hpp"
#include "AMG_cpu_matrix.hpp"

/* Parameters (values of the reference; informational for this build) */
#define th 2                         // OpenMP threads of the reference CPU solver
#define omega 0.66667                // relaxation factor of the Jacobi smoother
#define nsmooth 6                    // unused (as in the reference)
#define tol1 1e-8                    // absolute tolerance on ||b - Ax||_2
#define limit_upper 4000             // coarsen while the level has more rows than this
#define limit_lower 2000             // stop when a new level has fewer rows than this
#define level1 6                     // number of AMG levels
#define smooth_iter 6                // smoothing parameter (the CPU path performs smooth_iter+1 sweeps)

#define print_setup_phase_details 1
#define print_solve_phase_details 1

#define thgpu 1024                   // threads per block of the reference's CUDA kernels (unused here)

// Readers -------------------------------------------------------------------------------------
// matrix file: "nrow ncol nnz" then 0-based "row col val" triplets sorted by row; rhs file: count, values
void readcoo(char *matrixfile, char *rhsfile, sp_matrix_mg *&A, double *&b);
// banner line + '%' comments, sizes, 0-based triplets, then the rhs values in the same file
void read_coo_new_format(char *matrixfile, sp_matrix_mg *&A, double *&b);

// Additions to the reference's readers.  read_matrix_market: a real MatrixMarket
// coordinate reader -- 1-based indices, `%` comments, `general` / `symmetric` (mirrored) /
// `pattern` (values 1.0), entries in any order, duplicates summed -- returning a sorted CSR.
// write_csr_binary / read_csr_binary: a raw little-endian cache of a CSR matrix (header "SPARSHB1",
// nrow, ncol, nnz, then rowptr, colindex, val) so that 10 M-row inputs load in a fraction of a second.
// All three return false and leave A = nullptr on failure.
bool read_matrix_market(const char *file, sp_matrix_mg *&A);
bool write_csr_binary(const char *file, const sp_matrix_mg &A);
bool read_csr_binary(const char *file, sp_matrix_mg *&A);

// Solvers: x holds the initial guess on entry and the solution on return ----------------------
void AMG_Solver_CPU_baseline(sp_matrix_mg &A, double *&b, double *&x);  // AMG V(7,7) cycles until ||r|| <= tol1
void AMG_Solver_1(sp_matrix_mg &A, double *&b, double *&x);             // README name of the above
void AMG_Solver_2(sp_matrix_mg &A, double *&b, double *&x);             // SOR smoother: not in this build
void Solver_CG_1(sp_matrix_mg &A, double *&b, double *&x);
void Solver_CG_2(sp_matrix_mg &A, double *&b, double *&x);
void Solver_PCG_1(sp_matrix_mg &A, double *&b, double *&x);
void Solver_PCG_2(sp_matrix_mg &A, double *&b, double *&x);
void Solver_PCG_3(sp_matrix_mg &A, double *&b, double *&x);
void Solver_PCG_4(sp_matrix_mg &A, double *&b, double *&x);
void AMG_Solver_CPU_GPU_CI(sp_matrix_mg &A, double *&b, double *&x);
void AMG_Solver_CPU_GPU_MI(sp_matrix_mg &A, double *&b, double *&x);
void Solver_BiCG_1(sp_matrix_mg &A, double *&b, double *&x);
void Solver_PBiCG_1(sp_matrix_mg &A, double *&b, double *&x);
void Solver_PBiCG_2(sp_matrix_mg &A, double *&b, double *&x);
void Solver_PBiCG_3(sp_matrix_mg &A, double *&b, double *&x);
void Solver_PBiCG_4(sp_matrix_mg &A, double *&b, double *&x);
void coarsening_2(sp_matrix_mg &A, double *&b, double *&x);             // SOR test stub: not in this build

#endif
