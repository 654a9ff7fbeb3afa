/*
 * sparsh_amg.h -- C ABI of libsparsh_amg.so: the MI355X-native AMG solve phase.
 *
 * The reference (cmgcds/SParSH-AMG) has no C ABI or plugin interface; its boundary is the
 * C++ source-level API of include/AMG.hpp (17 free functions on sp_matrix_mg, statically
 * linked).  That C++ surface is re-exported by this library through include/AMG.hpp,
 * include/AMG_matrix.hpp and include/AMG_cpu_matrix.hpp of THIS repo (same names, same
 * signatures), so the reference's main.cpp links against libsparsh_amg.so unchanged.
 *
 * This header is the plain-C mirror a non-C++ host (ctypes, cgo, JNI ...) binds: plain
 * pointers and sizes, int return codes, no globals.  Each entry cites the reference
 * interface it stands for (file:line relative to the reference tree).
 *
 * Conventions: 0-based CSR, int32 indices, fp64 values, caller-owned host buffers unless a
 * name ends in _dev (device pointers in HBM of the handle's GPU).  Return 0 on success,
 * negative SPARSH_E* otherwise; sparsh_last_error() gives the text.
 */
#ifndef SPARSH_AMG_H_
#define SPARSH_AMG_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SPARSH_OK 0
#define SPARSH_EINVAL -1    /* bad argument */
#define SPARSH_ENODEV -2    /* no HIP device / HIP runtime error (on the solve path: sticky until the next sparsh_setup) */
#define SPARSH_ESTATE -3    /* call order violated (e.g. solve before setup) */
#define SPARSH_ENUMERIC -4  /* singular coarse matrix, NaN residual */
#define SPARSH_ENOCONV -5   /* iteration cap reached before ||r|| <= tol (x still returned) */
#define SPARSH_ECOMM -6     /* transport (RCCL) failure on the solve path; sticky until the next sparsh_setup */

/* Solver selection: which reference entry point's arithmetic is followed. */
#define SPARSH_AMG 0    /* AMG_Solver_CPU_baseline / _CPU_GPU_MI / _CPU_GPU_CI  (src/AMG_main_solvers.cpp:14-26,240-268) */
#define SPARSH_CG 1     /* Solver_CG_1 / Solver_CG_2        (src/AMG_main_solvers.cpp:47-103, .cu:35-139)  */
#define SPARSH_PCG 2    /* Solver_PCG_1..4                  (src/AMG_main_solvers.cpp:107-167, .cu:269-413) */
#define SPARSH_BICG 3   /* Solver_BiCG_1                    (src/AMG_main_solvers.cpp:271-355) */
#define SPARSH_PBICG 4  /* Solver_PBiCG_1..4                (src/AMG_main_solvers.cpp:358-458) */

/* Runtime form of the compile-time macros of the reference's include/AMG.hpp:15-27.
 * sparsh_default_params() fills the reference values; the env variables in brackets
 * override them inside sparsh_default_params(). */
typedef struct sparsh_params {
    double omega;       /* omega = 0.66667 (not 2/3)                          [SPARSH_OMEGA]   */
    double tol;         /* tol1 = 1e-8, absolute ||b-Ax||_2                   [SPARSH_TOL]     */
    int sweeps;         /* Jacobi sweeps per smoothing step. CPU path of the reference does
                           smooth_iter+1 = 7 (src/AMG_smoothers.cpp:59-60); its GPU path 6. [SPARSH_NU] */
    int max_levels;     /* level1 = 6                                          [SPARSH_LEVELS]  */
    int limit_upper;    /* limit_upper = 4000                                                  */
    int limit_lower;    /* limit_lower = 2000                                                  */
    int coarsening;     /* 0 = HEM pairwise aggregation (default, src/AMG_phases.cpp:60),
                           1 = Beck (src/AMG_phases.cpp:63)                    [SPARSH_COARSENING=hem|beck] */
    int max_iter;       /* guard the reference lacks: cap on cycles/iterations [SPARSH_MAXIT]   */
    int coarse_limit;   /* largest coarsest level max_levels may leave for the device direct solver
                           (default 40000: up to ~1.3 M rows the hierarchy is exactly the reference's,
                           src/AMG_phases.cpp:51,77,89).  If max_levels would leave more, coarsening
                           continues by the same rule until <= extend_until rows (documented deviation
                           from the reference, which hands any size to PARDISO; 1<<30 disables it).
                                                                               [SPARSH_COARSE_LIMIT] */
    int host_threads;   /* OpenMP threads for the host setup (0 = all)        [SPARSH_THREADS] */
    int device;         /* HIP device ordinal (-1 = current / LOCAL_RANK)                     */
    int print_setup;    /* print_setup_phase_details = 1                       [SPARSH_PRINT]   */
    int print_solve;    /* print_solve_phase_details = 1                       [SPARSH_PRINT]   */
    int check_every;    /* Krylov/AMG loops read the residual norm back every k iterations
                           (1 = reference behaviour: every iteration)                          */
    int use_graph;      /* capture one V-cycle into a hipGraph and replay it   [SPARSH_GRAPH]   */
    int replicate_rows; /* multi-GPU: levels with at most this many rows are held and computed by every
                           rank (no halo exchange below that size).  0 (default): chosen at setup from the measured
                           transport, see sparsh_set_comm_tuning (1 500 000 when that is off) [SPARSH_REPLICATE_ROWS] */
    int precond_fp32;   /* 0 (default): everything fp64, bitwise parity with the reference's arithmetic.
                           1: SPARSH_PCG / SPARSH_PBICG run their V-cycle on a float copy of the hierarchy (float values
                           and vectors: the sliced-diagonal value blocks where a level has that layout, the
                           CSR values otherwise; one GPU) while
                           the CG recurrences, residuals and the stopping test stay fp64.  Not a parity
                           mode: a different (cheaper) preconditioner, same solution to tol.
                                                                               [SPARSH_PRECOND_FP32] */
    int dense_limit;    /* coarsest levels up to this many rows are solved with an explicit dense inverse
                           (one GEMV per V-cycle); larger ones with the block-tridiagonal factorisation of
                           the RCM-ordered operator, factored and applied on the device (csrc/coarse.cpp).
                           Default 8192.                                       [SPARSH_DENSE_LIMIT] */
    int extend_until;   /* where a hierarchy that had to be extended past max_levels (see coarse_limit) stops:
                           0 (default) = at the first level of at most coarse_limit rows, i.e. as soon as the
                           device direct solver can take over -- the closest to the reference's "6 levels, then
                           a direct solve" that fits; > 0 = at the first level of at most that many rows
                           (4000 = limit_upper: round 2's behaviour, 13 levels at 10 M rows).
                                                                               [SPARSH_EXTEND_UNTIL] */
    int coarse_factor_mb; /* the reference's own coarsest level (what max_levels leaves) is kept, whatever its row count, while the
                           ESTIMATED size of its nested-dissection factors stays below this many MB (one breadth-first search gives
                           the graph's effective dimension: ~ n log n bytes for 2D-like operators, ~ n^(4/3) for 3D-like ones).
                           Default 1024: a 9 M-row 2D problem keeps the reference's 6 levels (281 250-row direct solve, 0.5 GB),
                           136^3 does (78 608 rows, 0.4 GB), 216^3 does not (314 928 rows, 2.4 GB: extended).  0: coarse_limit
                           alone decides.                                      [SPARSH_COARSE_FACTOR_MB] */
} sparsh_params;

typedef struct sparsh_handle_s *sparsh_handle;

const char *sparsh_last_error(void);
int sparsh_version(void);
int sparsh_device_count(void);
int sparsh_host_cpus(void); /* CPUs this process may use (affinity mask and cgroup quota) */

void sparsh_default_params(sparsh_params *p);

/* sp_matrix_mg + sp_matrix_fill + sp_matrix_fill_diagonal (include/AMG_cpu_matrix.hpp:12-51,
 * src/AMG_cpu_matrix.cpp:17-51): register a host CSR.  The arrays are aliased, not copied
 * (as AMG_solver::Av[0] aliases the caller's matrix, src/AMG_phases.cpp:40) and must
 * outlive the handle.  Columns must be sorted within each row. */
int sparsh_create_csr(int nrow, int ncol, const int *rowptr, const int *colindex, const double *val, sparsh_handle *out);
void sparsh_destroy(sparsh_handle h);

/* AMG_solver::AMG_solver_setup_jacobi (src/AMG_phases.cpp:35-90) followed by what
 * AMG_GPU1_solver::GPU_Allocations does (src/AMG_gpu_phases_2.cu:13-94): build the hierarchy
 * on the host, factor the coarsest level, upload everything to HBM (resident). */
int sparsh_setup(sparsh_handle h, const sparsh_params *p);

/* Change the stopping rule of an already set-up handle (tol1 of include/AMG.hpp:18; iteration
 * cap; how often the residual norm is read back).  max_iter/check_every <= 0 keep their value. */
int sparsh_set_stopping(sparsh_handle h, double tol, int max_iter, int check_every);

/* Per-handle choice of the SpMV-type kernel family (A/B measurements; all families produce
 * bitwise identical results; nothing here is process-wide state).  kind: 0 workgroup CSR-stream, 1 wave CSR-stream, 2 sliced-ELL mirror,
 * 3 sliced-diagonal mirror (default); 2 and 3 fall back (3 -> 2 -> 0) where the operator does not
 * qualify for the mirror; vec (CSR-stream kernels): 0 one entry per load, 1 paired 16-B/8-B loads in the stream phase,
 * 2 col/val staged in LDS with the x gathers issued in row-lane order (csr_rowlane_kernel), 3 (default) = 2 for operators
 * that stream from HBM, 1 for cache-resident ones, 4 = 2 with 16-bit delta-coded column indices (csr_rowlane16_kernel) where the
 * operator carries them (sparsh_set_index_compression); nt: non-temporal loads for the matrix stream; remap: 0 none, 1 each XCD owns a
 * contiguous eighth of the row blocks, G > 1 groups of G row blocks dealt round-robin to the XCDs.
 * nt < 0 or remap < 0 selects the built-in per-operator policy (default). */
int sparsh_set_kernel_config(sparsh_handle h, int kind, int vec, int nt, int remap);

/* Which layout the SpMV-type kernels of a level use under the current config (3 sliced diagonals,
 * 2 sliced ELL, 1 wave CSR-stream, 0 workgroup CSR-stream) and how many entries it stores
 * (padding included). */
int sparsh_level_format(sparsh_handle h, int level, int *kind, long *stored_entries);

/* Sliced-diagonal layout of a level: number of (slice, diagonal) slots and how many of them own a
 * 64-value block.  A slot whose present entries all carry the same value ("constant slot":
 * constant-coefficient stencils and their aggregated coarse operators) keeps that value once in
 * its header and owns no block; the arithmetic is unchanged (same products, same order).  Slices
 * made of at most 8 constant slots are described by one fixed-stride 192-byte record (offsets,
 * lane masks, constants, count) the kernel fetches with a single batch of scalar loads; when
 * (nearly) all of them draw their (offset, constant) pairs from one set of at most 8, that set
 * travels as a kernel argument and a slice only needs its 8 lane masks (64 B).
 * meta_bytes = bytes of slice/slot descriptors one sweep reads.  All 0 when the level does not use
 * the layout.  sparsh_set_const_slots(h, 0) before sparsh_setup(h, ...) turns the folding off for that
 * handle (A/B measurements; default on). */
int sparsh_level_layout(sparsh_handle h, int level, long *slots, long *value_blocks, long *meta_bytes);
int sparsh_set_const_slots(sparsh_handle h, int enable);
/* Compressed column indices for the CSR-stream family: per row block of the workgroup kernel the indices
 * are also kept as 16-bit deltas (first entry of a row relative to the block's smallest first column, every further entry
 * relative to the previous column of its row), 10 instead of 12 bytes per stored entry; blocks a delta does not fit (a gap
 * of 65536 or more, an unsorted row, a single row longer than the LDS buffer) keep the 32-bit indices.  Same products, same
 * order of additions: results are bitwise those of the other families.  mode, read by sparsh_setup: 0 never build the form,
 * 1 (default) for operators the default policy streams from HBM through the CSR-stream kernel (> 240 MB, no sliced mirror),
 * 2 for every operator (A/B measurements with sparsh_set_kernel_config(h, 0, 4, ...)).
 * sparsh_level_index16: how many row blocks of a level's operator use the 16-bit form, out of how many. */
int sparsh_set_index_compression(sparsh_handle h, int mode);
/* Consecutive Jacobi sweeps of a smoothing leg may walk the level's row blocks in alternating directions: a sweep then starts
 * on the part of the vectors the previous one touched last, which is still in the memory-side cache.  Placement only -- every
 * row is computed exactly as before.  mode 0: always ascending; 1 (default): alternate where one sweep streams more than
 * 640 MB = 2.5x the Infinity Cache (matrix-streaming layouts of large levels: -3...6 % per sweep; smaller levels and the
 * value-free table path measured neutral within +-1 %); 2: always alternate (A/B). */
int sparsh_set_alternate_sweeps(sparsh_handle h, int mode);
/* On a lexicographically ordered grid the aggregation (HEM, src/AMG_coarsening.cpp) joins rows 2J and 2J+1 on most
 * levels.  There the residual kernel of the table path hands each row's residual to the lane next door, adds the pair in the
 * order transfer_residual does and writes the coarse right-hand side and the coarse level's zero-guess sweep directly: the
 * level's residual vector is neither written nor read back and one launch replaces two (parallel::store_residual +
 * parallel::transfer_residual, src/AMG_cycle_utilities.cpp:115-123 and :97-104).  Same expressions in the same order --
 * results are bitwise unchanged.  enable: 1 (default) / 0 (A/B).
 * Box-grid levels (sparsh_level_double_sweep) whose aggregates pair a grid point with its neighbour one line or one plane up get the
 * same fusion from a kernel of their own: one thread per aggregate computes both residuals from the seven-point stencil; neither r nor
 * R is touched.
 * sparsh_level_paired: whether a level of the built hierarchy takes this path under the current configuration: 0 no, 1 row pairs
 * (2J, 2J+1), 2 / 3 box-grid level paired along y / z. */
int sparsh_set_paired_restriction(sparsh_handle h, int enable);
/* Up-leg: with an aggregation prolongator every fine row has exactly one coarse owner, so the last post-smoothing sweep of
 * level l can add its result to the rows of level l - 1 it owns (x_f = 1.0 * x_c + x_f, parallel::transfer_solution,
 * src/AMG_cycle_utilities.cpp:107-112) instead of storing it for a prolongation launch: one launch and one pass over the
 * coarse iterate less per level.  Used where the aggregates hold one or two rows (pairwise matching), on one GPU.  Same
 * expressions -- results are bitwise unchanged.  enable: 1 (default) / 0 (A/B).
 * sparsh_level_prolong_fused: whether level `level`'s last post-sweep prolongates into level - 1 itself: 0 no, 1 yes with
 * the aggregates being the row pairs (2J, 2J+1) (no index read), 2 yes through an 8-byte (first, second) record per aggregate. */
int sparsh_set_fused_prolongation(sparsh_handle h, int enable);
/* Levels whose diagonal is one constant (constant-coefficient stencils and their Galerkin products): the kernels that only
 * divide by d_i -- the zero-guess sweeps x = omega b / d (parallel::jacobi_smoother's first pass, src/AMG_smoothers.cpp:53-76)
 * fused into the PCG update and into the restriction -- take the constant as an argument instead of streaming diag[]
 * (8 of 64 bytes per row of the PCG update).  Same division, same bits.  enable: 1 (default) / 0 (A/B).
 * sparsh_level_constant_diagonal: whether a level of the built hierarchy qualifies under the current configuration, and the value. */
int sparsh_set_constant_diagonal(sparsh_handle h, int enable);
/* Double sweep.  A level whose operator is the 7-point stencil of an nx x ny x nz box grid in lexicographic order (constant
 * coefficients; the finest levels of BASELINE configs[1]'s 3D cases and their Galerkin products) can run two sweeps of
 * parallel::jacobi_smoother (src/AMG_smoothers.cpp:53-76) in one pass over x, b and the result: a workgroup marches through
 * the planes of its tile with the first sweep's plane in LDS, so that sweep's result is never written (temporal blocking;
 * ~30 instead of 48 bytes per row and pair of sweeps).  Every row is computed with the same products in the same order:
 * results are bitwise those of two single sweeps.  mode, read by sparsh_setup: 0 never; 1 (default) on levels of >= 400 000 rows
 * where the setup times it faster than two single sweeps; 2 on every box-grid level that has a launch plan (tests, A/B).
 * After setup the mode may be switched between 0 and its setup value.
 * sparsh_level_double_sweep: on = the level's smoothing legs use it; dims = {nx, ny, nz} (0: not a box grid); plan = {points per
 * thread, lines per tile, planes per chunk}; the setup's timings of two single sweeps / one double sweep in us (0: not timed).
 * Any output pointer may be NULL. */
int sparsh_set_double_sweep(sparsh_handle h, int mode);
/* The launches of a box-grid level that carry an epilogue of their own -- A p with the p.Ap dot (Solver_PCG's mv + cublasDdot,
 * src/AMG_main_solvers.cu), the last post-sweep with the z.r dot or with the prolongation, the residual with the pair restriction --
 * through the same plane-marching scheme with one stencil application per launch (sdia_box1_kernel): x of the current plane in LDS,
 * its neighbours in z in registers, so every x is read once instead of being gathered by seven rows.  The vectors it stores are
 * bitwise the table kernel's; its fused dot products are summed per workgroup of this kernel rather than per 256 rows -- same
 * terms, another order of additions, i.e. a difference in the last bits of a reduction (within the 1e-12 the parity tests hold
 * reductions to).  mode as for the double sweep (0 / 1 timed at setup, default / 2 forced).  sparsh_level_marching_ops: on, plan =
 * {points per thread, lines per tile, planes per chunk}, the setup's timing of the last post-sweep + dot through the table kernel
 * and through this one (us, 0: not timed). */
int sparsh_set_marching_ops(sparsh_handle h, int mode);
/* Double-sweep levels: a smoothing leg that starts from a zero guess (every down-leg below the finest level, and the finest level's
 * inside PCG) runs its first THREE sweeps as one launch: the matrix-free first sweep x = omega b / d (parallel::jacobi_smoother from
 * x = 0) is evaluated where the double sweep would load its input, from the right-hand side the two stencil stages read anyway, so
 * the iterate is not read at all (16 instead of 31 bytes per row for that launch).  Bitwise the three separate sweeps.
 * enable: 1 (default) / 0 (A/B). */
int sparsh_set_zero_start(sparsh_handle h, int enable);
/* PCG (Solver_PCG_*, src/AMG_main_solvers.cu): x += alpha p is applied by the kernel that updates the search direction at the end of
 * the same iteration instead of by the one that updates the residual -- nothing reads x in between and p is then read once for both
 * updates (one n-vector stream less per iteration).  Same expressions, same bits.  enable: 1 (default) / 0 (A/B). */
int sparsh_set_deferred_x(sparsh_handle h, int enable);
int sparsh_level_marching_ops(sparsh_handle h, int level, int *on, int *plan, double *table_us, double *marching_us);
int sparsh_level_double_sweep(sparsh_handle h, int level, int *on, int *dims, int *plan, double *single_us, double *double_us);
int sparsh_level_constant_diagonal(sparsh_handle h, int level, int *is_const, double *value);
int sparsh_level_prolong_fused(sparsh_handle h, int level, int *fused);
int sparsh_level_paired(sparsh_handle h, int level, int *paired);
/* PCG: the x / r update kernel also writes the zero-guess sweep z0 = omega r / d of the V-cycle that follows (same bits, one
 * read of r and one launch less), and streams x, p, Ap and d past the caches (non-temporal loads / stores) so that r and z0,
 * which the cycle's first sweep reads next, are what stays resident.  mode 2 (default): both; 1: fused, ordinary loads;
 * 0: separate launch (A/B). */
int sparsh_set_fused_zero_sweep(sparsh_handle h, int mode);
/* Placement search (single GPU, stencil-table levels whose three sweep vectors together are about the size of the 256 MB
 * Infinity Cache): sparsh_setup times the finest-level sweep on candidate triples among the equally sized buffers the
 * engine owns anyway (plus five spares, freed again) until one runs cache-resident, at most 260 triples (~0.1 s), and lets
 * the fastest triple hold iterate / ping-pong twin / Krylov residual -- the same sweep takes 44 to
 * 63 us depending on the physical pages behind the three vectors.  Pointers only: no extra memory, identical results.
 * sparsh_set_placement_search(h, 0) before sparsh_setup keeps the allocation order (A/B).  sparsh_placement_info: sweep time
 * of the chosen, the worst and the initial triple in microseconds, the number of triples timed (0: search not run) and the
 * seconds the search took. */
int sparsh_set_placement_search(sparsh_handle h, int enable);
int sparsh_placement_info(sparsh_handle h, double *chosen_us, double *worst_us, double *initial_us, int *triples, double *seconds);
/* Multi-GPU setup: by default rank 0 alone runs the host setup (coarsening, Galerkin products, coarse factor) and the other
 * ranks receive the finished hierarchy through the transport (one RCCL broadcast of its byte image, staged through HBM in
 * 256 MB pieces) instead of repeating the identical setup N times; every rank then cuts out and uploads its own row
 * blocks as before.  sparsh_set_setup_broadcast(h, 0) before sparsh_setup: every rank builds its own copy (round-1
 * behaviour; the hierarchies are identical either way -- the setup is deterministic).  Collective: all ranks must use
 * the same setting.  sparsh_setup_share_info: whether this rank built the hierarchy itself, and the image size in bytes
 * (0 when nothing was broadcast). */
int sparsh_set_setup_broadcast(sparsh_handle h, int enable);
int sparsh_setup_share_info(sparsh_handle h, int *built_locally, long *image_bytes);
/* Test hook (host only, after sparsh_setup_host): writes the byte image, reads it back (only the first truncate_to bytes when
 * truncate_to >= 0) and compares every array of the two hierarchies; returns the image size or a negative SPARSH_E*. */
long sparsh_debug_hierarchy_roundtrip(sparsh_handle h, long truncate_to);
int sparsh_level_index16(sparsh_handle h, int level, long *blocks16, long *blocks);
/* Test hook (host only, no handle): builds the row-block schedule and the 16-bit delta form of a CSR pattern, decodes it again
 * and compares with colindex; reports how many row blocks took the 16-bit form. */
int sparsh_debug_index16_roundtrip(int nrow, const int *rowptr, const int *colindex, long *blocks16, long *blocks);
/* Table levels of grid stencils (offsets -1, 0, +1, +-line[, +-plane]) run, on whole-level launches, a
 * variant that stages x[r0 - line, r0 + T + line) of every workgroup's T rows in LDS, so the centre, +-1 and
 * +-line neighbours come out of LDS and only the +-plane neighbours are gathered from L2 (bitwise the same
 * results).  sparsh_set_tile(h, 0) switches it off for the handle (A/B measurements; default on);
 * sparsh_level_tile_rows reports T for a level (0: the variant is not used there). */
int sparsh_set_tile(sparsh_handle h, int enable);
int sparsh_level_tile_rows(sparsh_handle h, int level, int *rows);
/* Multi-GPU diagnostics (collective: every rank calls it with the same arguments): average seconds
 * of one communication step alone, timed with HIP events on the engine's stream.  what = 0: halo
 * exchange of level `level`'s operator; 1: the 16-byte all-reduce of the fused scalars; 2: the
 * all-gather at the partitioned -> replicated boundary.  *avg_seconds = -1 when there is no such
 * step (single GPU, replicated level). */
int sparsh_bench_comm(sparsh_handle h, int what, int level, int reps, double *avg_seconds);
/* name of the kernel the SpMV-type operations of a level launch under the current config
 * ("sdia_tab_kernel", "sdia_kernel", "sell_kernel", "csr_wave_kernel", "csr_block_kernel") */
const char *sparsh_level_kernel(sparsh_handle h, int level);
/* placement the launcher uses for that level's operator under the handle's config: nt = 1 when the
 * matrix stream is read with non-temporal loads, remap = XCD mapping mode (see sparsh_set_kernel_config) */
int sparsh_level_placement(sparsh_handle h, int level, int *nt, int *remap);

/* Host half of sparsh_setup only (coarsening, Galerkin products, coarse factorisation); needs
 * no GPU.  Enables the inspection calls below; solvers still require sparsh_setup. */
int sparsh_setup_host(sparsh_handle h, const sparsh_params *p);

/* hierarchy inspection (tests; "Level k:\t nrow" lines of src/AMG_phases.cpp:55-58) */
int sparsh_num_levels(sparsh_handle h);
int sparsh_level_info(sparsh_handle h, int level, int *nrow, int *nnz, int *p_ncol, int *p_nnz);
/* which: 0 = A_level, 1 = P_level (level < last).  Buffers sized from sparsh_level_info. */
int sparsh_level_csr(sparsh_handle h, int level, int which, int *rowptr, int *colindex, double *val);
/* explicit inverse of the coarsest operator, row-major nL x nL (what the device GEMV applies in
 * place of Direct_Solver_Pardiso_solve); available after sparsh_setup_host when the coarsest level
 * has at most dense_limit rows */
int sparsh_coarse_inverse(sparsh_handle h, double *inv);
/* form of the coarsest-level direct solver: info6 = {rows, dense (1) or factored on the device (0), block size,
 * number of blocks, RCM bandwidth, hierarchy extended past max_levels (1/0)}; *bytes = HBM held by the
 * factors.  After sparsh_setup (block fields are 0 unless the block-tridiagonal form is in use). */
int sparsh_coarse_info(sparsh_handle h, int *info6, long *bytes);
/* Which direct solver a coarsest level above dense_limit rows gets (both replace Direct_Solver_Pardiso,
 * src/AMG_coarse_level_solver.cpp:9-76); call before sparsh_setup.
 *   form 0 (default): nested-dissection multifrontal factorisation -- the operator's graph is dissected recursively, every
 *     tree node's pivot block is inverted explicitly and its couplings to the ancestors are kept as dense blocks; a solve is
 *     one launch per tree level upwards and one downwards (csrc/nd_plan.cpp, nd_solver.cpp, nd_kernels.hip).
 *   form 1: block-tridiagonal factorisation of the RCM-ordered operator (csrc/coarse.cpp), ~ n / bandwidth dependent steps.
 * leaf > 0: largest subgraph kept as one dense block (default 64); merge_rows >= 0: separators of successive bisections are
 * eliminated as one pivot block while their total stays below this (default 384; 0 = plain bisection).
 * sparsh_coarse_nd_info: info6 = {nested dissection in use (1/0), tree nodes, tree levels, largest pivot block,
 * launches per solve, leaf size}. */
int sparsh_set_coarse_form(sparsh_handle h, int form, int leaf, int merge_rows);
/* the same cap for the nodes near the root of the dissection tree: top_merge_rows for the root, halved per tree depth until it
 * meets merge_rows (default 2048; 0 = merge_rows everywhere).  Near the root a tree level holds 1, 2, 4 ... nodes and costs two
 * dependent launches per solve whatever it holds. */
int sparsh_set_coarse_top_merge(sparsh_handle h, int top_merge_rows);
int sparsh_coarse_nd_info(sparsh_handle h, int *info6);
/* Interface form of the block-tridiagonal solve: where the RCM band is narrow against the block (2 * window <= block,
 * window = bandwidth rounded up to 64) only the first / last `window` rows of a block couple to its neighbours, so the
 * chain of dependent steps runs on those rows alone (products S_i^-1 A[i,neighbour] kept in HBM) and everything else is
 * two whole-level launches; the chain itself -- an affine recurrence with constant W x W matrices -- is unrolled at setup
 * into block-triangular matrices, so each of its two passes is ONE triangular matrix-vector product (while those matrices
 * stay under 1 GiB; otherwise one launch per chain step).  *window = 0 when the solver does not use the form.
 * sparsh_set_coarse_interface(h, mode) before sparsh_setup: 0 keeps the plain chain of B x B steps, 1 (default) as
 * described, 2 the interface form with one launch per chain step (A/B measurements). */
int sparsh_coarse_window(sparsh_handle h, int *window);
int sparsh_set_coarse_interface(sparsh_handle h, int enable);
/* Block size of the block-tridiagonal coarse factorisation: rows > 0 before sparsh_setup replaces the built-in rule
 * (rounded up to 64 and never below the RCM bandwidth); 0 restores the rule.  A/B measurements. */
int sparsh_set_coarse_block(sparsh_handle h, int rows);
double sparsh_setup_seconds(sparsh_handle h);

/* AMG_solver::AMG_solve_jacobi(b, x, iterations) (src/AMG_phases.cpp:151-230) ==
 * AMG_GPU1_solver::helper (src/AMG_gpu_phases_2.cu:242-263): host b/x, V-cycles on the device.
 * iterations > 0: exactly that many cycles; -1: until ||Ax-b|| <= tol.
 * hist[k] = residual after cycle k+1 (the value the reference prints). */
int sparsh_vcycle(sparsh_handle h, const double *b, double *x, int iterations, double *hist, int hist_cap, int *ncycles);

/* Same with b/x already in HBM: AMG_GPU1_solver::AMG_Solve (src/AMG_gpu_phases_2.cu:96-240). */
int sparsh_vcycle_dev(sparsh_handle h, const double *b_dev, double *x_dev, int iterations, double *hist, int hist_cap, int *ncycles);

/* The 17 solver entry points of include/AMG.hpp:40-85 reduce to five methods (SPARSH_*).
 * x: initial guess in, solution out.  hist[k] = residual the reference prints at step k. */
int sparsh_solve(sparsh_handle h, int method, const double *b, double *x, double *hist, int hist_cap, int *iters);

/* Same with b/x already in HBM (AMG_GPU1_solver::AMG_Solve / Solver_PCG_4 inner loop,
 * src/AMG_gpu_phases_2.cu:96-240, src/AMG_main_solvers.cu:269-413).  max_iters bounds the
 * loop (<=0: params.max_iter); *seconds returns the loop time measured with HIP events on the
 * engine's stream (NULL to skip). */
int sparsh_solve_dev(sparsh_handle h, int method, const double *b_dev, double *x_dev, int max_iters, double *hist, int hist_cap, int *iters, double *seconds);

/* Stepwise form of the CG / AMG-PCG loop: init = everything before the reference's while loop
 * (src/AMG_main_solvers.cpp:124-133), step = nsteps passes of the loop body (:136-159), stopping
 * early only when ||r|| <= tol.  Lets a caller time exactly k iterations.  *residual = last
 * ||r|| read back; history = residual after each iteration so far. */
int sparsh_krylov_init_dev(sparsh_handle h, int method, const double *b_dev, double *x_dev);
int sparsh_krylov_step_dev(sparsh_handle h, int nsteps, int *done, double *residual);
int sparsh_krylov_history(sparsh_handle h, double *hist, int hist_cap, int *iters);

/* ---- operator-level entry points (kernel parity tests, roofline measurement) ----
 * Host vectors in/out; each runs exactly one device operator of the given level.
 *   spmv      : y = A_l x                      mkl_sparse_d_mv / cusparseDcsrmv (src/AMG_gpu_phase_utilities.cu:143)
 *   jacobi    : sweeps x { x += omega (b - A_l x)/d }   parallel::jacobi_smoother (src/AMG_smoothers.cpp:53-76)
 *   residual  : r = b - A_l x                  parallel::store_residual (src/AMG_cycle_utilities.cpp:115-123)
 *   resnorm   : ||A_l x - b||_2                parallel::residual (src/AMG_cycle_utilities.cpp:83-94)
 *   restrict  : b_{l+1} = P_l^T r              parallel::transfer_residual (src/AMG_cycle_utilities.cpp:97-104)
 *   prolong   : x_l += P_l x_{l+1}             parallel::transfer_solution (src/AMG_cycle_utilities.cpp:107-112)
 *   coarse    : x_L = A_L^{-1} b_L             Direct_Solver_Pardiso_solve (src/AMG_coarse_level_solver.cpp:64-76)
 *   dot/nrm2/axpby : cublasDdot / cublasDnrm2 / daxpby kernel (src/AMG_main_solvers.cu:17-24)
 */
int sparsh_op_spmv(sparsh_handle h, int level, const double *x, double *y);
int sparsh_op_jacobi(sparsh_handle h, int level, const double *b, double *x, int sweeps, int x_is_zero);
int sparsh_op_residual(sparsh_handle h, int level, const double *b, const double *x, double *r);
int sparsh_op_resnorm(sparsh_handle h, int level, const double *b, const double *x, double *nrm);
int sparsh_op_restrict(sparsh_handle h, int level, const double *r, double *bc);
/* levels sparsh_level_paired reports: bc = P_l^T (b - A_l x) and xc = omega bc / d_{l+1} from the one fused launch the
 * V-cycle uses there (store_residual + transfer_residual + the coarse level's first sweep from a zero guess) */
int sparsh_op_residual_restrict(sparsh_handle h, int level, const double *b, const double *x, double *bc, double *xc);
int sparsh_op_prolong(sparsh_handle h, int level, const double *xc, double *xf);
/* levels sparsh_level_prolong_fused reports: xf (level - 1, in/out) += P_{level-1} J(x), J = one Jacobi sweep of `level`
 * from x with right-hand side b -- the one launch the V-cycle's up-leg uses there */
int sparsh_op_jacobi_prolong(sparsh_handle h, int level, const double *b, const double *x, double *xf);
int sparsh_op_coarse(sparsh_handle h, const double *b, double *x);
/* z = V32(r): one application of the opt-in fp32 preconditioner (params.precond_fp32 = 1): a V(nu,nu) cycle from a
 * zero guess on the float copy of the hierarchy, fp64 in/out.  Checked against oracle_vcycle_f32. */
int sparsh_op_precond_f32(sparsh_handle h, const double *r, double *z);
int sparsh_op_dot(sparsh_handle h, int n, const double *x, const double *y, double *out);
int sparsh_op_nrm2(sparsh_handle h, int n, const double *x, double *out);
int sparsh_op_axpby(sparsh_handle h, int n, double a, const double *x, double bcoef, double *y);

/* Time `reps` back-to-back launches of one operator on resident device data with HIP events
 * on the engine's stream; returns average seconds per launch.  op: 0 spmv, 1 fused jacobi
 * sweep, 2 residual, 3 restrict, 4 prolong, 5 coarse GEMV, 6 dot, 7 axpby, 8 int32 copy (4-byte
 * stream, calibrates the profiler's byte counters), 9 fused Jacobi sweeps ping-ponging between two vectors (the
 * access pattern of a smoothing leg), 10 the same on the level's own resident x / x2 / r buffers, 11 double sweeps
 * (sparsh_set_double_sweep) ping-ponging on those buffers: seconds per launch = per PAIR of sweeps. */
int sparsh_bench_op(sparsh_handle h, int op, int level, int reps, double *avg_seconds);

/* device memory helpers so a host language needs no HIP binding of its own */
int sparsh_dev_alloc(sparsh_handle h, long nbytes, void **out);
int sparsh_dev_free(sparsh_handle h, void *p);
int sparsh_dev_fill(sparsh_handle h, double *dst_dev, long n, double value); /* on the engine's stream (thrust::fill of the reference) */
int sparsh_h2d(sparsh_handle h, void *dst_dev, const void *src, long nbytes);
int sparsh_d2h(sparsh_handle h, void *dst, const void *src_dev, long nbytes);
int sparsh_sync(sparsh_handle h);

/* Per-kernel-class device time of the last sparsh_solve_dev call when profiling is enabled
 * (sparsh_profile(h,1)): HIP-event time of every fine-level fused Jacobi sweep launch.
 * out[0] = launches, out[1] = total seconds, out[2] = rows, out[3] = nnz of that level. */
int sparsh_profile(sparsh_handle h, int enable);
int sparsh_profile_read(sparsh_handle h, double *out4);

/* ---- multi-GPU (new design; the reference is single-GPU).  One process per GPU.  Every level
 * above params.replicate_rows is split into contiguous row blocks, one per rank; a smoothing leg
 * exchanges the ghost layers of its vectors once (deep halo, see sparsh_set_deep_halo), every other
 * SpMV-type kernel the neighbours' boundary entries of its input vector, over RCCL (grouped
 * ncclSend/ncclRecv, xGMI); fused scalars are all-reduced; smaller levels and the coarsest solve
 * are computed by every rank.  All ranks run the same host setup on the whole matrix.
 * Bootstrap: rank 0 calls sparsh_comm_unique_id, the 128 bytes travel by any side channel
 * (e.g. a torch.distributed broadcast), every rank calls sparsh_comm_init_rccl BEFORE
 * sparsh_setup.  After setup, vectors passed to the *_dev entry points are the rank's own block
 * [lo, hi) of level 0 (sparsh_local_range). */
int sparsh_set_device(int device); /* make `device` current for this thread (call before sparsh_comm_init_rccl) */
int sparsh_comm_unique_id(char id128[128]);
int sparsh_comm_init_rccl(sparsh_handle h, const char id128[128], int rank, int nranks);
int sparsh_local_range(sparsh_handle h, int level, int *lo, int *hi, int *replicated);
/* Overlap the halo exchange with computation (default off): the exchange is issued on a second
 * stream while the 64-row slices that reference no halo column are processed; the boundary slices
 * follow once the halo has landed.  Same arithmetic, same results.  May be toggled between solves. */
int sparsh_set_overlap(sparsh_handle h, int enable);

/* Deep-halo (communication-avoiding) smoothing on the partitioned levels, default ON; read by sparsh_setup.  Every rank
 * holds, next to its own rows, K = sweeps + 1 layers of ghost rows of the level operator; a smoothing leg exchanges the
 * ghost layers of its right-hand side and of its iterate ONCE and then sweeps a shrinking set of rows (sweep s updates the
 * own rows and the layers <= K - s), so the own rows and the first ghost layer end up exactly as in the global sweep and
 * the residual needs no exchange either: ~13 transport calls per AMG-PCG iteration instead of ~49 with three partitioned
 * levels.  Same arithmetic per row (bitwise the one-rank results for a fixed number of cycles).  0 restores one halo
 * exchange in front of every sweep (the overlap mode below applies to that schedule only).
 * sparsh_exchanges_issued: transport calls (halo or ghost-layer exchanges) this handle has issued so far. */
int sparsh_set_deep_halo(sparsh_handle h, int enable);
/* Inspection / test hooks of a deep-halo level of this rank: info4 = {K (0: not a deep level), local rows (own + padding +
 * ghost layers <= K-1), local columns (layers <= K), first ghost index}; layer_end(d) = local indices in layers 0..d;
 * prefix_spmv: y[0, rows) = rows [0, rows) of the rank-local operator times x_ext (local columns), no exchange -- lets
 * a test run every kernel family over every row prefix a smoothing leg launches. */
int sparsh_deep_info(sparsh_handle h, int level, int *info4);
int sparsh_deep_layer_end(sparsh_handle h, int level, int d, int *end);
int sparsh_deep_prefix_spmv(sparsh_handle h, int level, int rows, const double *x_ext, double *y);
long sparsh_exchanges_issued(sparsh_handle h);

/* In-process transport for tests: nranks handles driven by nranks host threads on one GPU. */
int sparsh_comm_group_create(int nranks, void **group);
void sparsh_comm_group_destroy(void *group);
int sparsh_comm_init_group(sparsh_handle h, void *group, int rank);
/* fault injection for tests: from its ncalls-th halo exchange on (counted per rank, 0-based) the in-process
 * transport fails on every rank.  Solvers must then return SPARSH_ECOMM, and the handle keeps
 * returning it (sticky) until sparsh_setup is called again.  ncalls < 0 switches the hook off. */
/* Multi-rank schedule chosen from measurements (round 3).  With more than one rank and replicate_rows <= 0 (the default)
 * sparsh_setup first measures the transport it was given -- one neighbour exchange (latency and rate), the 16-byte all-reduce,
 * one all-gather -- and the device's sweep floor and streaming rate, takes the slowest rank's numbers, models every level's
 * share of a V(nu,nu) cycle as partitioned with deep halos / partitioned with one exchange per sweep / replicated, and
 * partitions the prefix of levels (with one smoothing schedule) that minimises the total.  replicate_rows > 0 or
 * sparsh_set_comm_tuning(h, 0) keep the caller's threshold and sparsh_set_deep_halo.
 * sparsh_comm_schedule: info4 = {rows, boundary rows of a middle rank, partitioned, deep halo}, cost_us3 = modelled microseconds
 * per V-cycle {deep halo, exchange per sweep, replicated}.  sparsh_comm_measured: m7 = {exchange us, exchange us/MB, all-reduce us,
 * all-gather us, all-gather us/MB, sweep floor us, sweep us/MB}.
 * sparsh_comm_group_set_delay (in-process test transport only): every transport call occupies the caller's stream that many
 * microseconds first -- a slow link, to see the schedule move. */
int sparsh_set_comm_tuning(sparsh_handle h, int mode);
/* host-only what-if (after sparsh_setup_host; no device, no transport): the schedule the tuner would choose for `nranks` ranks from the
 * seven numbers m7 (layout of sparsh_comm_measured); read it back with sparsh_comm_schedule */
int sparsh_plan_comm_schedule(sparsh_handle h, int nranks, const double *m7);
int sparsh_comm_schedule(sparsh_handle h, int level, int *info4, double *cost_us3);
int sparsh_comm_measured(sparsh_handle h, double *m7);
int sparsh_comm_group_set_delay(void *group, double microseconds);
int sparsh_comm_group_fail_after(void *group, int ncalls);

/* Host-only planning query (after sparsh_setup_host): the block of operator `which` (0 A_l, 1 P_l,
 * 2 R_l) that `rank` of `nranks` holds when every level is partitioned, with its halo plan.
 * sizes8 = {rows, nnz, own input entries, halo entries, send segments, recv segments,
 * packed send entries, first global row}.  The _get call copies the arrays of the last query:
 * local CSR (columns renumbered: own entries first, then halo), global index of every halo
 * entry, indices to pack, and (peer, offset, count) triples of the send / receive segments. */
int sparsh_dist_local_op(sparsh_handle h, int level, int which, int rank, int nranks, int *sizes8);
int sparsh_dist_local_op_get(sparsh_handle h, int *rowptr, int *col, double *val, int *halo_global, int *send_idx, int *send_segs3,
                             int *recv_segs3);

/* Host-only planning query of the deep-halo layout (after sparsh_setup_host): the block of A_level that `rank` of `nranks`
 * holds with K ghost layers, and its exchange plan of the given depth.  sizes8 = {local rows (own + padding + layers <= K-1),
 * nnz, own rows, first ghost index (own rows rounded up to 64), local indices (all layers), send segments, receive segments,
 * packed send entries}.  _get copies: local CSR (local column numbering), global index of every local index (-1: padding),
 * layer_end[0..K], own indices to pack, (peer, offset, count) of the send segments, local position of every received entry,
 * (peer, offset, count) of the receive segments. */
int sparsh_dist_deep_op(sparsh_handle h, int level, int rank, int nranks, int K, int depth, int *sizes8);
int sparsh_dist_deep_op_get(sparsh_handle h, int *rowptr, int *col, double *val, int *global_of, int *layer_end, int *send_idx,
                            int *send_segs3, int *recv_pos, int *recv_segs3);

#ifdef __cplusplus
}
#endif
#endif
