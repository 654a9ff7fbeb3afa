// engine.hpp -- device-resident AMG hierarchy + V-cycle + Krylov loops (single GPU).
// Mirrors the behaviour of AMG_GPU1_solver ("MI": whole hierarchy resident,
// src/AMG_gpu_phases_2.cu:13-240) with the arithmetic of the CPU path
// (src/AMG_phases.cpp:151-230, src/AMG_main_solvers.cpp:47-458).
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "../../include/sparsh_amg.h"
#include "coarse.hpp"
#include "comm.hpp"
#include "dist.hpp"
#include "host_setup.hpp"
#include "kernels.hpp"

namespace sparsh {

struct DevLevel {
    // ---- deep-halo layout of a partitioned level (communication-avoiding smoothing, dist.hpp): the operator A
    // and the vectors x, x2, b carry the own rows, padding to a slice boundary and K ghost layers
    bool deep = false;
    int K = 0;                       // ghost layers (sweeps + 1)
    int npad = 0;                    // own rows rounded up to 64 = first ghost index
    std::vector<int> layer_end;      // layer_end[d] = local indices in layers 0..d
    DevDeepPlan dplan[3];            // exchange of ghost layers <= 1, <= K-1, <= K
    std::vector<int> blk_first, wblk_first;  // first row of every CSR row block / wave block (prefix launches)
    double *xc_stage = nullptr;      // prolongation input [own x_{l+1} | planP halo] (x_{l+1}'s own tail holds A's ghosts)
    double *b_ext = nullptr;         // level 0: the right-hand side with room for its ghost layers
    int n = 0;       // rows this rank holds (all of them on a replicated level)
    int nglob = 0;   // rows of the level
    bool replicated = true;  // every rank holds and computes the whole level
    DevPlan planA, planP, planR;  // halo plans of A_l x, P_l x_{l+1}, R_l r_l (empty on one GPU)
    DevCsr A, P, R;
    bool P_is_aggregation = false;
    int *members = nullptr;        // aggregates of at most two rows that are not such pairs: (first, second or -1) per coarse row, for the
                                   // coarser level's last post-sweep to prolongate into this level itself (OP_JACOBI_PROLONG)
    int pair_axis = 0;             // box-grid level whose aggregate J (lexicographic in the coarse box; < 0: from its far end) = grid point + its neighbour one line (1) /
                                   // one plane (2) up: residual + restriction without r and R (box_resid_pair_kernel)
    bool pair_aggregates = false;  // aggregate J = fine rows (2J, 2J+1) in R's stored order: residual + restriction fuse (OP_RESID_PAIR)
    double *diag = nullptr;
    double box1_table_us = 0.0, box1_us = 0.0;        // setup timing of the last post-sweep + dot: table kernel / plane-marching kernel
    double box_single_us = 0.0, box_double_us = 0.0;  // setup timing of two single sweeps / one double sweep (tune_box_kernels), 0 = not timed
    bool diag_is_const = false;  // every (own) row has the same diagonal entry, diag_const
    double diag_const = 0.0;
    double *x = nullptr, *x2 = nullptr;  // ping-pong solution buffers (Jacobi reads old, writes new)
    double *b = nullptr;                 // rhs of this level (level 0: points at the caller's vector)
    double *r = nullptr;                 // residual
    bool fine = false;                   // finest level of the hierarchy
};

struct KrylovState {
    bool active = false, precond = false;
    const double *b = nullptr;
    double *x = nullptr;
    int count = 0;
    double r1 = 0.0;
};

// float mirror of one level for the opt-in fp32 preconditioner
struct F32Level {
    SdiaF32 A;
    float *csr_val = nullptr;  // float copy of the CSR values on levels without the sliced-diagonal mirror
    float *diag = nullptr, *x = nullptr, *x2 = nullptr, *b = nullptr, *r = nullptr;
};

struct GraphState {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    double *x = nullptr;  // solution vector the captured iteration updates
    bool precond = false;
};

// HIP-event timing of the dominant kernel inside a solve: one event pair brackets each RUN of consecutive
// fused Jacobi launches on the finest level (the 6-7 sweeps of a smoothing leg), so the cost of the event
// packets themselves is spread over the run instead of being charged to every launch (an event pair
// around each single launch reads ~10 us high).
struct ProfileData {
    bool enabled = false;
    std::vector<hipEvent_t> ev;      // pairs
    std::vector<int> run_launches;   // launches bracketed by pair k
    size_t used = 0;
    double launches = 0, seconds = 0;
};

class Engine {
public:
    Engine(int nrow, int ncol, const int *rowptr, const int *col, const double *val);
    ~Engine();

    int setup_host(const sparsh_params &p);  // hierarchy on the host only (no device needed)
    // multi-GPU: rank 0 runs setup_host and the other ranks receive its hierarchy through the transport instead of
    // repeating the same setup (set_share_setup(false): every rank builds its own, as in round 1)
    int setup_host_shared(const sparsh_params &p);
    void set_share_setup(bool on) { share_setup_ = on; }
    bool built_locally() const { return built_locally_; }
    size_t shared_image_bytes() const { return image_bytes_; }
    int setup(const sparsh_params &p);       // setup_host + upload to HBM
    bool ready() const { return ready_; }
    bool host_ready() const { return host_ready_; }
    int nlevels() const { return (int)lev_.size(); }
    const HostHierarchy &host() const { return H_; }
    const CoarseSolver &coarse() const { return coarse_; }
    CoarseSolver &coarse_mut() { return coarse_; }
    const sparsh_params &params() const { return prm_; }
    // kernel-family / layout choices of THIS handle (const_slots is read when the layouts are built)
    // (mutable access is for the C ABI's setters, which call config_changed() afterwards)
    KernelConfig &kernel_cfg() { return cfg_; }
    const KernelConfig &kernel_cfg() const { return cfg_; }
    // a captured hipGraph of the iteration replays the kernels of the configuration it was captured under: drop it
    void config_changed() { drop_graph(); }
    // diag[] of a level for the kernels that only divide by it: nullptr (+ diag_const) where it is one constant
    const double *diag_stream(const DevLevel &L) const { return cfg_.const_diag && L.diag_is_const ? nullptr : L.diag; }
    // a smoothing leg of this level that starts from a zero guess runs sweeps 1 - 3 as one launch reading b alone
    bool zero_start(const DevLevel &L) const { return cfg_.zero_start && (!dist_ || L.replicated) && !L.deep && box2_applies(L.A, cfg_); }
    // whether level l's last post-sweep also prolongates into level l - 1 (OP_JACOBI_PROLONG)
    bool level_prolong_fused(int l) const
    {
        if (l < 1 || l + 1 >= (int)lev_.size() || !cfg_.fuse_prolong || prm_.sweeps < 1 || prm_.precond_fp32) return false;
        const DevLevel &F = lev_[l - 1];
        if (dist_ && !F.replicated) return false;  // several GPUs: between replicated levels only
        return F.P_is_aggregation && !F.deep && (F.pair_aggregates || F.members) && F.R.nrow == lev_[l].n;
    }
    // whether level l's residual, restriction and the next level's zero-guess sweep run as one launch (OP_RESID_PAIR)
    // 0 no; 1 aggregates = row pairs (2J, 2J+1): OP_RESID_PAIR of the table kernel; 2 / 3 box-grid level paired along y / z
    int level_paired(int l) const
    {
        if (!(l + 2 < (int)lev_.size() && prm_.sweeps > 0 && (!dist_ || lev_[l].replicated) && cfg_.pair_restrict)) return 0;
        if (lev_[l].pair_aggregates && resid_pair_applies(lev_[l].A, cfg_)) return 1;
        if (lev_[l].pair_axis != 0 && lev_[l].A.box_nx > 0 && csr_family(lev_[l].A, cfg_) == FAM_SDIA_TAB) return 1 + std::abs(lev_[l].pair_axis);
        return 0;
    }
    // average seconds of one communication step alone (collective: every rank calls it): what = 0 halo
    // exchange of level `level`'s operator, 1 the 16-byte all-reduce of the fused scalars, 2 the
    // all-gather at the partitioned -> replicated boundary.  -1 when the step does not exist.
    double bench_comm(int what, int level, int reps);
    hipStream_t stream() const { return st_; }
    const DevLevel &level(int l) const { return lev_[l]; }
    int n0() const { return A0_.nrow; }
    int local_n0() const { return lev_.empty() ? A0_.nrow : lev_[0].n; }
    // multi-GPU: install the transport before setup(); the engine owns it
    void set_comm(std::unique_ptr<Comm> c) { comm_ = std::move(c); }
    void set_overlap(bool on) { overlap_ = on; }
    // deep halo (default on for partitioned runs): one exchange per smoothing leg instead of one per sweep; read at setup
    void set_deep_halo(bool on) { deep_halo_ = on; }
    bool deep_halo() const { return deep_halo_; }
    // Multi-rank setup: measure the transport (neighbour exchange latency and rate, 16-byte all-reduce, all-gather) and the
    // device's sweep rate, then choose from those numbers which levels are partitioned and whether the partitioned levels smooth
    // with deep halos or exchange per sweep (mode 1, default).  Mode 0: replicate_rows / set_deep_halo decide, as in round 2.
    void set_comm_tuning(int mode) { comm_tune_ = mode; }
    struct CommLevelChoice {
        int rows = 0, halo_rows = 0;
        bool partitioned = false, deep = false;
        double cost_deep_us = 0.0, cost_per_sweep_us = 0.0, cost_replicated_us = 0.0;  // modelled time of the level's part of one V-cycle
    };
    struct CommMeasured {
        double exchange_us = 0.0, exchange_us_per_mb = 0.0, allreduce_us = 0.0, allgather_us = 0.0, allgather_us_per_mb = 0.0;
        double sweep_floor_us = 0.0, sweep_us_per_mb = 0.0;
        bool valid = false;
    };
    // host-only: the schedule the tuner would choose for `nranks` ranks from the given measurements (tests, what-if tables)
    void plan_comm_schedule(const sparsh_params &p, int nranks, const CommMeasured &m)
    {
        meas_ = m;
        meas_.valid = true;
        sched_.clear();
        tuned_repl_level_ = -1;
        decide_comm_schedule(p, nranks);
    }
    int tuned_partitioned_levels() const { return tuned_repl_level_; }
    const std::vector<CommLevelChoice> &comm_schedule() const { return sched_; }
    const CommMeasured &comm_measured() const { return meas_; }
    long exchanges_issued() const { return n_exchanges_; }
    // test hook: y[0, rows) = (A_l x_ext)[0, rows) on the local operator of a deep-halo level, no exchange
    bool debug_prefix_spmv(int l, int rows, const double *x_ext, double *y);
    bool overlap() const { return overlap_; }
    Comm *comm() const { return comm_.get(); }
    bool distributed() const { return dist_; }
    bool precond_fp32_active() const { return f32_ready_; }
    const Partition &partition(int l) const { return parts_[l]; }
    void set_stopping(double tol, int max_iter, int check_every)
    {
        prm_.tol = tol;
        if (max_iter > 0) prm_.max_iter = max_iter;
        if (check_every > 0) prm_.check_every = check_every;
    }

    // solvers on device vectors; return SPARSH_* code; iterations through *iters
    int solve_dev(int method, const double *b_dev, double *x_dev, int max_iters, double *hist, int hist_cap, int *iters,
                  double *seconds);
    int amg_solve_dev(const double *b_dev, double *x_dev, int iterations, double *hist, int hist_cap, int *ncycles);

    // stepwise CG / AMG-PCG (bench: time exactly k iterations)
    int pcg_init(const double *b_dev, double *x_dev, bool precond);
    int pcg_steps(int nsteps, int *done);
    int krylov_hist(double *hist, int hist_cap);  // copies the residual history, returns iterations so far
    double krylov_residual() const { return ks_.r1; }

    // operator-level (device pointers)
    void op_spmv(int l, const double *x, double *y);
    void op_jacobi(int l, const double *b, double *x, double *tmp, int sweeps, bool x_is_zero);  // result in x
    void op_residual(int l, const double *b, const double *x, double *r);
    double op_resnorm(int l, const double *b, const double *x);
    // fuse_zero: also write the coarse level's zero-guess sweep (aggregation P, no gather step); returns whether it did
    bool op_restrict(int l, const double *r, double *bc, bool fuse_zero = false);
    // level_paired(l) levels: b_{l+1} = R (b - A x) and x_{l+1} = omega b_{l+1} / d_{l+1} in one launch
    void op_residual_restrict(int l, const double *b, const double *x, double *bc, double *xc);
    // level_prolong_fused(l) levels: one sweep of level l from x whose result is added to xf (level l - 1) instead of stored
    void op_jacobi_prolong(int l, const double *b, const double *x, double *xf);
    void op_prolong(int l, const double *xc, double *xf);
    void op_coarse(const double *b, double *x);
    // z = V32(r): one application of the opt-in fp32 preconditioner (fp64 in/out); needs precond_fp32
    bool op_precond_f32(const double *r, double *z);
    double op_dot(int n, const double *x, const double *y);

    // device memory helpers
    void *dalloc(size_t bytes);
    void dfree(void *p);
    bool check(hipError_t e, const char *what);
    std::string error;
    // Sticky fault of the solve path: the first failed HIP call (SPARSH_ENODEV) or transport step
    // (SPARSH_ECOMM) since setup.  Solvers stop at their next check and return it; every later
    // call on the handle returns it too until sparsh_setup is called again.
    int fault() const { return fault_; }
    bool note_hip(hipError_t e, const char *what);
    bool note_comm(bool ok, const char *what);

    ProfileData prof;
    void profile_begin();    // (re)arm the event pool
    void profile_collect();  // sync and sum the recorded launch times
    double setup_seconds = 0.0;
    // placement search of the last setup: sweep time of the chosen / the worst / the initial triple (us), triples tried (0: not run)
    double place_best_us = 0.0, place_worst_us = 0.0, place_first_us = 0.0;
    int place_tried = 0;
    double place_seconds = 0.0;

private:
    // one V(nu,nu) cycle: rhs b0 (device, read-only), solution accumulates in lev_[0].x.
    // x0_zero: the initial guess of level 0 is zero (preconditioner use).
    // dot_partial: when non-null the last post-sweep also leaves partial sums of x.b there.
    void vcycle(const double *b0, bool x0_zero, double *dot_partial, int *dot_nblk, bool zero_done0 = false);
    // zero_done: the zero-guess sweep x = omega*b/d has already been written to L.x (fused into the restriction)
    // prolong_to: the finer level whose iterate the last sweep adds its result to (level_prolong_fused), nullptr = store it
    void smooth(DevLevel &L, const double *b, int sweeps, bool x_zero, double *dot_partial, int *dot_nblk, bool zero_done = false,
                DevLevel *prolong_to = nullptr);
    bool halo(const DevPlan &p, double *vec);  // pack + exchange (no-op on one GPU)
    // deep-halo level: fill the ghost layers <= depth (which: 0 depth 1, 1 depth K-1, 2 depth K) of vec from their owners
    bool deep_exchange(DevLevel &L, int which, double *vec);
    // launch an A_l-type operator on the first `rows` local rows of a deep level (no exchange)
    int launch_prefix(DevLevel &L, int rows, CsrOp op, const CsrArgs &a);
    bool upload_deep_plan(const DeepPlan &h, DevDeepPlan &d);
    // A_l-type operator on level L: exchanges the halo of a.x, then launches; with overlap enabled the
    // exchange runs on a second stream while the slices that touch no halo column are processed.
    // Returns the number of reduction partials written.
    int apply_A(DevLevel &L, CsrOp op, CsrArgs a);
    void finalize(Fin code, const double *p0, const double *p1, int nblk, int slot, double *hist, int it, int nblk1 = -1);
    bool upload_plan(const HaloPlan &h, DevPlan &d);
    double read_scalar(int slot);
    double read_hist(int it);

    bool setup_f32();
    // V(nu,nu) cycle on the float hierarchy from a zero guess: z64 = V32(r64), partial sums of z.r
    void vcycle_f32(const double *r64, double *z64, double *partial, int *nblk);
    void pcg_body(bool precond, int slot);
    bool capture_graph(bool precond);
    void drop_graph();
    int pcg(const double *b, double *x, int max_iters, double *hist, int hist_cap, int *iters, bool precond);
    int bicg(const double *b, double *x, int max_iters, double *hist, int hist_cap, int *iters, bool precond);

    std::unique_ptr<Comm> comm_;
    std::vector<Partition> parts_;  // row partition of every level
    Partition gather_part_;         // share of the first replicated level each rank restricts, then all-gathers
    int repl_level_ = 0;            // first replicated level (0: nothing is partitioned)
    bool dist_ = false;
    bool overlap_ = false;          // multi-GPU: overlap halo exchange with interior slices
    bool deep_halo_ = true;         // multi-GPU: deep-halo smoothing on the partitioned levels
    int comm_tune_ = 1;
    int tuned_repl_level_ = -1;     // >= 0: number of partitioned levels chosen by tune_comm_schedule
    bool tuned_deep_ = true;
    std::vector<CommLevelChoice> sched_;
    CommMeasured meas_;
    bool measure_transport();
    void tune_comm_schedule(const sparsh_params &p);
    void decide_comm_schedule(const sparsh_params &p, int G);
    long n_exchanges_ = 0;          // transport calls issued (halo / staged exchanges; diagnostics)
    hipStream_t st2_ = nullptr;     // exchange stream of the overlap path
    hipEvent_t ev_ready_ = nullptr, ev_halo_ = nullptr;
    std::vector<F32Level> f32_;
    float *coarse_inv_f32_ = nullptr;
    bool f32_ready_ = false;
    KrylovState ks_;
    HostCsr A0_;
    HostHierarchy H_;
    sparsh_params prm_{};
    KernelConfig cfg_;
    std::vector<DevLevel> lev_;
    CoarseSolver coarse_;  // coarsest-level direct solver (dense inverse or block-tridiagonal factors)
    double *coarse_tmp_ = nullptr;  // fp32 mode with the block-tridiagonal form: fp64 staging of b_L, x_L
    int nL_ = 0;
    hipStream_t st_ = nullptr;
    bool ready_ = false;
    bool host_ready_ = false;
    bool share_setup_ = true, built_locally_ = true;
    size_t image_bytes_ = 0;
    int fault_ = 0;
    int device_ = 0;

    // workspace
    double *scal_ = nullptr;      // S_COUNT device scalars
    double *part0_ = nullptr, *part1_ = nullptr;
    int part_cap_ = 0;
    double *hist_dev_ = nullptr;
    int *iter_ctr_ = nullptr;     // device-side residual-history index (graph replays)
    GraphState graph_;
    int hist_cap_dev_ = 0;
    double *pinned_ = nullptr;    // host-pinned staging for scalar read-back
    std::vector<double *> work_;  // level-0 work vectors of the Krylov loops
    std::vector<void *> allocs_;
    // which of the finest level's equally sized buffers play iterate / ping-pong twin / Krylov residual: chosen at setup by
    // timing the sweep on the candidates (see tune_placement)
    void tune_box_kernels();
    void tune_placement();
};

}  // namespace sparsh
