// nd_solver.hpp -- device side of the nested-dissection multifrontal coarse solver (plan: nd_plan.hpp).
// Stands where the reference calls PARDISO (src/AMG_coarse_level_solver.cpp:9-62 analyse + factor, :64-76 solve).
#pragma once

#include <hip/hip_runtime_api.h>

#include <memory>
#include <string>
#include <vector>

#include "nd_plan.hpp"

namespace sparsh {

struct NdDevNode {
    long long foff, boff, loff, ioff;
    int first, np, nu, rel;  // rel: offset of this node's row positions in its parent's front (rel_idx)
    int parent, pad;
};

struct NdGemm {  // C (M x N) = alpha * A (M x K) * B (K x N) + (beta ? C : 0), all row-major
    const double *a, *b;
    double *c;
    int lda, ldb, ldc, M, N, K;
    double alpha;
    int beta, tiles_n;
};

struct NdGjNode {  // one pivot block of the batched whole-chip inversion (nd_kernels.hip)
    double *a, *b;     // the block itself (in its front) and a scratch copy: the steps ping-pong between them
    double *c0, *c1;   // column magnitudes for the pivot search, ping-pong
    double *out;       // B_k
    int *piv, *cmap;
    int p, lda, ldb, ldo, wg0;  // wg0: first workgroup of this node in the launch
};

// transfer_solution fused into the backward pass (aggregation prolongator: every fine row has one owner): x_f = 1.0 * x_c + x_f for the (at most
// two) fine rows of coarse row J -- members[2J], members[2J + 1] (-1: none), or rows 2J, 2J + 1 when members is null
struct NdProlong {
    double *xf = nullptr;
    const int *members = nullptr;
    int nfine = 0;
};

class NdSolver {
public:
    ~NdSolver() { release(); }
    NdSolver() = default;
    NdSolver(const NdSolver &) = delete;
    NdSolver &operator=(const NdSolver &) = delete;

    // host half alone (no device): dissect, size the factors; false when the operator is refused.  The plan is kept for setup().
    bool plan(const HostCsr &A, const NdParams &prm, std::string &err);
    // why_failed: 1 the plan was refused (graph / memory limits), 2 singular, 3 device error
    bool setup(const HostCsr &A, const NdParams &prm, hipStream_t st, std::string &err, int *why_failed);
    // device vectors, the operator's own numbering; pr: the backward pass also adds every x it produces to the rows of the finer level that row owns
    void solve(const double *b, double *x, hipStream_t st, NdProlong pr = NdProlong()) const;
    void release();

    bool ready() const { return n_ > 0; }
    int n() const { return n_; }
    int nlevels() const { return nlevels_; }
    int nnodes() const { return nnodes_; }
    int leaf() const { return leaf_; }
    int max_pivot_rows() const { return max_np_; }
    int launches_per_solve() const { return launches_; }
    size_t bytes() const { return factor_bytes_; }
    double factor_seconds = 0.0, plan_seconds = 0.0;

private:
    std::unique_ptr<NdPlan> pending_;  // plan() ran ahead of setup()
    const void *pending_key_ = nullptr;
    int n_ = 0, nlevels_ = 0, nnodes_ = 0, leaf_ = 0, max_np_ = 0, launches_ = 0;
    size_t factor_bytes_ = 0;
    std::vector<void *> allocs_;
    double *Bm_ = nullptr, *Lf_ = nullptr, *w_ = nullptr;  // backward rows, forward rows (per target row), [c | x] in the new numbering
    int *bidx_ = nullptr, *fidx_ = nullptr;
    struct Pass {
        NdRow *rows = nullptr;
        int nrows = 0, nwide = 0;
    };
    std::vector<Pass> fwd_, bwd_;  // per tree level
};

// launchers of nd_kernels.hip
void nd_launch_scatter(long long cnt, const long long *dst, const double *val, double *fronts, hipStream_t st);
void nd_launch_extend_add(const NdDevNode *nodes, const int *children, int nchildren, int max_nu, const int *rel_idx, double *fronts, hipStream_t st);
void nd_launch_invert(const NdDevNode *nodes, const int *list, int count, const double *fronts, double *Bm, int *singular, hipStream_t st);
void nd_launch_gj_batched(const NdGjNode *nodes, int nnodes, const int *wg_node, int nwg, int max_p, int *singular, hipStream_t st);
void nd_launch_gemm(const NdGemm *problems, const int *tiles, int ntiles, hipStream_t st);
void nd_launch_repack(long long nseg, const NdSegment *segs, const double *Lh, double *Lf, hipStream_t st);
void nd_launch_pass(bool forward, const NdRow *rows, int nrows, int nwide, int n, const double *M, const int *idx, double *w, const double *b, double *x,
                    hipStream_t st, NdProlong pr = NdProlong());
constexpr int kNdTinyPivot = 80;     // pivot blocks up to this many rows are inverted by one workgroup in LDS (80 x 81 doubles = 51 KB);
                                     // larger ones by the batched whole-chip inversion

}  // namespace sparsh
