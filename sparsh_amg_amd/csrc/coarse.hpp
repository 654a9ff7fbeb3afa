// coarse.hpp -- coarsest-level direct solver on the device (stands where the reference calls
// Direct_Solver_Pardiso: analyse + factor once, src/AMG_coarse_level_solver.cpp:9-62; one solve per
// V-cycle, :64-76).  Three forms, chosen by the order of the system:
//   n <= dense_limit : explicit dense inverse (host: RCM + banded LU), one GEMV per solve;
//   larger (default) : nested-dissection multifrontal factorisation (nd_plan.hpp, nd_solver.hpp, nd_kernels.hip):
//                      2 x (tree height) dependent launches per solve, factors of O(n^(4/3)) (3D) / O(n log n) (2D);
//   larger (form 1)  : block-tridiagonal ("twisted") factorisation of the RCM-ordered operator, kept as
//                      the explicit inverses of the Schur-complement diagonal blocks in HBM, factored and
//                      applied by the kernels of coarse_kernels.hip (round 2's solver: ~ n / bandwidth dependent steps).
// Either of the last two is what lets the reference's level1 = 6 policy (coarsest level = N/32 rows,
// src/AMG_phases.cpp:51,77) run on the device.
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "host_setup.hpp"
#include "nd_solver.hpp"

namespace sparsh {

struct BtDevCsr {
    int *rp = nullptr;
    int *ci = nullptr;
    double *v = nullptr;
};

// ELL form of a coupling piece: entry k of row g at [k*n + g]; k = 0 when the piece is too ragged for it (a row with more
// than 8 entries: the unstructured FEM level, which then takes the CSR walk -- an ELL of 8 plus overflow measured 2 % slower)
struct BtDevEll {
    int k = 0;
    int *ci = nullptr;
    double *v = nullptr;
};

// Host plan of the block-tridiagonal factorisation (no device needed; inspected by tests).
struct BtPlan {
    int n = 0, bw = 0, B = 0, nb = 0, mid = 0;
    std::vector<int> perm;  // perm[new] = old
    // CSR pieces of the permuted operator (rows and columns in the new numbering):
    //   diag : entries inside the row's own block
    //   out  : entries in the outer neighbour block(s) (towards the chain's end; both neighbours for the middle block)
    //   in   : entries in the inner neighbour block (towards the middle)
    //   inT  : transpose of `in` (row = column index of the entry, column = its row)
    HostCsr diag, out, in, inT;
    int block_rows(int i) const { return i + 1 < nb ? B : n - (nb - 1) * B; }
    size_t sinv_bytes() const { return (size_t)nb * B * B * sizeof(double); }
};

// false when the operator cannot take this form (bandwidth above max_block)
// block_hint > 0: use blocks of that many rows (rounded up to 64, at least the bandwidth) instead of the built-in rule
bool bt_make_plan(const HostCsr &A, int target_blocks, int max_block, BtPlan &plan, std::string &err, int block_hint = 0);

class CoarseSolver {
public:
    ~CoarseSolver() { release(); }
    CoarseSolver() = default;
    CoarseSolver(const CoarseSolver &) = delete;
    CoarseSolver &operator=(const CoarseSolver &) = delete;

    // dense form from a host inverse (row-major n x n)
    bool setup_dense(int n, const double *inv_host, std::string &err);
    // block-tridiagonal form: plan on the host, factor on the device (stream st; synchronises once at the end).
    // why_failed: 1 the operator has no usable band structure, 2 singular, 3 device error
    bool setup_bt(const HostCsr &A, hipStream_t st, std::string &err, int *why_failed = nullptr);
    // host-only check that the configured form accepts the operator (band / separators, memory budget); the nested-dissection
    // plan is kept for setup_nd
    bool probe(const HostCsr &A, std::string &err);
    // nested-dissection multifrontal form (default above dense_limit); why_failed as for setup_bt
    bool setup_nd(const HostCsr &A, hipStream_t st, std::string &err, int *why_failed = nullptr);
    // x = A^-1 b, device vectors in the operator's own numbering; enqueues on st
    void solve(const double *b, double *x, hipStream_t st) const;
    // nested-dissection form only: the solve's backward pass also prolongates into the finer level (NdProlong)
    void solve_prolong(const double *b, double *x, hipStream_t st, NdProlong pr) const { nd_.solve(b, x, st, pr); }
    void release();

    bool ready() const { return n_ > 0; }
    bool dense() const { return dense_; }
    bool nested() const { return nd_.ready(); }
    const NdSolver &nd() const { return nd_; }
    // 0 = nested dissection above dense_limit (default), 1 = block-tridiagonal (round 2's solver, kept for A/B runs)
    void set_form(int form) { form_ = form; }
    int form() const { return form_; }
    void set_nd_params(int leaf, int merge_rows, int top_merge_rows = -1)
    {
        if (leaf > 0) nd_prm_.leaf = leaf;
        if (merge_rows >= 0) nd_prm_.merge_rows = merge_rows;
        if (top_merge_rows >= 0) nd_prm_.top_merge_rows = top_merge_rows;
    }
    const NdParams &nd_params() const { return nd_prm_; }
    int n() const { return n_; }
    int block() const { return plan_.B; }
    int nblocks() const { return plan_.nb; }
    int bandwidth() const { return plan_.bw; }
    bool windowed() const { return windowed_; }
    int window() const { return win_; }
    void set_block_hint(int rows) { block_hint_ = rows; }  // before setup_bt; 0 = built-in rule
    void set_allow_windowed(bool on) { allow_windowed_ = on; }  // before setup_bt (A/B measurements)
    void set_unrolled_chain(bool on) { unroll_chain_ = on; }  // interface form: chain passes as triangular products (default) or step by step
    bool unrolled() const { return unrolled_; }
    const double *dense_inverse() const { return inv_; }
    size_t bytes() const
    {
        if (nd_.ready()) return nd_.bytes();
        return dense_ ? (size_t)n_ * n_ * 8 : plan_.sinv_bytes() + (windowed_ ? (size_t)2 * plan_.nb * plan_.B * win_ * 8 : 0) + tri_bytes_;
    }
    double factor_seconds = 0.0;

private:
    NdSolver nd_;
    NdParams nd_prm_;
    int form_ = 0;
    bool allow_windowed_ = true, unroll_chain_ = true;
    int block_hint_ = 0;
    int n_ = 0;
    bool dense_ = true;
    double *inv_ = nullptr;  // dense form
    BtPlan plan_;            // (pieces dropped after upload; sizes kept)
    double *sinv_ = nullptr, *z_ = nullptr;
    int *perm_ = nullptr;
    BtDevCsr out_, in_;
    BtDevEll out_ell_, in_ell_;
    // interface ("window") form of the solve for narrow bands (2 * window <= block): only the first / last `win_` rows
    // of a block couple to its neighbours, so the dependent chain runs on those rows alone, with the products
    // G_i = S_i^-1 A[i,outer] and H_i = S_i^-1 A[i,inner] (restricted to the neighbour's window columns) kept in HBM
    bool windowed_ = false;
    int win_ = 0;                      // window width W (multiple of 64)
    double *g_ = nullptr, *h_ = nullptr;  // nb x B x W each (row-major per block, leading dimension W)
    double *y_ = nullptr, *xw_ = nullptr;
    int *wdesc_ = nullptr;             // per block: {gwin0, gwn, hwin0, hwn}
    // the two chains (0 top, 1 bottom) unrolled: L_ (m W x m W) and U_ ((m + 1) W x (m + 1) W) per chain, position -> row index maps
    bool unrolled_ = false;
    double *L_[2] = {nullptr, nullptr}, *U_[2] = {nullptr, nullptr};
    int *in_idx_[2] = {nullptr, nullptr}, *out_idx_[2] = {nullptr, nullptr};
    int chain_m_[2] = {0, 0};
    size_t tri_bytes_ = 0;
    std::vector<void *> allocs_;
    struct Step {
        int r0[2], bs[2], blk[2], nblk, mode, final_;
    };
    std::vector<Step> steps_;
};

// launchers of coarse_kernels.hip
void bt_launch_diag(int r0, int bs, int ld, const int *rp, const int *ci, const double *v, double *S, hipStream_t st);
void bt_launch_schur(int r0, int bs, int o0, int obs, int ld, const BtDevCsr &out, const BtDevCsr &inT, const double *SinvO, double *S,
                     hipStream_t st);
void bt_launch_invert(int bs, int ld, double *S, double *S2, double *col0, double *col1, int *pivots, int *colmap, int *singular,
                      double *out, hipStream_t st);
void bt_launch_solve_step(const int r0[2], const int bs[2], const int blk[2], int nblk, int mode, int final_, int ld, size_t blk_stride,
                          const double *sinv, const int *perm, const BtDevCsr &A, const BtDevEll &E, int n, const double *b, double *z, double *x,
                          hipStream_t st);
// G or H of one block: out[r][c] = sum_k Sinv[r][k] X[r0+k][win0+c], XT = CSR of the transposed coupling piece
void bt_launch_winprod(int r0, int bs, int ld, const double *Sinv, const BtDevCsr &XT, int win0, int wn, int ldw, double *out, hipStream_t st);
// interface-form solve: pre-pass y = blockdiag(S^-1) b[perm] over all blocks
void bt_launch_prepass(int nb, int B, int n, size_t blk_stride, const double *sinv, const int *perm, const double *b, double *y, hipStream_t st);
// one chain step on the window rows of up to two blocks; mode 0 inward (zw = y - G zw[gwin]), 2 middle block
// (zw = xw = y - G zw[gwin] - H zw[hwin]), 1 outward (xw = zw - H xw[hwin])
void bt_launch_win_step(const int blk[2], int nblk, int mode, int B, int n, int W, const int *wdesc, const double *g, const double *h,
                        const double *y, double *zw, double *xw, hipStream_t st);
// chain unrolled at setup: Out = -C X (W x K), identity on a diagonal block, and the triangular matrix-vector product of a pass
void bt_launch_chain_mul(const double *C, int nt, int W, const double *X, int ldx, int K, double *Out, int ldo, hipStream_t st);
void bt_launch_identity(double *M, int ld, int nt, hipStream_t st);
void bt_launch_tri_gemv(const double *const M[2], const int *const idx[2], const int ld[2], const int rows[2], int W, int lower,
                        const double *src, double *dst, hipStream_t st);
// z at the outward-facing windows of all chain blocks + the middle block (between the two chain passes)
void bt_launch_win_mid(int nb, int B, int n, int W, int mid, const int *wdesc, const double *g, const double *h, const double *y, double *zw,
                       double *xw, hipStream_t st);
// all rows: x[perm[r]] = y[r] - G zw[gwin] - H (mid ? zw : xw)[hwin]
void bt_launch_win_final(int nb, int B, int n, int W, int mid, const int *wdesc, const double *g, const double *h, const double *y,
                         const double *zw, const double *xw, const int *perm, double *x, hipStream_t st);
void launch_cvt_f2d(int n, const float *in, double *out, hipStream_t st);

}  // namespace sparsh
