// coarse.hpp -- coarsest-level direct solver on the device (stands where the reference calls
// Direct_Solver_Pardiso: analyse + factor once, src/AMG_coarse_level_solver.cpp:9-62; one solve per
// V-cycle, :64-76).  Two forms, chosen by the order of the system:
//   n <= dense_limit : explicit dense inverse (host: RCM + banded LU), one GEMV per solve;
//   larger           : block-tridiagonal ("twisted") factorisation of the RCM-ordered operator, kept as
//                      the explicit inverses of the Schur-complement diagonal blocks in HBM, factored and
//                      applied by the kernels of coarse_kernels.hip.  This is what lets the reference's
//                      level1 = 6 policy (coarsest level = N/32 rows, src/AMG_phases.cpp:51,77) run on
//                      the device.
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "host_setup.hpp"

namespace sparsh {

struct BtDevCsr {
    int *rp = nullptr;
    int *ci = nullptr;
    double *v = nullptr;
};

// ELL form of a coupling piece: entry k of row g at [k*n + g]; k = 0 when the piece is too ragged for it
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
bool bt_make_plan(const HostCsr &A, int target_blocks, int max_block, BtPlan &plan, std::string &err);

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
    // x = A^-1 b, device vectors in the operator's own numbering; enqueues on st
    void solve(const double *b, double *x, hipStream_t st) const;
    void release();

    bool ready() const { return n_ > 0; }
    bool dense() const { return dense_; }
    int n() const { return n_; }
    int block() const { return plan_.B; }
    int nblocks() const { return plan_.nb; }
    int bandwidth() const { return plan_.bw; }
    const double *dense_inverse() const { return inv_; }
    size_t bytes() const { return dense_ ? (size_t)n_ * n_ * 8 : plan_.sinv_bytes(); }
    double factor_seconds = 0.0;

private:
    int n_ = 0;
    bool dense_ = true;
    double *inv_ = nullptr;  // dense form
    BtPlan plan_;            // (pieces dropped after upload; sizes kept)
    double *sinv_ = nullptr, *z_ = nullptr;
    int *perm_ = nullptr;
    BtDevCsr out_, in_;
    BtDevEll out_ell_, in_ell_;
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
void launch_cvt_f2d(int n, const float *in, double *out, hipStream_t st);

}  // namespace sparsh
