// coarse_kernels.hip -- device kernels of the block-tridiagonal direct solver for large coarsest
// levels (replaces Direct_Solver_Pardiso, src/AMG_coarse_level_solver.cpp:9-76, where the explicit
// dense inverse would not fit: the reference's level1 = 6 policy hands N/32 rows to PARDISO).
//
// The coarsest operator, renumbered by reverse Cuthill-McKee, is block tridiagonal for any block
// size B >= its bandwidth.  It is factored from both ends towards a middle block ("twisted"
// block LU, pivoting inside the diagonal blocks):
//     top chain     S_i = D_i - A[i,i-1] S_{i-1}^-1 A[i-1,i]        i = 0 .. mid-1
//     bottom chain  S_i = D_i - A[i,i+1] S_{i+1}^-1 A[i+1,i]        i = nb-1 .. mid+1
//     middle        S_m = D_m - A[m,m-1] S_{m-1}^-1 A[m-1,m] - A[m,m+1] S_{m+1}^-1 A[m+1,m]
// and only the explicit inverses S_i^-1 (B x B, dense) are kept.  A solve is
//     inward   z_i = S_i^-1 (b_i - A[i,outer] z_outer)              both chains at once, then the middle
//     outward  z_i = z_i - S_i^-1 (A[i,inner] z_inner)              both chains at once
// i.e. about nb sequential steps of one dense B x B GEMV (two per launch) plus a sparse product
// with the off-diagonal blocks of the permuted operator.  Setup kernels: Schur-complement
// assembly, Gauss-Jordan inversion with partial pivoting (one launch per pivot, ping-pong
// buffers, the next pivot column's magnitudes produced by the previous step), column unscramble.
// Everything is deterministic (no atomics).
#include <hip/hip_runtime.h>

#include "coarse.hpp"

namespace sparsh {

namespace {

constexpr int kCB = 256;  // threads per workgroup

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// S (bs x bs, leading dimension ld) = dense copy of the diagonal block of the permuted operator:
// rows [r0, r0+bs), entries with columns in the same range
__global__ __launch_bounds__(kCB) void bt_diag_kernel(int r0, int bs, int ld, const int *__restrict__ rp, const int *__restrict__ ci,
                                                      const double *__restrict__ v, double *__restrict__ S)
{
    const int r = blockIdx.x;
    double *row = S + (size_t)r * ld;
    for (int j = threadIdx.x; j < bs; j += kCB) row[j] = 0.0;
    __syncthreads();
    const int g = r0 + r;
    for (int j = rp[g] + threadIdx.x; j < rp[g + 1]; j += kCB) row[ci[j] - r0] = v[j];
}

// S -= A[i,o] Sinv_o A[o,i] for one neighbour block o.  One workgroup per row r of block i:
//   T = sum_k A[r,k] Sinv_o[k,:]        (k over the entries of row r in block o: few)
//   S[r,c] -= sum_m T[m] A[o*B+m, c]    (column c of A[o,i] = row c of the transposed piece)
// out: CSR of the entries that couple a block to its outer neighbour(s); inT: CSR of the TRANSPOSE of
// the entries that couple a block to its inner neighbour (row = column index of the entry).
__global__ __launch_bounds__(kCB) void bt_schur_kernel(int r0, int bs, int o0, int obs, int ld, const int *__restrict__ out_rp,
                                                       const int *__restrict__ out_ci, const double *__restrict__ out_v,
                                                       const int *__restrict__ inT_rp, const int *__restrict__ inT_ci,
                                                       const double *__restrict__ inT_v, const double *__restrict__ SinvO,
                                                       double *__restrict__ S)
{
    extern __shared__ double T[];  // obs doubles
    const int r = blockIdx.x, g = r0 + r;
    const int j0 = out_rp[g], j1 = out_rp[g + 1];
    int cnt = 0;
    for (int j = j0; j < j1; ++j) cnt += (out_ci[j] >= o0 && out_ci[j] < o0 + obs);
    if (cnt == 0) return;  // row does not touch block o: nothing to subtract (uniform over the workgroup)
    for (int m = threadIdx.x; m < obs; m += kCB) {
        double t = 0.0;
        for (int j = j0; j < j1; ++j) {
            const int k = out_ci[j];
            if (k >= o0 && k < o0 + obs) t += out_v[j] * SinvO[(size_t)(k - o0) * ld + m];
        }
        T[m] = t;
    }
    __syncthreads();
    double *row = S + (size_t)r * ld;
    for (int c = threadIdx.x; c < bs; c += kCB) {
        const int gc = r0 + c;
        double acc = 0.0;
        for (int j = inT_rp[gc]; j < inT_rp[gc + 1]; ++j) {
            const int m = inT_ci[j];  // source row of the entry (permuted numbering)
            if (m >= o0 && m < o0 + obs) acc += T[m - o0] * inT_v[j];
        }
        row[c] -= acc;
    }
}

// |S[i][0]| for the first pivot search
__global__ __launch_bounds__(kCB) void bt_col0_kernel(int bs, int ld, const double *__restrict__ S, double *__restrict__ colmag)
{
    for (int i = blockIdx.x * kCB + threadIdx.x; i < bs; i += gridDim.x * kCB) colmag[i] = fabs(S[(size_t)i * ld]);
}

// One Gauss-Jordan step (pivot column k) of the in-place inversion with row interchanges, written
// out of place (src -> dst) so no workgroup reads what another one writes:
//   p = argmax_{i>=k} |src[i][k]| (lowest index on ties; magnitudes come in colcur)
//   dst[k]   = src[p] / piv, dst[k][k] = 1/piv
//   dst[i]   = src[s] - f * dst[k], dst[i][k] = -f/piv     with s = (i == p ? k : i), f = src[s][k]
//   colnext[i] = |dst[i][k+1]|                              (the next step's pivot search)
constexpr int kGjRows = 8;  // rows per workgroup

__global__ __launch_bounds__(kCB) void bt_gj_kernel(int k, int bs, int ld, const double *__restrict__ src, double *__restrict__ dst,
                                                    const double *__restrict__ colcur, double *__restrict__ colnext,
                                                    int *__restrict__ pivots, int *__restrict__ singular)
{
    extern __shared__ double prow[];  // bs doubles: the scaled pivot row
    __shared__ double smax[kCB / 64];
    __shared__ int sidx[kCB / 64];
    __shared__ int s_p;
    // pivot search (every workgroup finds the same p)
    double best = -1.0;
    int bi = k;
    for (int i = k + threadIdx.x; i < bs; i += kCB) {
        const double m = colcur[i];
        if (m > best) {  // ascending i per thread: strict > keeps the lowest index
            best = m;
            bi = i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        smax[w] = best;
        sidx[w] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = smax[0];
        int ix = sidx[0];
        for (int q = 1; q < kCB / 64; ++q)
            if (smax[q] > b || (smax[q] == b && sidx[q] < ix)) {
                b = smax[q];
                ix = sidx[q];
            }
        s_p = ix;
        if (blockIdx.x == 0) {
            pivots[k] = ix;
            if (!(b > 0.0)) *singular = 1;
        }
    }
    __syncthreads();
    const int p = s_p;
    const double piv = src[(size_t)p * ld + k];
    const double rpiv = 1.0 / piv;
    const double *__restrict__ sp = src + (size_t)p * ld;
    for (int j = threadIdx.x; j < bs; j += kCB) prow[j] = (j == k) ? rpiv : sp[j] * rpiv;
    __syncthreads();
    const int i0 = blockIdx.x * kGjRows;
#pragma unroll 1
    for (int q = 0; q < kGjRows; ++q) {
        const int i = i0 + q;
        if (i >= bs) break;
        double *__restrict__ d = dst + (size_t)i * ld;
        if (i == k) {
            for (int j = threadIdx.x; j < bs; j += kCB) d[j] = prow[j];
            continue;  // row k is never a pivot candidate again
        }
        const int s = (i == p) ? k : i;
        const double *__restrict__ sr = src + (size_t)s * ld;
        const double f = sr[k];
        for (int j = threadIdx.x; j < bs; j += kCB) {
            const double nv = (j == k) ? -f * rpiv : sr[j] - f * prow[j];
            d[j] = nv;
            if (j == k + 1) colnext[i] = fabs(nv);
        }
    }
}

// out[i][j] = R[i][colmap[j]] where colmap undoes the row interchanges: (PA)^-1 = A^-1 P^-1, so the
// columns are swapped back in reverse pivot order.  One workgroup builds colmap in LDS (sequential,
// bs steps), then all copy.
__global__ __launch_bounds__(kCB) void bt_colmap_kernel(int bs, const int *__restrict__ pivots, int *__restrict__ colmap)
{
    extern __shared__ int cm[];
    for (int j = threadIdx.x; j < bs; j += kCB) cm[j] = j;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = bs - 1; k >= 0; --k) {
            const int p = pivots[k];
            const int t = cm[k];
            cm[k] = cm[p];
            cm[p] = t;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < bs; j += kCB) colmap[j] = cm[j];
}

__global__ __launch_bounds__(kCB) void bt_unscramble_kernel(int bs, int ld, const double *__restrict__ R, const int *__restrict__ colmap,
                                                            double *__restrict__ out)
{
    const int i = blockIdx.x;
    const double *__restrict__ r = R + (size_t)i * ld;
    double *__restrict__ o = out + (size_t)i * ld;
    for (int j = threadIdx.x; j < ld; j += kCB) o[j] = (j < bs) ? r[colmap[j]] : 0.0;
}

// One step of the solve for up to two blocks (blockIdx.y).  mode 0 (inward):
//   w = b[perm] - Aout z ; z_blk = Sinv w.       mode 1 (outward): w = Ain z ; z_blk -= Sinv w.
// When `final_` the block's rows of the solution are also scattered to x (original numbering).
// Each workgroup owns kSolveRows rows of the block's GEMV (a wave takes two rows at a time) and first
// builds the whole w in LDS (sparse, few entries per row).
constexpr int kSolveRows = 8;

struct BtStep {
    int r0[2], bs[2], blk[2];
    int nblk;
};

__global__ __launch_bounds__(kCB) void bt_solve_kernel(BtStep s, int mode, int final_, int ld, size_t blk_stride,
                                                       const double *__restrict__ sinv, const int *__restrict__ perm,
                                                       const int *__restrict__ a_rp, const int *__restrict__ a_ci,
                                                       const double *__restrict__ a_v, int ell_k, int n,
                                                       const int *__restrict__ e_ci, const double *__restrict__ e_v,
                                                       const double *__restrict__ b, double *__restrict__ z, double *__restrict__ x)
{
    extern __shared__ double w[];
    const int which = blockIdx.y;
    const int r0 = s.r0[which], bs = s.bs[which];
    const int row0 = blockIdx.x * kSolveRows;
    if (row0 >= bs) return;  // the second block of a step may be the shorter last block
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double *__restrict__ M = sinv + (size_t)s.blk[which] * blk_stride;
    const int ra = row0 + 2 * wv, rb = ra + 1;
    const bool ha = ra < bs, hb = rb < bs;
    const double *__restrict__ ma = M + (size_t)(ha ? ra : 0) * ld;
    const double *__restrict__ mb = M + (size_t)(hb ? rb : (ha ? ra : 0)) * ld;
    // The rows of S^-1 do not depend on w: request the first 1024 columns of both rows now (16-byte
    // loads, all in flight), so their HBM latency passes while w is assembled.  ld is a multiple of 64
    // and rows are zero-padded, so pairs stay aligned and in bounds.
    constexpr int U = 8;  // 8 x 128 columns
    double2 va[U], vb[U];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = c0 + lane * 2 + u * 128;
            const bool ok = j < bs;
            va[u] = ok ? *reinterpret_cast<const double2 *>(ma + j) : double2{0.0, 0.0};
            vb[u] = ok ? *reinterpret_cast<const double2 *>(mb + j) : double2{0.0, 0.0};
        }
    };
    load_chunk(0);
    if (ell_k > 0) {
        // ELL form of the coupling piece (entry k of row g at [k*n + g], at most 8 per row; padding: value 0 on the row's own
        // index, so every gather is a valid access).  A thread owns up to 4 rows per pass (r, r + 256, ...) and issues the index /
        // value loads of all of them, then all z gathers, then the sums: two dependent loads deep for the whole block instead of
        // two per row -- 100^3: 357 -> 268 us per solve.  (Gathers must stay unconditional: predicated ones serialise and made the
        // same idea 25 % slower.)
        constexpr int RB = 4, KMAX = 8;
        for (int rb = threadIdx.x; rb < bs; rb += RB * kCB) {
            int ci[RB][KMAX], g[RB];
            double ev[RB][KMAX], zv[RB][KMAX], bg[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int r = rb + q * kCB;
                g[q] = r0 + (r < bs ? r : 0);
                bg[q] = mode == 0 ? b[perm[g[q]]] : 0.0;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) {
                    const int kk = k < ell_k ? k : 0;
                    ci[q][k] = e_ci[(size_t)kk * n + g[q]];
                    ev[q][k] = k < ell_k ? e_v[(size_t)kk * n + g[q]] : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < RB; ++q)
#pragma unroll
                for (int k = 0; k < KMAX; ++k) zv[q][k] = z[ci[q][k]];
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int r = rb + q * kCB;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < KMAX; ++k) acc += ev[q][k] != 0.0 ? ev[q][k] * zv[q][k] : 0.0;  // padding must not pick up a NaN left in z
                if (r < bs) w[r] = mode == 0 ? bg[q] - acc : acc;
            }
        }
    } else {
        for (int r = threadIdx.x; r < bs; r += kCB) {
            const int g = r0 + r;
            double acc = 0.0;
            for (int j = a_rp[g]; j < a_rp[g + 1]; ++j) acc += a_v[j] * z[a_ci[j]];
            w[r] = mode == 0 ? b[perm[g]] - acc : acc;
        }
    }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
    for (int c0 = 0; c0 < bs; c0 += U * 128) {
        if (c0 > 0) load_chunk(c0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = c0 + lane * 2 + u * 128;
            const double w0 = j < bs ? w[j] : 0.0, w1 = j + 1 < bs ? w[j + 1] : 0.0;
            sa += va[u].x * w0;
            sa += va[u].y * w1;
            sb += vb[u].x * w0;
            sb += vb[u].y * w1;
        }
    }
    sa = wsum(sa);
    sb = wsum(sb);
    if (lane == 0 && ha) {
        const int ga = r0 + ra;
        const double za = mode == 0 ? sa : z[ga] - sa;
        z[ga] = za;
        if (final_) x[perm[ga]] = za;
        if (hb) {
            const int gb = r0 + rb;
            const double zb = mode == 0 ? sb : z[gb] - sb;
            z[gb] = zb;
            if (final_) x[perm[gb]] = zb;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Interface ("window") form of the solve for narrow bands.  With block size B >= 2 W (W = bandwidth rounded up
// to 64) only the first / last W rows of a block have entries in a neighbour block, so with
//     G_i = S_i^-1 A[i,outer]   H_i = S_i^-1 A[i,inner]        (columns restricted to the neighbour's window)
// the recurrences of the solve read
//     y   = blockdiag(S^-1) b                                       one launch over all blocks, no dependency
//     z_i = y_i - G_i z_outer[window]                               inward, needed on the window rows only
//     x_i = z_i - H_i x_inner[window]                               outward, window rows only
//     x_i = y_i - G_i z_outer[window] - H_i x_inner[window]         one launch over all rows at the end
// The dependent chain then moves 2W x W numbers per block instead of B x B.
// G / H of one block: one workgroup per row r of the block, one thread per window column c:
//     out[r][c] = sum over the entries (k, v) of column win0 + c of the coupling piece with k in the block: Sinv[r][k - r0] * v
// (XT = CSR of the transposed piece, so a column's entries come in a fixed order: deterministic).
__global__ __launch_bounds__(kCB) void bt_winprod_kernel(int r0, int bs, int ld, const double *__restrict__ Sinv, const int *__restrict__ t_rp,
                                                         const int *__restrict__ t_ci, const double *__restrict__ t_v, int win0, int wn, int ldw,
                                                         double *__restrict__ out)
{
    const int r = blockIdx.x;
    const double *__restrict__ srow = Sinv + (size_t)r * ld;
    double *__restrict__ orow = out + (size_t)r * ldw;
    for (int c = threadIdx.x; c < ldw; c += kCB) {
        double acc = 0.0;
        if (c < wn) {
            const int gc = win0 + c;
            for (int j = t_rp[gc]; j < t_rp[gc + 1]; ++j) {
                const int k = t_ci[j];
                if (k >= r0 && k < r0 + bs) acc += srow[k - r0] * t_v[j];
            }
        }
        orow[c] = acc;
    }
}

// y = blockdiag(S^-1) b[perm]: blockIdx.y = block, each workgroup kSolveRows rows (a wave takes two), b of the block in LDS
__global__ __launch_bounds__(kCB) void bt_prepass_kernel(int B, int n, size_t blk_stride, const double *__restrict__ sinv,
                                                         const int *__restrict__ perm, const double *__restrict__ b, double *__restrict__ y)
{
    extern __shared__ double w[];
    const int blk = blockIdx.y;
    const int r0 = blk * B;
    const int bs = min(B, n - r0);
    const int row0 = blockIdx.x * kSolveRows;
    if (row0 >= bs) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double *__restrict__ M = sinv + (size_t)blk * blk_stride;
    const int ra = row0 + 2 * wv, rb = ra + 1;
    const bool ha = ra < bs, hb = rb < bs;
    const double *__restrict__ ma = M + (size_t)(ha ? ra : 0) * B;
    const double *__restrict__ mb = M + (size_t)(hb ? rb : (ha ? ra : 0)) * B;
    constexpr int U = 8;
    double2 va[U], vb[U];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = c0 + lane * 2 + u * 128;
            const bool ok = j < bs;
            va[u] = ok ? *reinterpret_cast<const double2 *>(ma + j) : double2{0.0, 0.0};
            vb[u] = ok ? *reinterpret_cast<const double2 *>(mb + j) : double2{0.0, 0.0};
        }
    };
    load_chunk(0);
    for (int r = threadIdx.x; r < bs; r += kCB) w[r] = b[perm[r0 + r]];
    __syncthreads();
    double sa = 0.0, sb = 0.0;
    for (int c0 = 0; c0 < bs; c0 += U * 128) {
        if (c0 > 0) load_chunk(c0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = c0 + lane * 2 + u * 128;
            const double w0 = j < bs ? w[j] : 0.0, w1 = j + 1 < bs ? w[j + 1] : 0.0;
            sa += va[u].x * w0;
            sa += va[u].y * w1;
            sb += vb[u].x * w0;
            sb += vb[u].y * w1;
        }
    }
    sa = wsum(sa);
    sb = wsum(sb);
    if (lane == 0 && ha) {
        y[r0 + ra] = sa;
        if (hb) y[r0 + rb] = sb;
    }
}

// dot of one row of G / H (W doubles, W a multiple of 64) with the window vector v[0 .. wn): whole wave, result in all lanes of lane 0's reduction
__device__ __forceinline__ double win_dot(const double *__restrict__ row, const double *__restrict__ v, int wn, int W, int lane)
{
    double acc = 0.0;
    for (int c = lane; c < W; c += 64) {
        const double m = row[c];
        const double t = c < wn ? v[c] : 0.0;
        acc += m != 0.0 ? m * t : 0.0;  // zero padding / structural zeros must not pick up a NaN a broken-down solve left behind
    }
    return wsum(acc);
}

struct BtWinStep {
    int blk[2];
    int nblk;
};

// window rows of a block: the first and the last nt = min(W, bs) rows (all rows when they overlap)
__device__ __forceinline__ int win_row(int q, int bs, int nt)
{
    if (2 * nt >= bs) return q;
    return q < nt ? q : bs - 2 * nt + q;
}

// one chain step (see bt_launch_win_step); one wave per window row
__global__ __launch_bounds__(kCB) void bt_win_step_kernel(BtWinStep s, int mode, int B, int n, int W, const int *__restrict__ wdesc,
                                                          const double *__restrict__ g, const double *__restrict__ h,
                                                          const double *__restrict__ y, double *__restrict__ zw, double *__restrict__ xw)
{
    const int blk = s.blk[blockIdx.y];
    const int r0 = blk * B;
    const int bs = min(B, n - r0);
    const int nt = min(W, bs);
    const int nq = 2 * nt >= bs ? bs : 2 * nt;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (kCB / 64) + (threadIdx.x >> 6);
    if (q >= nq) return;  // wave-uniform
    const int lr = win_row(q, bs, nt);
    const int gwin0 = wdesc[4 * blk], gwn = wdesc[4 * blk + 1], hwin0 = wdesc[4 * blk + 2], hwn = wdesc[4 * blk + 3];
    const size_t ro = ((size_t)blk * B + lr) * W;
    const int gr = r0 + lr;
    double val;
    if (mode == 1) {
        val = zw[gr];
        if (hwn > 0) val -= win_dot(h + ro, xw + hwin0, hwn, W, lane);
    } else {
        val = y[gr];
        if (gwn > 0) val -= win_dot(g + ro, zw + gwin0, gwn, W, lane);
        if (mode == 2 && hwn > 0) val -= win_dot(h + ro, zw + hwin0, hwn, W, lane);
    }
    if (lane == 0) {
        if (mode == 1) {
            xw[gr] = val;
        } else {
            zw[gr] = val;
            if (mode == 2) xw[gr] = val;
        }
    }
}

// ---- the window chain unrolled at setup -------------------------------------------------------------------
// The inward recurrence on the inward-facing windows, zw_k = yw_k - C_k zw_{k-1} (k = chain position, C_k = the window
// rows of G of that block, W x W), is an affine recurrence with constant matrices: unrolled once at setup it reads
// zw = L yw with L block lower triangular, L[k][k] = I, L[k][j] = -C_k L[k-1][j].  Likewise the outward recurrence on the
// outward-facing windows, xo_k = zo_k - D_k xo_{k+1}: xo = U [zo; x_mid] with U block upper triangular.  A chain pass is
// then ONE whole-chip launch (a triangular matrix-vector product of (m W)^2 / 2 numbers) instead of m dependent steps.
// Out[q][c] = - sum_t C[q][t] X[t][c]   (q < nt rows of C are valid, the rest of Out stays zero); grid (ceil(K / 256), W)
__global__ __launch_bounds__(kCB) void bt_chain_mul_kernel(const double *__restrict__ C, int nt, int W, const double *__restrict__ X, int ldx,
                                                           int K, double *__restrict__ Out, int ldo)
{
    const int q = blockIdx.y;
    const int c = blockIdx.x * kCB + threadIdx.x;
    if (c >= K) return;
    double acc = 0.0;
    if (q < nt) {
        const double *__restrict__ crow = C + (size_t)q * W;
        for (int t = 0; t < W; ++t) acc += crow[t] * X[(size_t)t * ldx + c];
    }
    Out[(size_t)q * ldo + c] = -acc;
}

__global__ __launch_bounds__(kCB) void bt_identity_kernel(double *__restrict__ M, int ld, int nt)
{
    const int i = blockIdx.x * kCB + threadIdx.x;
    if (i < nt) M[(size_t)i * ld + i] = 1.0;
}

struct BtTri {
    const double *M[2];   // per chain: (rows x ld) row-major
    const int *idx[2];    // per chain: position -> index into src / dst (-1: padding)
    int ld[2], rows[2];   // rows to compute (m_c * W)
};

// dst[idx[r]] = sum_c M[r][c] src[idx[c]] over the columns of the triangle: lower: c < (r / W + 1) W; upper: (r / W) W <= c < ld.
// blockIdx.y = chain; a workgroup takes 4 consecutive rows (one per wave; same row block, W is a multiple of 64) and first
// gathers the vector segment they share into LDS, so the row loads (4 in flight per lane) depend on nothing.
__global__ __launch_bounds__(kCB) void bt_tri_gemv_kernel(BtTri t, int W, int lower, const double *__restrict__ src, double *__restrict__ dst)
{
    extern __shared__ double v[];
    const int c = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * (kCB / 64);
    if (r0 >= t.rows[c]) return;  // whole workgroup
    const int *__restrict__ idx = t.idx[c];
    const int ld = t.ld[c];
    const int k = r0 / W;
    const int c0 = lower ? 0 : k * W, c1 = lower ? (k + 1) * W : ld;
    for (int j = c0 + threadIdx.x; j < c1; j += kCB) {
        const int ix = idx[j];
        v[j - c0] = ix >= 0 ? src[ix] : 0.0;
    }
    __syncthreads();
    const int r = r0 + (threadIdx.x >> 6);
    if (r >= t.rows[c]) return;
    const int out = idx[r];
    if (out < 0) return;  // padding row of a short block
    const double *__restrict__ row = t.M[c] + (size_t)r * ld;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int j = c0 + lane; j < c1; j += 256) {
        const int j1 = j + 64, j2 = j + 128, j3 = j + 192;
        const double m0 = row[j];
        const double m1 = j1 < c1 ? row[j1] : 0.0;
        const double m2 = j2 < c1 ? row[j2] : 0.0;
        const double m3 = j3 < c1 ? row[j3] : 0.0;
        // structural zeros must not pick up a NaN a broken-down solve left behind
        a0 += m0 != 0.0 ? m0 * v[j - c0] : 0.0;
        a1 += m1 != 0.0 ? m1 * v[j1 - c0] : 0.0;
        a2 += m2 != 0.0 ? m2 * v[j2 - c0] : 0.0;
        a3 += m3 != 0.0 ? m3 * v[j3 - c0] : 0.0;
    }
    const double acc = wsum((a0 + a1) + (a2 + a3));
    if (lane == 0) dst[out] = acc;
}

// between the two chain launches: z at the outward-facing window of every chain block (needs only what the inward chain left),
// and the middle block: z = x = y - G z[gwin] - H z[hwin] at both its windows.  blockIdx.y = block, one wave per row.
__global__ __launch_bounds__(kCB) void bt_win_mid_kernel(int B, int n, int W, int mid, const int *__restrict__ wdesc, const double *__restrict__ g,
                                                         const double *__restrict__ h, const double *__restrict__ y, double *__restrict__ zw,
                                                         double *__restrict__ xw)
{
    const int blk = blockIdx.y;
    const int r0 = blk * B;
    const int bs = min(B, n - r0);
    const int nt = min(W, bs);
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (kCB / 64) + (threadIdx.x >> 6);
    int lr;
    if (blk == mid) {
        const int nq = 2 * nt >= bs ? bs : 2 * nt;
        if (q >= nq) return;
        lr = win_row(q, bs, nt);
    } else {
        if (q >= nt) return;
        lr = blk < mid ? q : bs - nt + q;  // top chain: first rows face outward; bottom chain: last rows
    }
    const int gwin0 = wdesc[4 * blk], gwn = wdesc[4 * blk + 1], hwin0 = wdesc[4 * blk + 2], hwn = wdesc[4 * blk + 3];
    const size_t ro = ((size_t)r0 + lr) * W;
    const int gr = r0 + lr;
    double val = y[gr];
    if (gwn > 0) val -= win_dot(g + ro, zw + gwin0, gwn, W, lane);
    if (blk == mid && hwn > 0) val -= win_dot(h + ro, zw + hwin0, hwn, W, lane);
    if (lane == 0) {
        zw[gr] = val;
        if (blk == mid) xw[gr] = val;
    }
}

// all rows of all blocks: x[perm[r]] = y[r] - G zw[gwin] - H (block == mid ? zw : xw)[hwin].  A wave takes kFinalRows consecutive
// rows (same block: B is a multiple of 64) so that all their G / H loads are in flight before the first reduction.
constexpr int kFinalRows = 4;

__global__ __launch_bounds__(kCB) void bt_win_final_kernel(int B, int n, int W, int mid, const int *__restrict__ wdesc, const double *__restrict__ g,
                                                           const double *__restrict__ h, const double *__restrict__ y, const double *__restrict__ zw,
                                                           const double *__restrict__ xw, const int *__restrict__ perm, double *__restrict__ x)
{
    const int lane = threadIdx.x & 63;
    const int gr0 = (blockIdx.x * (kCB / 64) + (threadIdx.x >> 6)) * kFinalRows;
    if (gr0 >= n) return;
    const int blk = gr0 / B;
    const int gwin0 = wdesc[4 * blk], gwn = wdesc[4 * blk + 1], hwin0 = wdesc[4 * blk + 2], hwn = wdesc[4 * blk + 3];
    const double *__restrict__ hv = (blk == mid ? zw : xw) + hwin0;
    const double *__restrict__ gv = zw + gwin0;
    double acc[kFinalRows];
#pragma unroll
    for (int q = 0; q < kFinalRows; ++q) acc[q] = 0.0;
    for (int c = lane; c < W; c += 64) {
        const double tg = c < gwn ? gv[c] : 0.0;
        const double th = c < hwn ? hv[c] : 0.0;
        double mg[kFinalRows], mh[kFinalRows];
#pragma unroll
        for (int q = 0; q < kFinalRows; ++q) {
            const bool on = gr0 + q < n;
            const size_t ro = (size_t)(gr0 + q) * W + c;
            mg[q] = on ? g[ro] : 0.0;
            mh[q] = on ? h[ro] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kFinalRows; ++q) {
            acc[q] += mg[q] != 0.0 ? mg[q] * tg : 0.0;
            acc[q] += mh[q] != 0.0 ? mh[q] * th : 0.0;
        }
    }
#pragma unroll
    for (int q = 0; q < kFinalRows; ++q) {
        const double d = wsum(acc[q]);
        if (lane == 0 && gr0 + q < n) x[perm[gr0 + q]] = y[gr0 + q] - d;
    }
}

__global__ __launch_bounds__(kCB) void cvt_f2d_kernel(int n, const float *__restrict__ in, double *__restrict__ out)
{
    for (int i = blockIdx.x * kCB + threadIdx.x; i < n; i += gridDim.x * kCB) out[i] = (double)in[i];
}

}  // namespace

void bt_launch_diag(int r0, int bs, int ld, const int *rp, const int *ci, const double *v, double *S, hipStream_t st)
{
    hipLaunchKernelGGL(bt_diag_kernel, dim3(bs), dim3(kCB), 0, st, r0, bs, ld, rp, ci, v, S);
}

void bt_launch_schur(int r0, int bs, int o0, int obs, int ld, const BtDevCsr &out, const BtDevCsr &inT, const double *SinvO, double *S,
                     hipStream_t st)
{
    hipLaunchKernelGGL(bt_schur_kernel, dim3(bs), dim3(kCB), (size_t)obs * sizeof(double), st, r0, bs, o0, obs, ld, out.rp, out.ci, out.v,
                       inT.rp, inT.ci, inT.v, SinvO, S);
}

// inverts S (bs x bs, ld) into out; S and S2 are scratch.  Enqueues bs + 3 launches.
void bt_launch_invert(int bs, int ld, double *S, double *S2, double *col0, double *col1, int *pivots, int *colmap, int *singular,
                      double *out, hipStream_t st)
{
    hipLaunchKernelGGL(bt_col0_kernel, dim3((bs + kCB - 1) / kCB), dim3(kCB), 0, st, bs, ld, S, col0);
    const int grid = (bs + kGjRows - 1) / kGjRows;
    double *src = S, *dst = S2, *cc = col0, *cn = col1;
    for (int k = 0; k < bs; ++k) {
        hipLaunchKernelGGL(bt_gj_kernel, dim3(grid), dim3(kCB), (size_t)bs * sizeof(double), st, k, bs, ld, src, dst, cc, cn, pivots, singular);
        double *t = src;
        src = dst;
        dst = t;
        t = cc;
        cc = cn;
        cn = t;
    }
    hipLaunchKernelGGL(bt_colmap_kernel, dim3(1), dim3(kCB), (size_t)bs * sizeof(int), st, bs, pivots, colmap);
    hipLaunchKernelGGL(bt_unscramble_kernel, dim3(bs), dim3(kCB), 0, st, bs, ld, src, colmap, out);
}

void bt_launch_solve_step(const int r0[2], const int bs[2], const int blk[2], int nblk, int mode, int final_, int ld, size_t blk_stride,
                          const double *sinv, const int *perm, const BtDevCsr &A, const BtDevEll &E, int n, const double *b, double *z, double *x,
                          hipStream_t st)
{
    BtStep s;
    int mx = 0;
    for (int q = 0; q < 2; ++q) {
        s.r0[q] = r0[q < nblk ? q : 0];
        s.bs[q] = bs[q < nblk ? q : 0];
        s.blk[q] = blk[q < nblk ? q : 0];
        if (q < nblk) mx = bs[q] > mx ? bs[q] : mx;
    }
    s.nblk = nblk;
    const int gx = (mx + kSolveRows - 1) / kSolveRows;
    hipLaunchKernelGGL(bt_solve_kernel, dim3(gx, nblk), dim3(kCB), (size_t)mx * sizeof(double), st, s, mode, final_, ld, blk_stride, sinv, perm,
                       A.rp, A.ci, A.v, E.k, n, E.ci, E.v, b, z, x);
}

void bt_launch_winprod(int r0, int bs, int ld, const double *Sinv, const BtDevCsr &XT, int win0, int wn, int ldw, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(bt_winprod_kernel, dim3(bs), dim3(kCB), 0, st, r0, bs, ld, Sinv, XT.rp, XT.ci, XT.v, win0, wn, ldw, out);
}

void bt_launch_prepass(int nb, int B, int n, size_t blk_stride, const double *sinv, const int *perm, const double *b, double *y, hipStream_t st)
{
    hipLaunchKernelGGL(bt_prepass_kernel, dim3((B + kSolveRows - 1) / kSolveRows, nb), dim3(kCB), (size_t)B * sizeof(double), st, B, n, blk_stride, sinv,
                       perm, b, y);
}

void bt_launch_win_step(const int blk[2], int nblk, int mode, int B, int n, int W, const int *wdesc, const double *g, const double *h,
                        const double *y, double *zw, double *xw, hipStream_t st)
{
    BtWinStep s;
    s.blk[0] = blk[0];
    s.blk[1] = blk[nblk > 1 ? 1 : 0];
    s.nblk = nblk;
    const int rows = 2 * W < B ? 2 * W : B;  // most window rows a block can have
    hipLaunchKernelGGL(bt_win_step_kernel, dim3((rows + kCB / 64 - 1) / (kCB / 64), nblk), dim3(kCB), 0, st, s, mode, B, n, W, wdesc, g, h, y, zw, xw);
}

void bt_launch_chain_mul(const double *C, int nt, int W, const double *X, int ldx, int K, double *Out, int ldo, hipStream_t st)
{
    if (K <= 0) return;
    hipLaunchKernelGGL(bt_chain_mul_kernel, dim3((K + kCB - 1) / kCB, W), dim3(kCB), 0, st, C, nt, W, X, ldx, K, Out, ldo);
}

void bt_launch_identity(double *M, int ld, int nt, hipStream_t st)
{
    if (nt <= 0) return;
    hipLaunchKernelGGL(bt_identity_kernel, dim3((nt + kCB - 1) / kCB), dim3(kCB), 0, st, M, ld, nt);
}

void bt_launch_tri_gemv(const double *const M[2], const int *const idx[2], const int ld[2], const int rows[2], int W, int lower,
                        const double *src, double *dst, hipStream_t st)
{
    BtTri t;
    int mx = 0;
    for (int c = 0; c < 2; ++c) {
        t.M[c] = M[c];
        t.idx[c] = idx[c];
        t.ld[c] = ld[c];
        t.rows[c] = rows[c];
        mx = rows[c] > mx ? rows[c] : mx;
    }
    if (mx <= 0) return;
    const int ldmax = ld[0] > ld[1] ? ld[0] : ld[1];
    hipLaunchKernelGGL(bt_tri_gemv_kernel, dim3((mx + kCB / 64 - 1) / (kCB / 64), 2), dim3(kCB), (size_t)ldmax * sizeof(double), st, t, W, lower, src, dst);
}

void bt_launch_win_mid(int nb, int B, int n, int W, int mid, const int *wdesc, const double *g, const double *h, const double *y, double *zw,
                       double *xw, hipStream_t st)
{
    const int rows = 2 * W < B ? 2 * W : B;
    hipLaunchKernelGGL(bt_win_mid_kernel, dim3((rows + kCB / 64 - 1) / (kCB / 64), nb), dim3(kCB), 0, st, B, n, W, mid, wdesc, g, h, y, zw, xw);
}

void bt_launch_win_final(int nb, int B, int n, int W, int mid, const int *wdesc, const double *g, const double *h, const double *y,
                         const double *zw, const double *xw, const int *perm, double *x, hipStream_t st)
{
    (void)nb;
    const int per_wg = (kCB / 64) * kFinalRows;
    hipLaunchKernelGGL(bt_win_final_kernel, dim3((n + per_wg - 1) / per_wg), dim3(kCB), 0, st, B, n, W, mid, wdesc, g, h, y, zw, xw, perm, x);
}

void launch_cvt_f2d(int n, const float *in, double *out, hipStream_t st)
{
    if (n <= 0) return;
    int g = (n + kCB * 2 - 1) / (kCB * 2);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(cvt_f2d_kernel, dim3(g), dim3(kCB), 0, st, n, in, out);
}

}  // namespace sparsh
