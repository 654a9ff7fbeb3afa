// nd_kernels.hip -- device kernels of the nested-dissection multifrontal coarse solver (nd_plan.hpp, nd_solver.hpp).
// Replaces Direct_Solver_Pardiso (src/AMG_coarse_level_solver.cpp:9-76) on coarsest levels above dense_limit rows.
//
// Factorisation, one tree level at a time from the leaves up (all nodes of a level in the same launches):
//   scatter      entries of the permuted operator -> fronts (once)
//   extend-add   F_parent[rel, rel] += F22 of each child, one pass per child slot (so two children never race)
//   invert       D^-1 of the pivot block, Gauss-Jordan with partial pivoting inside the block: one workgroup per node up to
//                kNdTinyPivot rows; larger blocks of a level together, spread over the chip, one launch per pivot step
//   gemm         -D^-1 F12 -> B_k ; F21 D^-1 -> Lh_k ; F22 += F21 (-D^-1 F12)      (batched 64 x 64 tiles, fp64)
// Solve (2 * levels - 1 launches): per level upwards c[r] = b[perm r] - (row of Lh laid out per target row) . c ; per level
// downwards x[P_k] = B_k [c[P_k]; x[U_k]], scattered back to the caller's numbering; both through the same gathered-dot
// kernel.  No atomics: every sum has a fixed order.
#include <hip/hip_runtime.h>

#include "nd_solver.hpp"

namespace sparsh {

namespace {

constexpr int kNB = 256;

__device__ __forceinline__ double nd_wsum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(kNB) void nd_scatter_kernel(long long cnt, const long long *__restrict__ dst, const double *__restrict__ val,
                                                         double *__restrict__ fronts)
{
    for (long long i = (long long)blockIdx.x * kNB + threadIdx.x; i < cnt; i += (long long)gridDim.x * kNB) fronts[dst[i]] = val[i];
}

// blockIdx.x = child in the pass, blockIdx.y = chunk of 4 rows of its update block (one wave per row)
__global__ __launch_bounds__(kNB) void nd_extend_add_kernel(const NdDevNode *__restrict__ nodes, const int *__restrict__ children,
                                                            const int *__restrict__ rel_idx, double *__restrict__ fronts)
{
    const NdDevNode c = nodes[children[blockIdx.x]];
    const int i = blockIdx.y * (kNB / 64) + (threadIdx.x >> 6);
    if (i >= c.nu) return;
    const NdDevNode p = nodes[c.parent];
    const int lane = threadIdx.x & 63;
    const int ldc = c.np + c.nu, ldp = p.np + p.nu;
    const int *__restrict__ rel = rel_idx + c.rel;
    const double *__restrict__ src = fronts + c.foff + (size_t)(c.np + i) * ldc + c.np;
    double *__restrict__ dst = fronts + p.foff + (size_t)rel[i] * ldp;
    for (int j = lane; j < c.nu; j += 64) dst[rel[j]] += src[j];
}

// Gauss-Jordan inversion with partial pivoting of a SMALL pivot block (np <= kNdTinyPivot rows, at the top left of the front, leading
// dimension np + nu), one workgroup per node, the whole block in LDS; the inverse goes to the first np columns of B_k.
//   step k: p = argmax_{i >= k} |M[i][k]| (lowest index on ties); rows k, p swapped; row k scaled by 1 / pivot with
//   M[k][k] = 1 / pivot; every other row i: M[i][j] -= f_i * M[k][j], M[i][k] = -f_i / pivot  (f_i = old M[i][k]).
//   (PA)^-1 = A^-1 P^-1: the columns are swapped back in reverse pivot order at the end.
constexpr int kInvThreads = 256;

__global__ __launch_bounds__(kInvThreads) void nd_invert_kernel(const NdDevNode *__restrict__ nodes, const int *__restrict__ list,
                                                               const double *__restrict__ fronts, double *__restrict__ Bm, int *__restrict__ singular)
{
    __shared__ double M[kNdTinyPivot * (kNdTinyPivot + 1)];
    __shared__ double prow[kNdTinyPivot], fcol[kNdTinyPivot];
    __shared__ int cm[kNdTinyPivot], pivs[kNdTinyPivot];
    __shared__ double smax[kInvThreads / 64];
    __shared__ int sidx[kInvThreads / 64];
    __shared__ int s_p;
    const NdDevNode nd = nodes[list[blockIdx.x]];
    const int p = nd.np, ld = nd.np + nd.nu;
    if (p > kNdTinyPivot) return;
    const int lm = p + 1;  // odd-ish row pitch: column walks do not hit one bank
    const double *__restrict__ F = fronts + nd.foff;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int e = tid; e < p * p; e += kInvThreads) M[(e / p) * lm + e % p] = F[(size_t)(e / p) * ld + e % p];
    __syncthreads();
    for (int k = 0; k < p; ++k) {
        double best = -1.0;
        int bi = k;
        for (int i = k + tid; i < p; i += kInvThreads) {
            const double m = fabs(M[i * lm + k]);
            if (m > best) {
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
        if (lane == 0) {
            smax[wv] = best;
            sidx[wv] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            double b = smax[0];
            int ix = sidx[0];
            for (int q = 1; q < kInvThreads / 64; ++q)
                if (smax[q] > b || (smax[q] == b && sidx[q] < ix)) {
                    b = smax[q];
                    ix = sidx[q];
                }
            s_p = ix;
            pivs[k] = ix;
            if (!(b > 0.0)) *singular = 1;
        }
        __syncthreads();
        const int pv = s_p;
        if (pv != k)
            for (int j = tid; j < p; j += kInvThreads) {
                const double a = M[k * lm + j], b = M[pv * lm + j];
                M[k * lm + j] = b;
                M[pv * lm + j] = a;
            }
        __syncthreads();
        const double rpiv = 1.0 / M[k * lm + k];
        for (int j = tid; j < p; j += kInvThreads) {
            prow[j] = (j == k) ? rpiv : M[k * lm + j] * rpiv;
            fcol[j] = M[j * lm + k];
        }
        __syncthreads();
        for (int e = tid; e < p * p; e += kInvThreads) {
            const int i = e / p, j = e % p;
            double *m = &M[i * lm + j];
            *m = (i == k) ? prow[j] : ((j == k) ? -fcol[i] * rpiv : *m - fcol[i] * prow[j]);
        }
        __syncthreads();
    }
    for (int j = tid; j < p; j += kInvThreads) cm[j] = j;
    __syncthreads();
    if (tid == 0)
        for (int k = p - 1; k >= 0; --k) {
            const int q = pivs[k];
            const int t = cm[k];
            cm[k] = cm[q];
            cm[q] = t;
        }
    __syncthreads();
    double *__restrict__ out = Bm + nd.boff;
    for (int e = tid; e < p * p; e += kInvThreads) out[(size_t)(e / p) * ld + e % p] = M[(e / p) * lm + cm[e % p]];
}

// ---- pivot blocks above kNdTinyPivot rows: Gauss-Jordan with partial pivoting spread over the whole chip, ALL such nodes of a
// tree level in the same launches -- one launch per pivot step (the same scheme as bt_gj_kernel of coarse_kernels.hip, batched):
// out of place between two buffers per node (so no workgroup reads what another one writes in the same launch), every workgroup
// finds the step's pivot itself from the column magnitudes the previous step left, the scaled pivot row sits in LDS.
//   p = argmax_{i >= k} |src[i][k]| (lowest index on ties)
//   dst[k] = src[p] / piv, dst[k][k] = 1 / piv ; dst[i] = src[s] - f dst[k], dst[i][k] = -f / piv  (s = i == p ? k : i, f = src[s][k])
constexpr int kGjRowsPerWg = 8;

__global__ __launch_bounds__(kNB) void nd_gj_col0_kernel(const NdGjNode *__restrict__ nodes, const int *__restrict__ wg_node)
{
    const NdGjNode nd = nodes[wg_node[blockIdx.x]];
    const int i0 = ((int)blockIdx.x - nd.wg0) * kGjRowsPerWg;
    for (int q = threadIdx.x; q < kGjRowsPerWg; q += kNB) {
        const int i = i0 + q;
        if (i < nd.p) nd.c0[i] = fabs(nd.a[(size_t)i * nd.lda]);
    }
}

__global__ __launch_bounds__(kNB) void nd_gj_step_kernel(const NdGjNode *__restrict__ nodes, const int *__restrict__ wg_node, int k,
                                                         int *__restrict__ singular)
{
    extern __shared__ double prow[];
    __shared__ double smax[kNB / 64];
    __shared__ int sidx[kNB / 64];
    __shared__ int s_p;
    const NdGjNode nd = nodes[wg_node[blockIdx.x]];
    const int bs = nd.p;
    if (k >= bs) return;  // (workgroup-uniform) this node is done
    const bool odd = k & 1;
    const double *__restrict__ src = odd ? nd.b : nd.a;
    double *__restrict__ dst = odd ? nd.a : nd.b;
    const int lds = odd ? nd.ldb : nd.lda, ldd = odd ? nd.lda : nd.ldb;
    const double *__restrict__ colcur = odd ? nd.c1 : nd.c0;
    double *__restrict__ colnext = odd ? nd.c0 : nd.c1;
    double best = -1.0;
    int bi = k;
    for (int i = k + threadIdx.x; i < bs; i += kNB) {
        const double m = colcur[i];
        if (m > best) {
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
        for (int q = 1; q < kNB / 64; ++q)
            if (smax[q] > b || (smax[q] == b && sidx[q] < ix)) {
                b = smax[q];
                ix = sidx[q];
            }
        s_p = ix;
        if ((int)blockIdx.x == nd.wg0) {
            nd.piv[k] = ix;
            if (!(b > 0.0)) *singular = 1;
        }
    }
    __syncthreads();
    const int p = s_p;
    const double rpiv = 1.0 / src[(size_t)p * lds + k];
    const double *__restrict__ sp = src + (size_t)p * lds;
    for (int j = threadIdx.x; j < bs; j += kNB) prow[j] = (j == k) ? rpiv : sp[j] * rpiv;
    __syncthreads();
    const int i0 = ((int)blockIdx.x - nd.wg0) * kGjRowsPerWg;
#pragma unroll 1
    for (int q = 0; q < kGjRowsPerWg; ++q) {
        const int i = i0 + q;
        if (i >= bs) break;
        double *__restrict__ d = dst + (size_t)i * ldd;
        if (i == k) {
            for (int j = threadIdx.x; j < bs; j += kNB) d[j] = prow[j];
            continue;
        }
        const int sr_i = (i == p) ? k : i;
        const double *__restrict__ sr = src + (size_t)sr_i * lds;
        const double f = sr[k];
        for (int j = threadIdx.x; j < bs; j += kNB) {
            const double nv = (j == k) ? -f * rpiv : sr[j] - f * prow[j];
            d[j] = nv;
            if (j == k + 1) colnext[i] = fabs(nv);
        }
    }
}

// (PA)^-1 = A^-1 P^-1: the columns are swapped back in reverse pivot order.  One workgroup per node builds the column map ...
__global__ __launch_bounds__(kNB) void nd_gj_colmap_kernel(const NdGjNode *__restrict__ nodes)
{
    const NdGjNode nd = nodes[blockIdx.x];
    for (int j = threadIdx.x; j < nd.p; j += kNB) nd.cmap[j] = j;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = nd.p - 1; k >= 0; --k) {
            const int q = nd.piv[k];
            const int t = nd.cmap[k];
            nd.cmap[k] = nd.cmap[q];
            nd.cmap[q] = t;
        }
}

// ... and all rows copy through it into B_k (the buffer the last step wrote: a for an even number of steps, b for an odd one)
__global__ __launch_bounds__(kNB) void nd_gj_unscramble_kernel(const NdGjNode *__restrict__ nodes, const int *__restrict__ wg_node)
{
    const NdGjNode nd = nodes[wg_node[blockIdx.x]];
    const bool in_b = nd.p & 1;
    const double *__restrict__ R = in_b ? nd.b : nd.a;
    const int ldr = in_b ? nd.ldb : nd.lda;
    const int i0 = ((int)blockIdx.x - nd.wg0) * kGjRowsPerWg;
    for (int q = 0; q < kGjRowsPerWg; ++q) {
        const int i = i0 + q;
        if (i >= nd.p) break;
        for (int j = threadIdx.x; j < nd.p; j += kNB) nd.out[(size_t)i * nd.ldo + j] = R[(size_t)i * ldr + nd.cmap[j]];
    }
}

// batched GEMM, one 64 x 64 tile of one problem per workgroup, K in steps of 16 through LDS, 4 x 4 results per thread
constexpr int kTM = 64, kTN = 64, kTK = 16;

__global__ __launch_bounds__(kNB) void nd_gemm_kernel(const NdGemm *__restrict__ problems, const int *__restrict__ tiles)
{
    __shared__ double As[kTK][kTM + 1];
    __shared__ double Bs[kTK][kTN + 1];
    const NdGemm g = problems[tiles[2 * blockIdx.x]];
    const int tile = tiles[2 * blockIdx.x + 1];
    const int m0 = (tile / g.tiles_n) * kTM, n0 = (tile % g.tiles_n) * kTN;
    const int t = threadIdx.x;
    const int ty = t >> 4, tx = t & 15;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    const int ar = t >> 2, ak = (t & 3) * 4;    // A tile: row ar, k offsets ak .. ak + 3
    const int bk = t >> 4, bc = (t & 15) * 4;   // B tile: k row bk, columns bc .. bc + 3
    for (int k0 = 0; k0 < g.K; k0 += kTK) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gm = m0 + ar, gk = k0 + ak + q;
            As[ak + q][ar] = (gm < g.M && gk < g.K) ? g.a[(size_t)gm * g.lda + gk] : 0.0;
            const int gk2 = k0 + bk, gn = n0 + bc + q;
            Bs[bk][bc + q] = (gk2 < g.K && gn < g.N) ? g.b[(size_t)gk2 * g.ldb + gn] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kTK; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= g.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tx * 4 + j;
            if (gn >= g.N) continue;
            double *cp = g.c + (size_t)gm * g.ldc + gn;
            const double v = g.alpha * acc[i][j];
            *cp = g.beta ? *cp + v : v;
        }
    }
}

// Lh rows (row-major per source node, as the products leave them) -> the per-target-row layout of the forward pass:
// one wave per segment copies p numbers
__global__ __launch_bounds__(kNB) void nd_repack_kernel(long long nseg, const NdSegment *__restrict__ segs, const double *__restrict__ Lh,
                                                        double *__restrict__ Lf)
{
    const int lane = threadIdx.x & 63;
    for (long long s = (long long)blockIdx.x * (kNB / 64) + (threadIdx.x >> 6); s < nseg; s += (long long)gridDim.x * (kNB / 64)) {
        const NdSegment g = segs[s];
        for (int t = lane; t < g.p; t += 64) Lf[g.dst + t] = Lh[g.moff + t];
    }
}

// One pass of a solve over the rows of one tree level:  dot = sum_t M[moff + t] * vec[idx[ioff + t]],  vec = [c | x | b]
// (w holds c and x in the new numbering, b is the caller's vector in its own numbering).
//   FWD:  c[out] = b[bsrc] - dot            BWD:  x[out] = dot, also scattered to the caller's x[bsrc]
// The first nwide rows are long: one workgroup each; the others one wave each, four to a workgroup.  All loads of a
// row's first 8 (x 64 or x 256) elements are issued before the first use: record -> {M, idx} -> vec is the whole
// dependent chain of a launch.
template <bool FWD>
__global__ __launch_bounds__(kNB) void nd_gdot_kernel(const NdRow *__restrict__ rows, int nrows, int nwide, int n, const double *__restrict__ M,
                                                      const int *__restrict__ idx, double *__restrict__ w, const double *__restrict__ b,
                                                      double *__restrict__ x, NdProlong pr)
{
    __shared__ double part[kNB / 64];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool wide = (int)blockIdx.x < nwide;
    const int q = wide ? (int)blockIdx.x : nwide + ((int)blockIdx.x - nwide) * (kNB / 64) + wv;
    if (q >= nrows) return;  // narrow tail only: a wide workgroup keeps all its waves
    const NdRow R = rows[q];
    const double *__restrict__ row = M + R.moff;
    const int *__restrict__ ix = idx + R.ioff;
    const int tid = wide ? (int)threadIdx.x : lane;
    const int str = wide ? kNB : 64;
    const int n2 = 2 * n;
    double b0 = 0.0;
    if (FWD && lane == 0) b0 = b[R.bsrc];
    constexpr int U = 8;
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.0;
    for (int t0 = tid; t0 < R.len; t0 += U * str) {
        double m[U], v[U];
        int g[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u * str;
            const bool on = t < R.len;
            m[u] = on ? row[t] : 0.0;
            g[u] = on ? ix[t] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = g[u] < n2 ? w[g[u]] : b[g[u] - n2];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += (t0 + u * str < R.len) ? m[u] * v[u] : 0.0;
    }
    double a = nd_wsum(((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])));
    if (wide) {
        if (lane == 0) part[wv] = a;
        __syncthreads();
        a = (part[0] + part[1]) + (part[2] + part[3]);
        if (threadIdx.x != 0) return;
    } else if (lane != 0) {
        return;
    }
    if (FWD) {
        w[R.out] = b0 - a;
    } else {
        w[n + R.out] = a;
        x[R.bsrc] = a;
        if (pr.xf) {  // the finer level's rows this coarse row owns: x_f = 1.0 * x_c + x_f
            const int J = R.bsrc;
            int f0 = 2 * J, f1 = 2 * J + 1 < pr.nfine ? 2 * J + 1 : -1;
            if (pr.members) {
                f0 = pr.members[2 * (size_t)J];
                f1 = pr.members[2 * (size_t)J + 1];
            }
            pr.xf[f0] = 1.0 * a + pr.xf[f0];
            if (f1 >= 0) pr.xf[f1] = 1.0 * a + pr.xf[f1];
        }
    }
}

}  // namespace

void nd_launch_scatter(long long cnt, const long long *dst, const double *val, double *fronts, hipStream_t st)
{
    if (cnt <= 0) return;
    long long g = (cnt + kNB - 1) / kNB;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(nd_scatter_kernel, dim3((unsigned)g), dim3(kNB), 0, st, cnt, dst, val, fronts);
}

void nd_launch_extend_add(const NdDevNode *nodes, const int *children, int nchildren, int max_nu, const int *rel_idx, double *fronts, hipStream_t st)
{
    if (nchildren <= 0 || max_nu <= 0) return;
    hipLaunchKernelGGL(nd_extend_add_kernel, dim3(nchildren, (max_nu + kNB / 64 - 1) / (kNB / 64)), dim3(kNB), 0, st, nodes, children, rel_idx, fronts);
}

void nd_launch_invert(const NdDevNode *nodes, const int *list, int count, const double *fronts, double *Bm, int *singular, hipStream_t st)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(nd_invert_kernel, dim3(count), dim3(kInvThreads), 0, st, nodes, list, fronts, Bm, singular);
}

void nd_launch_gj_batched(const NdGjNode *nodes, int nnodes, const int *wg_node, int nwg, int max_p, int *singular, hipStream_t st)
{
    if (nnodes <= 0 || nwg <= 0) return;
    hipLaunchKernelGGL(nd_gj_col0_kernel, dim3(nwg), dim3(kNB), 0, st, nodes, wg_node);
    for (int k = 0; k < max_p; ++k) hipLaunchKernelGGL(nd_gj_step_kernel, dim3(nwg), dim3(kNB), (size_t)max_p * sizeof(double), st, nodes, wg_node, k, singular);
    hipLaunchKernelGGL(nd_gj_colmap_kernel, dim3(nnodes), dim3(kNB), 0, st, nodes);
    hipLaunchKernelGGL(nd_gj_unscramble_kernel, dim3(nwg), dim3(kNB), 0, st, nodes, wg_node);
}

void nd_launch_gemm(const NdGemm *problems, const int *tiles, int ntiles, hipStream_t st)
{
    if (ntiles <= 0) return;
    hipLaunchKernelGGL(nd_gemm_kernel, dim3(ntiles), dim3(kNB), 0, st, problems, tiles);
}

void nd_launch_repack(long long nseg, const NdSegment *segs, const double *Lh, double *Lf, hipStream_t st)
{
    if (nseg <= 0) return;
    long long g = (nseg + kNB / 64 - 1) / (kNB / 64);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(nd_repack_kernel, dim3((unsigned)g), dim3(kNB), 0, st, nseg, segs, Lh, Lf);
}

void nd_launch_pass(bool forward, const NdRow *rows, int nrows, int nwide, int n, const double *M, const int *idx, double *w, const double *b, double *x,
                    hipStream_t st, NdProlong pr)
{
    if (nrows <= 0) return;
    const int grid = nwide + (nrows - nwide + kNB / 64 - 1) / (kNB / 64);
    if (forward)
        hipLaunchKernelGGL(nd_gdot_kernel<true>, dim3(grid), dim3(kNB), 0, st, rows, nrows, nwide, n, M, idx, w, b, x, NdProlong());
    else
        hipLaunchKernelGGL(nd_gdot_kernel<false>, dim3(grid), dim3(kNB), 0, st, rows, nrows, nwide, n, M, idx, w, b, x, pr);
}

}  // namespace sparsh
