// kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4): the AMG solve-phase
// operators.  Wave = 64 lanes, workgroup = 256 threads (one wave per SIMD).  All kernels are
// memory bound (SpMV: 2 flop per 12 B); there is no MFMA here on purpose.
//
// SpMV-type operators (y = A x with a fused epilogue OP_*: SpMV, residual, Jacobi sweep, p.Ap,
// ||Ax-b||^2, ...) exist in these families, all bitwise identical, chosen per operator at launch:
//   csr_block_kernel / csr_wave_kernel   CSR-stream, any matrix (irregular operators, P, R)
//   sell_kernel                          sliced ELL (64-row slices), near-uniform row lengths
//   sdia_kernel                          sliced diagonals: per slice the distinct offsets once, no
//                                        column indices; paths per slice: general (value blocks),
//                                        plain (no constant slot), const (> 8 constant slots),
//                                        record (<= 8 constant slots, one 192-B descriptor)
//   sdia_tab_kernel                      sliced diagonals with a level-wide stencil passed as kernel
//                                        arguments, 64-B lane-mask record per slice; near variant:
//                                        x[r-1], x[r], x[r+1] from one gather (DPP + scalar edge loads)
//
// Arithmetic contract (parity with the reference's CPU path, built with -ffp-contract=off):
//   * a row sum adds the products a_ij*x_j one by one in stored column order, each product
//     rounded before the add -- the order of a scalar CPU loop over the CSR row;
//   * Jacobi: h = b - s ; x_new = x + (omega*h)/d  (src/AMG_smoothers.cpp:62-72);
//   * reductions are two-stage (per-workgroup partial -> one finalize workgroup), fixed tree
//     order: bitwise reproducible run to run, no float atomics.
#include <hip/hip_runtime.h>

#include "kernels.hpp"

#include <algorithm>
#include <vector>

namespace sparsh {

namespace {

// ------------------------------------------------------------------ helpers

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0
}

// sum over the workgroup; result valid in thread 0.  red must hold kBlock/64 doubles.
__device__ __forceinline__ double block_sum(double v, double *red)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < kBlock / 64; ++k) s += red[k];
    }
    return s;
}

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD, MI355X
// microarch guide).  Give each XCD one contiguous range of row blocks so the x-vector lines
// its neighbouring row blocks share stay in that XCD's 4 MiB L2.  Placement affects speed only.
// remap = 1: XCD x owns the x-th contiguous eighth of the row blocks.
// remap = G > 1: XCD x owns the groups x, x+8, x+16, ... of G consecutive row blocks: all XCDs
// sweep the same neighbourhood of the matrix at a time (DRAM locality), while blocks that share
// x-vector lines (neighbours within a group) share an L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk, int mode)
{
    const int xcd = b & 7, i = b >> 3;
    if (mode == 1) {
        const int chunk = (nblk + 7) >> 3;
        return xcd * chunk + i;
    }
    return ((i / mode) * 8 + xcd) * mode + (i % mode);
}

inline int remap_grid(int nblk, int mode)
{
    if (mode <= 0) return nblk;
    const int q = 8 * (mode == 1 ? 1 : mode);
    return ((nblk + q - 1) / q) * q;
}

using i2v = int __attribute__((ext_vector_type(2)));
using d2v = double __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ double ld_stream(const double *p)
{
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}
template <bool NT>
__device__ __forceinline__ int ld_stream(const int *p)
{
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
}

// ------------------------------------------------------------------ SpMV-type kernels
//
// Three kernel families share one epilogue (what happens to the row sum) and one contract:
// the products of a row are added one by one in stored order.
//
//  * csr_block_kernel (kind 0): a workgroup owns a run of consecutive rows whose products fit
//    its LDS buffer.  Phase 1: 256 threads walk the run's nnz range with unit stride (coalesced
//    val/col reads), gather x[col] and park the rounded products in LDS.  Phase 2: one thread
//    per row adds its products.  Row operands (b, d, x_i, rowptr) are fetched before phase 1
//    so their latency hides under the stream.
//  * csr_wave_kernel (kind 1): the same per wave (64 rows, wave-private LDS slice), no
//    workgroup barrier between the phases.
//  * sell_kernel (kind 2): sliced-ELL mirror, lane = row, no LDS: every load of a wave is one
//    contiguous segment (and for stencil matrices so is the x gather).

constexpr bool op_needs_b(int OP) { return OP == OP_RESID || OP == OP_JACOBI || OP == OP_JACOBI_DOT || OP == OP_RESNORM || OP == OP_RESID_PAIR || OP == OP_JACOBI_PROLONG; }
constexpr bool op_needs_d(int OP) { return OP == OP_JACOBI || OP == OP_JACOBI_DOT || OP == OP_JACOBI_PROLONG; }
constexpr bool op_needs_xi(int OP) { return OP == OP_JACOBI || OP == OP_JACOBI_DOT || OP == OP_SPMV_DOT || OP == OP_JACOBI_PROLONG; }
constexpr bool op_reduces(int OP) { return OP == OP_SPMV_DOT || OP == OP_RESNORM || OP == OP_JACOBI_DOT; }

struct RowOperands {
    double bi = 0.0, di = 1.0, xi = 0.0;
};

// LOAD_D = false: the kernel picks the diagonal out of the matrix stream it reads anyway (the
// entry with col == row; same value as diag[], src/AMG_cpu_matrix.cpp:35-51) and skips the 8 B/row
template <int OP, bool LOAD_D = true, bool LOAD_XI = true>
__device__ __forceinline__ RowOperands load_row_operands(const CsrArgs &a, int row)
{
    RowOperands o;
    if constexpr (op_needs_b(OP)) o.bi = a.b[row];
    if constexpr (op_needs_d(OP) && LOAD_D) o.di = a.d[row];
    if constexpr (op_needs_d(OP) && !LOAD_D) o.di = 0.0;  // a row without a diagonal entry keeps diag[] = 0
    if constexpr (op_needs_xi(OP) && LOAD_XI) o.xi = a.x[row];
    return o;
}

// stores the row result, returns this row's contribution to the fused reduction (if any)
template <int OP>
__device__ __forceinline__ double row_epilogue(const CsrArgs &a, int row, double sum, const RowOperands &o)
{
    if constexpr (OP == OP_SPMV) {
        a.y[row] = sum;
    } else if constexpr (OP == OP_RESID) {
        a.y[row] = 1.0 * o.bi + (-1.0) * sum;
    } else if constexpr (OP == OP_JACOBI || OP == OP_JACOBI_DOT) {
        const double h = 1.0 * o.bi + (-1.0) * sum;
        const double xn = o.xi + a.omega * h / o.di;
        a.y[row] = xn;
        if constexpr (OP == OP_JACOBI_DOT) return xn * o.bi;
    } else if constexpr (OP == OP_JACOBI_PROLONG) {
        const double h = 1.0 * o.bi + (-1.0) * sum;
        const double xn = o.xi + a.omega * h / o.di;
        // transfer_solution for the rows this aggregate owns: x_f = 1.0 * x_c + x_f (nobody else writes them)
        if (a.members) {
            const i2v m = *reinterpret_cast<const i2v *>(a.members + 2 * (size_t)row);
            a.y2[m.x] = 1.0 * xn + a.y2[m.x];
            if (m.y >= 0) a.y2[m.y] = 1.0 * xn + a.y2[m.y];
        } else {
            const int f = 2 * row;
            if (f + 1 < a.nfine) {
                d2v *p = reinterpret_cast<d2v *>(a.y2 + f);
                d2v v = *p;
                v.x = 1.0 * xn + v.x;
                v.y = 1.0 * xn + v.y;
                *p = v;
            } else {
                a.y2[f] = 1.0 * xn + a.y2[f];
            }
        }
    } else if constexpr (OP == OP_ADD) {
        a.y[row] = sum + a.y[row];
    } else if constexpr (OP == OP_SPMV_DOT) {
        a.y[row] = sum;
        return o.xi * sum;
    } else if constexpr (OP == OP_RESNORM) {
        const double h = sum + (-1.0) * o.bi;
        return h * h;
    }
    return 0.0;
}

// Phase 1 of the stream kernels: products of entries [j0, j1) into prod[0 .. j1-j0), walked by
// `nthr` threads with unit stride.  Loads are issued in batches of U independent requests per
// thread (indices clamped into the run, so no load sits behind a branch) -- a wave keeps 2U-3U
// loads in flight instead of one dependent col -> x chain at a time.
// VEC: two entries per request through 16-B / 8-B loads at even indices (col/val carry kCsrPad
// zeroed tail entries, so the pair stays in bounds).
template <bool NT, int VEC>
__device__ __forceinline__ void stream_products(const int *__restrict__ col, const double *__restrict__ val,
                                                const double *__restrict__ x, int j0, int j1, int t, int nthr,
                                                double *__restrict__ prod)
{
    if (j1 <= j0) return;
    if constexpr (VEC > 0) {
        constexpr int U = 2;
        const int jb = j0 & ~1;
        const int jlast = (j1 - 1) & ~1;  // last valid even pair index
        for (int j = jb + 2 * t; j < j1; j += 2 * U * nthr) {
            i2v c[U];
            d2v v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int jj = j + 2 * u * nthr;
                jj = jj < jlast ? jj : jlast;
                if constexpr (NT) {
                    c[u] = __builtin_nontemporal_load(reinterpret_cast<const i2v *>(col + jj));
                    v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(val + jj));
                } else {
                    c[u] = *reinterpret_cast<const i2v *>(col + jj);
                    v[u] = *reinterpret_cast<const d2v *>(val + jj);
                }
            }
            double xa[U], xb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xa[u] = x[c[u].x];
                xb[u] = x[c[u].y];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j + 2 * u * nthr;
                const double p0 = v[u].x * xa[u];
                const double p1 = v[u].y * xb[u];
                if (jj >= j0 && jj < j1) prod[jj - j0] = p0;
                if (jj + 1 < j1) prod[jj + 1 - j0] = p1;
            }
        }
    } else {
        constexpr int U = 4;
        const int jlast = j1 - 1;
        for (int j = j0 + t; j < j1; j += U * nthr) {
            int c[U];
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int jj = j + u * nthr;
                jj = jj < jlast ? jj : jlast;
                c[u] = ld_stream<NT>(col + jj);
                v[u] = ld_stream<NT>(val + jj);
            }
            double xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = x[c[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j + u * nthr;
                const double p = v[u] * xv[u];
                if (jj < j1) prod[jj - j0] = p;
            }
        }
    }
}

// Phase 2: add the products prod[s .. e) of one row in order.  LDS reads are issued eight at a
// time (clamped index), the adds stay strictly sequential.
__device__ __forceinline__ double row_sum_lds(const double *__restrict__ prod, int s, int e)
{
    double sum = 0.0;
    for (int k = s; k < e; k += 8) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k + u < e ? k + u : e - 1;
            t[u] = prod[kk];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sum = (k + u < e) ? sum + t[u] : sum;
    }
    return sum;
}

// TAG: 1 for launches on the finest level.  Identical code; a distinct symbol lets profiler
// summaries (rocprofv3 --stats) separate the dominant finest-level launches from coarse ones.
template <int OP, bool NT, int VEC, int TAG>
__global__ __launch_bounds__(kBlock) void csr_block_kernel(const int *__restrict__ rowblk, int nblk, int remap,
                                                            const int *__restrict__ rowptr, const int *__restrict__ col,
                                                            const double *__restrict__ val, CsrArgs a)
{
    __shared__ double prod[kStreamNnz];
    __shared__ double red[kBlock / 64];
    int bid = remap ? xcd_remap(blockIdx.x, nblk, remap) : blockIdx.x;
    if (a.reverse && bid < nblk) bid = nblk - 1 - bid;
    if (bid >= nblk) return;  // whole workgroup leaves together
    const int tid = threadIdx.x;
    // one 16-byte record per block {first row, end row, first entry, end entry}: the stream loads start one
    // dependent round trip earlier than through rowblk -> rowptr
    const int4 br = reinterpret_cast<const int4 *>(rowblk)[bid];
    const int r0 = br.x, r1 = br.y;
    const int nrows = r1 - r0;
    const int j0 = br.z, j1 = br.w;
    const double *__restrict__ x = a.x;

    // row operands first: their latency hides under the product stream
    const bool has_row = tid < nrows;
    const int row = r0 + tid;
    int s = 0, e = 0;
    RowOperands o;
    if (has_row) {
        s = rowptr[row] - j0;
        e = rowptr[row + 1] - j0;
        o = load_row_operands<OP>(a, row);
    }

    double sum = 0.0;
    bool store = has_row;
    if (nrows == 1 && j1 - j0 > kStreamNnz) {  // one long row: strided partial sums, tree-combined
        double part = 0.0;
        for (int j = j0 + tid; j < j1; j += kBlock) part += ld_stream<NT>(val + j) * x[ld_stream<NT>(col + j)];
        part = block_sum(part, red);
        __syncthreads();
        sum = part;
        store = (tid == 0);
    } else {
        stream_products<NT, VEC>(col, val, x, j0, j1, tid, kBlock, prod);
        __syncthreads();
        sum = row_sum_lds(prod, s, e);
    }

    double acc = 0.0;
    if (store) acc = row_epilogue<OP>(a, row, sum, o);
    if constexpr (op_reduces(OP)) {
        __syncthreads();
        const double t = block_sum(acc, red);
        if (tid == 0) a.partial[bid] = t;
    }
}

// csr_rowlane_kernel (kind 0, vec = 2): the CSR-stream kernel with the x gathers issued in ROW-LANE order.
// Counters (profiles/r02_pmc/diag_csr_block_vs_sell_counters.txt) show csr_block_kernel address-unit bound: TA busy 93 % of
// the launch, 99 M L1 accesses per sweep against 54 M for the sliced-ELL kernel -- its gathers run in CSR order, where the 64
// lanes of one instruction cover ~9 rows x 7 columns = a dozen cache lines.  Here phase 1 only copies the block's col/val
// run into LDS (coalesced, no gathers); in phase 2 lane = row walks its entries in stored order reading (col, val) from LDS
// and gathering x[col]: the k-th entries of 64 consecutive rows of a banded / mesh-ordered operator are neighbours in x, so
// a gather touches 4-6 lines instead of 12.  Same rounded products, same order of additions.  The diagonal is picked out of
// the stream (col == row), so diag[] is not read.
// The two phases of the row-lane CSR-stream kernels for one row block with <= kStreamNnz entries.  COMP: the block's column
// indices come as 16-bit deltas (see csr_rowlane_body).  All 256 threads call it (workgroup barrier inside).
template <bool NT, bool COMP>
__device__ __forceinline__ void rowlane_stage_sum(int j0, int j1, int s, int e, int row, int cb, const int *__restrict__ col,
                                                  const unsigned short *__restrict__ col16, const double *__restrict__ val,
                                                  const double *__restrict__ x, double *sval, int *scol, double &sum, double &dv)
{
    const int tid = threadIdx.x;
    unsigned short *sc16 = reinterpret_cast<unsigned short *>(scol);
    // phase 1: the block's col/val run -> LDS, paired loads at even indices (16 B of values with 8 B of indices or 4 B of
    // deltas; col/val/col16 carry kCsrPad zeroed tail entries)
    if (j1 > j0) {
        const int jb = j0 & ~1;
        const int jlast = (j1 - 1) & ~1;
        constexpr int U = 4;  // 8 entries per thread in flight: one pass over a full block (measured 4-5 % faster than 2 at 216^3)
        for (int j = jb + 2 * tid; j < j1; j += 2 * U * kBlock) {
            i2v c[U];
            unsigned w[U];
            d2v v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int jj = j + 2 * u * kBlock;
                jj = jj < jlast ? jj : jlast;
                if constexpr (COMP) {
                    if constexpr (NT) w[u] = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(col16 + jj));
                    else w[u] = *reinterpret_cast<const unsigned *>(col16 + jj);
                } else {
                    if constexpr (NT) c[u] = __builtin_nontemporal_load(reinterpret_cast<const i2v *>(col + jj));
                    else c[u] = *reinterpret_cast<const i2v *>(col + jj);
                }
                if constexpr (NT) v[u] = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(val + jj));
                else v[u] = *reinterpret_cast<const d2v *>(val + jj);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {  // element-wise LDS writes (whole-pair b64/b128 writes measured 5-8 % slower)
                const int jj = j + 2 * u * kBlock;
                if (jj >= j0 && jj < j1) {
                    if constexpr (COMP) sc16[jj - jb] = (unsigned short)(w[u] & 0xffffu);
                    else scol[jj - jb] = c[u].x;
                    sval[jj - jb] = v[u].x;
                }
                if (jj + 1 >= j0 && jj + 1 < j1) {
                    if constexpr (COMP) sc16[jj + 1 - jb] = (unsigned short)(w[u] >> 16);
                    else scol[jj + 1 - jb] = c[u].y;
                    sval[jj + 1 - jb] = v[u].y;
                }
            }
        }
    }
    __syncthreads();
    // phase 2: lane = row; entries in stored order, eight (col, val) pairs and their gathers in flight at a time
    int ccur = cb;
    const int sh = j0 & 1;  // LDS position of entry j0
    for (int k = s; k < e; k += 8) {
        int c[8];
        double v[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = (k + u < e ? k + u : e - 1) + sh;
            if constexpr (COMP) c[u] = (int)sc16[kk];
            else c[u] = scol[kk];
            v[u] = sval[kk];
        }
        if constexpr (COMP) {  // deltas -> columns: running sum along the row, started at the block's base
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ccur = k + u < e ? ccur + c[u] : ccur;
                c[u] = ccur;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) xv[u] = (k + u < e) ? x[c[u]] : 0.0;  // lanes past their row's end issue no access (ragged rows)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double t = v[u] * xv[u];
            const bool on = k + u < e;
            sum = on ? sum + t : sum;
            dv = (on && c[u] == row) ? v[u] : dv;
        }
    }
}

// C16 (csr_rowlane16_kernel, vec = 4): the column indices of a block travel as 16-bit deltas (DevCsr::col16): the first entry of a
// row holds col - cbase[block], every further entry col - previous col of its row; phase 2 walks a row in stored order anyway, so
// the decode is one integer add per entry.  10 instead of 12 bytes per entry; same products, same order of additions.  Blocks a
// delta does not fit (cbase < 0: a gap >= 65536 inside a row, first columns spread over >= 65536, or one long row) read col[].
template <int OP, bool NT, bool C16>
__device__ __forceinline__ void csr_rowlane_body(const int *__restrict__ rowblk, int nblk, int remap, const int *__restrict__ rowptr,
                                                 const int *__restrict__ col, const unsigned short *__restrict__ col16,
                                                 const int *__restrict__ cbase, const double *__restrict__ val, const CsrArgs &a)
{
    __shared__ __attribute__((aligned(16))) double sval[kStreamNnz + 2];  // + 2: the run is staged from the even index at or below j0
    __shared__ __attribute__((aligned(16))) int scol[kStreamNnz + 2];
    __shared__ double red[kBlock / 64];
    int bid = remap ? xcd_remap(blockIdx.x, nblk, remap) : blockIdx.x;
    if (a.reverse && bid < nblk) bid = nblk - 1 - bid;
    if (bid >= nblk) return;  // whole workgroup leaves together
    const int tid = threadIdx.x;
    const int4 br = reinterpret_cast<const int4 *>(rowblk)[bid];
    const int r0 = br.x, r1 = br.y;
    const int nrows = r1 - r0;
    const int j0 = br.z, j1 = br.w;
    int cb = -1;
    if constexpr (C16) cb = cbase[bid];
    const bool comp = C16 && cb >= 0;  // workgroup-uniform
    const double *__restrict__ x = a.x;
    const bool has_row = tid < nrows;
    const int row = r0 + tid;
    int s = 0, e = 0;
    RowOperands o;
    if (has_row) {
        s = rowptr[row] - j0;
        e = rowptr[row + 1] - j0;
        o = load_row_operands<OP, false>(a, row);
    }
    double sum = 0.0;
    bool store = has_row;
    if (nrows == 1 && j1 - j0 > kStreamNnz) {  // one long row: strided partial sums, tree-combined (as csr_block_kernel)
        double part = 0.0, dv = 0.0;
        for (int j = j0 + tid; j < j1; j += kBlock) {
            const int c = ld_stream<NT>(col + j);
            const double v = ld_stream<NT>(val + j);
            part += v * x[c];
            if (c == r0) dv = v;  // r0, not row: every thread of the workgroup works on the one long row
        }
        part = block_sum(part, red);
        __syncthreads();
        dv = block_sum(dv, red);  // exactly one thread holds the diagonal entry (others 0): an exact sum, valid in thread 0
        o.di = dv;                // thread 0 is the only one that runs the epilogue
        sum = part;
        store = (tid == 0);
    } else {
        double dv = 0.0;
        if (comp) rowlane_stage_sum<NT, true>(j0, j1, s, e, row, cb, col, col16, val, x, sval, scol, sum, dv);
        else rowlane_stage_sum<NT, false>(j0, j1, s, e, row, cb, col, col16, val, x, sval, scol, sum, dv);
        o.di = dv;
    }
    double acc = 0.0;
    if (store) acc = row_epilogue<OP>(a, row, sum, o);
    if constexpr (op_reduces(OP)) {
        __syncthreads();
        const double t = block_sum(acc, red);
        if (tid == 0) a.partial[bid] = t;
    }
}

template <int OP, bool NT, int TAG>
__global__ __launch_bounds__(kBlock) void csr_rowlane_kernel(const int *__restrict__ rowblk, int nblk, int remap,
                                                              const int *__restrict__ rowptr, const int *__restrict__ col,
                                                              const double *__restrict__ val, CsrArgs a)
{
    csr_rowlane_body<OP, NT, false>(rowblk, nblk, remap, rowptr, col, nullptr, nullptr, val, a);
}

template <int OP, bool NT, int TAG>
__global__ __launch_bounds__(kBlock) void csr_rowlane16_kernel(const int *__restrict__ rowblk, int nblk, int remap,
                                                                const int *__restrict__ rowptr, const int *__restrict__ col,
                                                                const unsigned short *__restrict__ col16, const int *__restrict__ cbase,
                                                                const double *__restrict__ val, CsrArgs a)
{
    csr_rowlane_body<OP, NT, true>(rowblk, nblk, remap, rowptr, col, col16, cbase, val, a);
}

template <int OP, bool NT, int VEC, int TAG>
__global__ __launch_bounds__(kBlock) void csr_wave_kernel(const int *__restrict__ waveblk, int nwblk, int ngroups, int remap,
                                                           const int *__restrict__ rowptr, const int *__restrict__ col,
                                                           const double *__restrict__ val, CsrArgs a)
{
    __shared__ double prod_all[kBlock / 64][kWaveNnz];
    __shared__ double red[kBlock / 64];
    int gid = remap ? xcd_remap(blockIdx.x, ngroups, remap) : blockIdx.x;  // group of 4 wave-blocks
    if (a.reverse && gid < ngroups) gid = ngroups - 1 - gid;
    if (gid >= ngroups) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wb = gid * (kBlock / 64) + w;
    double acc = 0.0;
    if (wb < nwblk) {  // wave-uniform
        double *__restrict__ prod = prod_all[w];
        const int r0 = waveblk[wb], r1 = waveblk[wb + 1];
        const int nrows = r1 - r0;
        const int j0 = rowptr[r0], j1 = rowptr[r1];
        const double *__restrict__ x = a.x;
        const bool has_row = lane < nrows;
        const int row = r0 + lane;
        int s = 0, e = 0;
        RowOperands o;
        if (has_row) {
            s = rowptr[row] - j0;
            e = rowptr[row + 1] - j0;
            o = load_row_operands<OP>(a, row);
        }
        double sum = 0.0;
        bool store = has_row;
        if (nrows == 1 && j1 - j0 > kWaveNnz) {
            double part = 0.0;
            for (int j = j0 + lane; j < j1; j += 64) part += ld_stream<NT>(val + j) * x[ld_stream<NT>(col + j)];
            sum = wave_sum(part);
            store = (lane == 0);
        } else {
            stream_products<NT, VEC>(col, val, x, j0, j1, lane, 64, prod);
            // same wave wrote and reads: LDS executes a wave's accesses in order; the fence keeps
            // the compiler from moving the reads above the writes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            sum = row_sum_lds(prod, s, e);
        }
        if (store) acc = row_epilogue<OP>(a, row, sum, o);
    }
    if constexpr (op_reduces(OP)) {
        const double t = block_sum(acc, red);
        if (threadIdx.x == 0) a.partial[gid] = t;
    }
}

// sliced ELL: slice = 64 consecutive rows, entry k of row r at slice_ptr[slice] + k*64 + (r & 63).
// Padding entries hold (col = 0, val = 0) and are never added (k < len select), so the row sum is
// the same sequence of additions as the CSR loop.  The slice width is wave-uniform: common widths
// dispatch to a fully unrolled body whose L col loads, L val loads and L gathers are all issued
// before the first add.
template <int L, bool NT>
__device__ __forceinline__ double sell_chunk(const int *__restrict__ cp, const double *__restrict__ vp,
                                             const double *__restrict__ x, int k0, int len, int row, double &dv, double sum)
{
    int c[L];
    double v[L], xv[L];
#pragma unroll
    for (int u = 0; u < L; ++u) {
        c[u] = ld_stream<NT>(cp + (k0 + u) * 64);
        v[u] = ld_stream<NT>(vp + (k0 + u) * 64);
    }
#pragma unroll
    for (int u = 0; u < L; ++u) xv[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const double t = v[u] * xv[u];
        const bool on = k0 + u < len;
        sum = on ? sum + t : sum;
        dv = (on && c[u] == row) ? v[u] : dv;
    }
    return sum;
}

template <bool NT>
__device__ __forceinline__ double sell_row(const int *__restrict__ cp, const double *__restrict__ vp,
                                           const double *__restrict__ x, int slen, int len, int row, double &dv)
{
    double sum = 0.0;
    int k = 0;
    for (; k + 8 <= slen; k += 8) sum = sell_chunk<8, NT>(cp, vp, x, k, len, row, dv, sum);
    switch (slen - k) {  // wave-uniform
    case 7: sum = sell_chunk<7, NT>(cp, vp, x, k, len, row, dv, sum); break;
    case 6: sum = sell_chunk<6, NT>(cp, vp, x, k, len, row, dv, sum); break;
    case 5: sum = sell_chunk<5, NT>(cp, vp, x, k, len, row, dv, sum); break;
    case 4: sum = sell_chunk<4, NT>(cp, vp, x, k, len, row, dv, sum); break;
    case 3: sum = sell_chunk<3, NT>(cp, vp, x, k, len, row, dv, sum); break;
    case 2: sum = sell_chunk<2, NT>(cp, vp, x, k, len, row, dv, sum); break;
    case 1: sum = sell_chunk<1, NT>(cp, vp, x, k, len, row, dv, sum); break;
    default: break;
    }
    return sum;
}

template <int OP, bool NT, int TAG>
__global__ __launch_bounds__(kBlock) void sell_kernel(int nrow, int nslice, int ngroups, int remap,
                                                       const int *__restrict__ slice_ptr, const int *__restrict__ rowptr,
                                                       const int *__restrict__ scol, const double *__restrict__ sval, CsrArgs a)
{
    __shared__ double red[kBlock / 64];
    int gid = remap ? xcd_remap(blockIdx.x, ngroups, remap) : blockIdx.x;
    if (a.reverse && gid < ngroups) gid = ngroups - 1 - gid;
    if (gid >= ngroups) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = gid * (kBlock / 64) + w;
    const int nwork = a.slice_list ? a.nlist : nslice;
    double acc = 0.0;
    if (idx < nwork) {
        const int sl = a.slice_list ? __builtin_amdgcn_readfirstlane(a.slice_list[idx]) : idx;
        const int row = sl * 64 + lane;
        const bool has_row = row < nrow;
        const int base = __builtin_amdgcn_readfirstlane(slice_ptr[sl]);
        const int slen = (__builtin_amdgcn_readfirstlane(slice_ptr[sl + 1]) - base) >> 6;
        int len = 0;
        RowOperands o;
        if (has_row) {
            len = rowptr[row + 1] - rowptr[row];
            o = load_row_operands<OP, false>(a, row);
        }
        const double sum = sell_row<NT>(scol + base + lane, sval + base + lane, a.x, slen, len, row, o.di);
        if (has_row) acc = row_epilogue<OP>(a, row, sum, o);
    }
    if constexpr (op_reduces(OP)) {
        const double t = block_sum(acc, red);
        if (threadIdx.x == 0) a.partial[a.partial_off + gid] = t;
    }
}

// sliced diagonals: slot d of slice s holds, for every row r of the slice that has an entry in
// column r + off[d], its value.  Slots are sorted by offset, i.e. by column, so a row's present
// slots are visited in exactly its CSR entry order; absent slots are skipped by the lane mask.
template <int L, bool NT>
__device__ __forceinline__ double sdia_chunk(const int *__restrict__ off, const unsigned long long *__restrict__ mask,
                                             const int *__restrict__ vidx, const double *__restrict__ cval,
                                             const double *__restrict__ vp, const double *__restrict__ x, int d0, int row,
                                             int lane, double &dv, double sum)
{
    double v[L], xv[L];
    bool on[L];
    int oo[L];
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const int o = __builtin_amdgcn_readfirstlane(off[d0 + u]);
        oo[u] = o;
        const unsigned long long m = mask[d0 + u];  // wave-uniform address
        on[u] = (m >> lane) & 1ull;
        const int vi = __builtin_amdgcn_readfirstlane(vidx[d0 + u]);
        if (vi >= 0)  // wave-uniform: the slot owns a value block
            v[u] = ld_stream<NT>(vp + (size_t)vi * 64);
        else          // constant slot: one scalar for the whole slice
            v[u] = cval[d0 + u];
        xv[u] = x[on[u] ? row + o : row];  // clamp absent lanes onto a valid address
    }
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const double t = v[u] * xv[u];
        sum = on[u] ? sum + t : sum;
        if (oo[u] == 0) dv = on[u] ? v[u] : dv;  // wave-uniform test: the main diagonal's slot
    }
    return sum;
}

template <bool NT>
__device__ __forceinline__ double sdia_row(const int *__restrict__ off, const unsigned long long *__restrict__ mask,
                                           const int *__restrict__ vidx, const double *__restrict__ cval,
                                           const double *__restrict__ vp, const double *__restrict__ x, int nd, int row, int lane,
                                           double &dv)
{
    double sum = 0.0;
    int d = 0;
    for (; d + 8 <= nd; d += 8) sum = sdia_chunk<8, NT>(off, mask, vidx, cval, vp, x, d, row, lane, dv, sum);
    switch (nd - d) {  // wave-uniform
    case 7: sum = sdia_chunk<7, NT>(off, mask, vidx, cval, vp, x, d, row, lane, dv, sum); break;
    case 6: sum = sdia_chunk<6, NT>(off, mask, vidx, cval, vp, x, d, row, lane, dv, sum); break;
    case 5: sum = sdia_chunk<5, NT>(off, mask, vidx, cval, vp, x, d, row, lane, dv, sum); break;
    case 4: sum = sdia_chunk<4, NT>(off, mask, vidx, cval, vp, x, d, row, lane, dv, sum); break;
    case 3: sum = sdia_chunk<3, NT>(off, mask, vidx, cval, vp, x, d, row, lane, dv, sum); break;
    case 2: sum = sdia_chunk<2, NT>(off, mask, vidx, cval, vp, x, d, row, lane, dv, sum); break;
    case 1: sum = sdia_chunk<1, NT>(off, mask, vidx, cval, vp, x, d, row, lane, dv, sum); break;
    default: break;
    }
    return sum;
}

// slice without any constant slot: its value blocks are consecutive, starting at block vidx[first slot]
template <int L, bool NT>
__device__ __forceinline__ double sdia_chunk_plain(const int *__restrict__ off, const unsigned long long *__restrict__ mask,
                                                   const double *__restrict__ vp, const double *__restrict__ x, int d0, int row,
                                                   int lane, double &dv, double sum)
{
    double v[L], xv[L];
    bool on[L];
    int oo[L];
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const int o = __builtin_amdgcn_readfirstlane(off[d0 + u]);
        oo[u] = o;
        const unsigned long long m = mask[d0 + u];  // wave-uniform address
        on[u] = (m >> lane) & 1ull;
        v[u] = ld_stream<NT>(vp + (size_t)(d0 + u) * 64);
        xv[u] = x[on[u] ? row + o : row];  // clamp absent lanes onto a valid address
    }
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const double t = v[u] * xv[u];
        sum = on[u] ? sum + t : sum;
        if (oo[u] == 0) dv = on[u] ? v[u] : dv;  // wave-uniform test: the main diagonal's slot
    }
    return sum;
}

template <bool NT>
__device__ __forceinline__ double sdia_row_plain(const int *__restrict__ off, const unsigned long long *__restrict__ mask,
                                                 const double *__restrict__ vp, const double *__restrict__ x, int nd, int row, int lane,
                                                 double &dv)
{
    double sum = 0.0;
    int d = 0;
    for (; d + 8 <= nd; d += 8) sum = sdia_chunk_plain<8, NT>(off, mask, vp, x, d, row, lane, dv, sum);
    switch (nd - d) {  // wave-uniform
    case 7: sum = sdia_chunk_plain<7, NT>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 6: sum = sdia_chunk_plain<6, NT>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 5: sum = sdia_chunk_plain<5, NT>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 4: sum = sdia_chunk_plain<4, NT>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 3: sum = sdia_chunk_plain<3, NT>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 2: sum = sdia_chunk_plain<2, NT>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 1: sum = sdia_chunk_plain<1, NT>(off, mask, vp, x, d, row, lane, dv, sum); break;
    default: break;
    }
    return sum;
}

// fully constant slice (every slot folded): no value stream at all.  Offsets, lane masks and the
// slot constants are wave-uniform scalars; the lane mask is used directly as the lane predicate.
template <int L>
__device__ __forceinline__ double sdia_chunk_const(const int *__restrict__ off, const unsigned long long *__restrict__ mask,
                                                   const double *__restrict__ cval, const double *__restrict__ x, int d0, int row,
                                                   unsigned long long &dmask, double &dconst, double sum)
{
    double xv[L], cv[L];
    bool on[L];
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const int o = __builtin_amdgcn_readfirstlane(off[d0 + u]);
        const unsigned long long m = mask[d0 + u];     // wave-uniform
        on[u] = __builtin_amdgcn_inverse_ballot_w64(m);  // bit i of the mask <-> lane i
        cv[u] = cval[d0 + u];
        if (o == 0) {  // the main diagonal's slot (scalar selects)
            dmask = m;
            dconst = cv[u];
        }
        xv[u] = x[on[u] ? row + o : row];  // clamp absent lanes onto a valid address
    }
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const double t = cv[u] * xv[u];
        sum = on[u] ? sum + t : sum;
    }
    return sum;
}

__device__ __forceinline__ double sdia_row_const(const int *__restrict__ off, const unsigned long long *__restrict__ mask,
                                                 const double *__restrict__ cval, const double *__restrict__ x, int nd, int row,
                                                 double &dv)
{
    double sum = 0.0;
    unsigned long long dmask = 0ull;
    double dconst = 0.0;
    int d = 0;
    for (; d + 8 <= nd; d += 8) sum = sdia_chunk_const<8>(off, mask, cval, x, d, row, dmask, dconst, sum);
    switch (nd - d) {  // wave-uniform
    case 7: sum = sdia_chunk_const<7>(off, mask, cval, x, d, row, dmask, dconst, sum); break;
    case 6: sum = sdia_chunk_const<6>(off, mask, cval, x, d, row, dmask, dconst, sum); break;
    case 5: sum = sdia_chunk_const<5>(off, mask, cval, x, d, row, dmask, dconst, sum); break;
    case 4: sum = sdia_chunk_const<4>(off, mask, cval, x, d, row, dmask, dconst, sum); break;
    case 3: sum = sdia_chunk_const<3>(off, mask, cval, x, d, row, dmask, dconst, sum); break;
    case 2: sum = sdia_chunk_const<2>(off, mask, cval, x, d, row, dmask, dconst, sum); break;
    case 1: sum = sdia_chunk_const<1>(off, mask, cval, x, d, row, dmask, dconst, sum); break;
    default: break;
    }
    if (__builtin_amdgcn_inverse_ballot_w64(dmask)) dv = dconst;
    return sum;
}

// record path: the whole slice description sits at rec (see DevCsr::sd_rec) and is fetched by one
// batch of scalar loads before anything depends on it; count in 1..8
struct SdRecord {
    int off[8];
    unsigned long long mask[8];
    double cval[8];
    int count;
};

__device__ __forceinline__ SdRecord load_sd_record(const int *__restrict__ rec)
{
    SdRecord r;
    const unsigned long long *m = reinterpret_cast<const unsigned long long *>(rec + 8);
    const double *c = reinterpret_cast<const double *>(rec + 24);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        r.off[u] = rec[u];
        r.mask[u] = m[u];
        r.cval[u] = c[u];
    }
    r.count = rec[40];
    // keep the four scalar loads together ahead of the branch on count: without this the compiler
    // sinks the three wide ones below that branch and the wave pays two dependent round trips
    asm volatile("" ::"s"(r.count), "s"(r.off[0]), "s"(r.off[1]), "s"(r.off[2]), "s"(r.off[3]), "s"(r.off[4]), "s"(r.off[5]), "s"(r.off[6]),
                 "s"(r.off[7]), "s"(r.mask[0]), "s"(r.mask[1]), "s"(r.mask[2]), "s"(r.mask[3]), "s"(r.mask[4]), "s"(r.mask[5]), "s"(r.mask[6]),
                 "s"(r.mask[7]), "s"(r.cval[0]), "s"(r.cval[1]), "s"(r.cval[2]), "s"(r.cval[3]), "s"(r.cval[4]), "s"(r.cval[5]), "s"(r.cval[6]),
                 "s"(r.cval[7]));
    return r;
}

template <int L>
__device__ __forceinline__ double sdia_rec_apply(const SdRecord &r, const double *__restrict__ x, int row, double &dv)
{
    double xv[L];
    bool on[L];
    unsigned long long dmask = 0ull;
    double dconst = 0.0;
#pragma unroll
    for (int u = 0; u < L; ++u) {
        on[u] = __builtin_amdgcn_inverse_ballot_w64(r.mask[u]);  // bit i of the mask <-> lane i
        if (r.off[u] == 0) {  // the main diagonal's slot (scalar selects)
            dmask = r.mask[u];
            dconst = r.cval[u];
        }
        xv[u] = x[on[u] ? row + r.off[u] : row];  // clamp absent lanes onto a valid address
    }
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const double t = r.cval[u] * xv[u];
        sum = on[u] ? sum + t : sum;
    }
    if (__builtin_amdgcn_inverse_ballot_w64(dmask)) dv = dconst;
    return sum;
}

__device__ __forceinline__ double sdia_row_rec(const SdRecord &r, const double *__restrict__ x, int row, double &dv)
{
    switch (r.count) {  // wave-uniform
    case 8: return sdia_rec_apply<8>(r, x, row, dv);
    case 7: return sdia_rec_apply<7>(r, x, row, dv);
    case 6: return sdia_rec_apply<6>(r, x, row, dv);
    case 5: return sdia_rec_apply<5>(r, x, row, dv);
    case 4: return sdia_rec_apply<4>(r, x, row, dv);
    case 3: return sdia_rec_apply<3>(r, x, row, dv);
    case 2: return sdia_rec_apply<2>(r, x, row, dv);
    case 1: return sdia_rec_apply<1>(r, x, row, dv);
    default: return 0.0;
    }
}

template <int OP, bool NT, int TAG>
__global__ __launch_bounds__(kBlock) void sdia_kernel(int nrow, int nslice, int ngroups, int remap, const int *__restrict__ sd_ptr,
                                                       const int *__restrict__ sd_off, const unsigned long long *__restrict__ sd_mask,
                                                       const int *__restrict__ sd_vidx, const double *__restrict__ sd_cval,
                                                       const double *__restrict__ sd_val, const int *__restrict__ sd_rec, CsrArgs a)
{
    __shared__ double red[kBlock / 64];
    int gid = remap ? xcd_remap(blockIdx.x, ngroups, remap) : blockIdx.x;
    if (a.reverse && gid < ngroups) gid = ngroups - 1 - gid;
    if (gid >= ngroups) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = __builtin_amdgcn_readfirstlane(gid * (kBlock / 64) + w);  // wave-uniform: keeps the slice metadata on the scalar path
    const int nwork = a.slice_list ? a.nlist : nslice;
    double acc = 0.0;
    if (idx < nwork) {
        const int sl = a.slice_list ? __builtin_amdgcn_readfirstlane(a.slice_list[idx]) : idx;
        int row = sl * 64 + lane;
        const bool has_row = row < nrow;
        if (!has_row) row = nrow - 1;  // tail lanes of the last slice: masks are clear, keep addresses valid
        RowOperands o;
        if (has_row) o = load_row_operands<OP, false>(a, row);
        double sum;
        bool done = false;
        if (sd_rec) {  // value-free slices are described by fixed-stride records
            const SdRecord r = load_sd_record(sd_rec + (size_t)sl * kSdRecInts);
            if (r.count >= 0) {  // wave-uniform
                sum = sdia_row_rec(r, a.x, row, o.di);
                done = true;
            }
        }
        if (!done) {
            const int p0 = __builtin_amdgcn_readfirstlane(sd_ptr[sl]);
            const int s0 = p0 & kSdPtrMask;
            const int nd = (__builtin_amdgcn_readfirstlane(sd_ptr[sl + 1]) & kSdPtrMask) - s0;
            if (p0 & kSdConstBit) {  // every slot of this slice is a constant slot (more than 8 of them)
                sum = sdia_row_const(sd_off + s0, sd_mask + s0, sd_cval + s0, a.x, nd, row, o.di);
            } else if (p0 & kSdPlainBit) {  // none is: consecutive value blocks, no per-slot indirection
                const int v0 = __builtin_amdgcn_readfirstlane(sd_vidx[s0]);
                sum = sdia_row_plain<NT>(sd_off + s0, sd_mask + s0, sd_val + (size_t)v0 * 64 + lane, a.x, nd, row, lane, o.di);
            } else {
                sum = sdia_row<NT>(sd_off + s0, sd_mask + s0, sd_vidx + s0, sd_cval + s0, sd_val + lane, a.x, nd, row, lane, o.di);
            }
        }
        if (has_row) acc = row_epilogue<OP>(a, row, sum, o);
    }
    if constexpr (op_reduces(OP)) {
        const double t = block_sum(acc, red);
        if (threadIdx.x == 0) a.partial[a.partial_off + gid] = t;
    }
}

// Table path (DevCsr::sd_tab): offsets and constants are kernel arguments, so the x gathers leave
// at once (out-of-range rows clamped; lanes without the entry are dropped by the mask afterwards).
template <int L>
__device__ __forceinline__ double sdia_tab_apply(const SdTable &tab, const unsigned long long *__restrict__ mrec, const double *__restrict__ x,
                                                 int row, int xlen, double &dv, bool &conform, const int *__restrict__ conf_ptr)
{
    double xv[L];
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const int idx = row + tab.off[u];
        xv[u] = x[(unsigned)idx < (unsigned)xlen ? idx : row];  // xlen = entries of x (own + halo on a rank-local block)
    }
    unsigned long long m[L];
#pragma unroll
    for (int u = 0; u < L; ++u) m[u] = mrec[u];  // wave-uniform: scalar loads, issued beside the gathers
    conform = __builtin_amdgcn_readfirstlane(*conf_ptr) != 0;
    double sum = 0.0;
    unsigned long long dmask = 0ull;
    double dconst = 0.0;
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const bool on = __builtin_amdgcn_inverse_ballot_w64(m[u]);
        if (tab.off[u] == 0) {
            dmask = m[u];
            dconst = tab.cval[u];
        }
        const double t = tab.cval[u] * xv[u];
        sum = on ? sum + t : sum;
    }
    if (__builtin_amdgcn_inverse_ballot_w64(dmask)) dv = dconst;
    return sum;
}

// value of lane-1 / lane+1 across the whole wave (DPP wave shifts); `edge` fills the lane that has
// no neighbour (lane 0 resp. lane 63)
__device__ __forceinline__ double lane_from_below(double v, double edge)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_from_above(double v, double edge)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(edge), __double2loint(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(edge), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Table path for stencils whose offsets -1, 0, +1 sit in slots C0-1, C0, C0+1 (every grid stencil in
// lexicographic order).  The kernel is bound by the number of wave-level memory instructions
// (profiles/r01_band_experiment_vmem_bound.txt: ~2.7 us per gather at 216^3 whether it hits L1 or
// not), so the three gathers of x[r-1], x[r], x[r+1] become ONE: the neighbours come from the lanes
// next door by DPP, the two values beyond the slice's ends by scalar loads; x_i for the Jacobi /
// dot epilogues is that same centre value instead of a load of its own.
template <int L, int C0>
__device__ __forceinline__ double sdia_tab_apply_near(const SdTable &tab, const unsigned long long *__restrict__ mrec,
                                                      const double *__restrict__ x, int row, int row0, int xlen, double &dv, double &xc,
                                                      bool &conform, const int *__restrict__ conf_ptr)
{
    double xv[L];
#pragma unroll
    for (int u = 0; u < L; ++u) {
        if (u == C0 - 1 || u == C0 + 1) continue;
        const int idx = row + tab.off[u];
        xv[u] = x[(unsigned)idx < (unsigned)xlen ? idx : row];
    }
    // beyond the slice's ends (wave-uniform addresses; clamped, the masks drop what does not exist)
    const double below = x[row0 > 0 ? row0 - 1 : 0];
    const double above = x[row0 + 64 < xlen ? row0 + 64 : xlen - 1];
    unsigned long long m[L];
#pragma unroll
    for (int u = 0; u < L; ++u) m[u] = mrec[u];  // wave-uniform: scalar loads, issued beside the gathers
    conform = __builtin_amdgcn_readfirstlane(*conf_ptr) != 0;
    xc = xv[C0];
    xv[C0 - 1] = lane_from_below(xv[C0], below);
    xv[C0 + 1] = lane_from_above(xv[C0], above);
    double sum = 0.0;
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const bool on = __builtin_amdgcn_inverse_ballot_w64(m[u]);
        const double t = tab.cval[u] * xv[u];
        sum = on ? sum + t : sum;
    }
    if (__builtin_amdgcn_inverse_ballot_w64(m[C0])) dv = tab.cval[C0];
    return sum;
}

// a slice that is off its level's stencil table: through its record / slot headers (as in sdia_kernel)
template <bool NT>
__device__ __forceinline__ double sdia_offtable_row(int sl, int row, int lane, const double *__restrict__ x, const int *__restrict__ sd_ptr,
                                                    const int *__restrict__ sd_off, const unsigned long long *__restrict__ sd_mask,
                                                    const int *__restrict__ sd_vidx, const double *__restrict__ sd_cval,
                                                    const double *__restrict__ sd_val, const int *__restrict__ sd_rec, double &dv)
{
    if (sd_rec) {
        const SdRecord r = load_sd_record(sd_rec + (size_t)sl * kSdRecInts);
        if (r.count >= 0) return sdia_row_rec(r, x, row, dv);
    }
    const int p0 = __builtin_amdgcn_readfirstlane(sd_ptr[sl]);
    const int s0 = p0 & kSdPtrMask;
    const int nd = (__builtin_amdgcn_readfirstlane(sd_ptr[sl + 1]) & kSdPtrMask) - s0;
    if (p0 & kSdConstBit) return sdia_row_const(sd_off + s0, sd_mask + s0, sd_cval + s0, x, nd, row, dv);
    if (p0 & kSdPlainBit) {
        const int v0 = __builtin_amdgcn_readfirstlane(sd_vidx[s0]);
        return sdia_row_plain<NT>(sd_off + s0, sd_mask + s0, sd_val + (size_t)v0 * 64 + lane, x, nd, row, lane, dv);
    }
    return sdia_row<NT>(sd_off + s0, sd_mask + s0, sd_vidx + s0, sd_cval + s0, sd_val + lane, x, nd, row, lane, dv);
}

template <int OP, bool NT, int TAG>
__global__ __launch_bounds__(kBlock) void sdia_tab_kernel(int nrow, int xlen, int nslice, int ngroups, int remap, SdTable tab,
                                                           const int *__restrict__ slist,  // optional list of slices (interior / boundary launches)
                                                           const unsigned long long *__restrict__ sd_tmask, const int *__restrict__ sd_tconf,
                                                           const int *__restrict__ sd_ptr, const int *__restrict__ sd_off,
                                                           const unsigned long long *__restrict__ sd_mask, const int *__restrict__ sd_vidx,
                                                           const double *__restrict__ sd_cval, const double *__restrict__ sd_val,
                                                           const int *__restrict__ sd_rec, CsrArgs a)
{
    __shared__ double red[kBlock / 64];
    int gid = remap ? xcd_remap(blockIdx.x, ngroups, remap) : blockIdx.x;
    if (a.reverse && gid < ngroups) gid = ngroups - 1 - gid;
    if (gid >= ngroups) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int idx = __builtin_amdgcn_readfirstlane(gid * (kBlock / 64) + w);
    double acc = 0.0;
    if (idx < nslice) {  // nslice = number of work items (list length when a list is given)
        const int sl = slist ? slist[idx] : idx;  // wave-uniform address: scalar load
        int row = sl * 64 + lane;
        const bool has_row = row < nrow;
        // lanes past the launched rows keep their own index while it is inside the vector: on a prefix launch of a
        // deep-halo operator they are real rows, and the lane next door takes x[r+1] from their centre gather
        if (!has_row) row = row < xlen ? row : xlen - 1;
        const unsigned long long *mrec = sd_tmask + (size_t)sl * 8;
        const int *cp = sd_tconf + sl;
        bool conform = false;
        double sum = 0.0;
        RowOperands o;
        double dv = 0.0;
        double dpair = 1.0;  // RESID_PAIR: diagonal of the coarse row this even lane writes (in flight beside the gathers)
        if constexpr (OP == OP_RESID_PAIR) {
            if (!a.d || !a.y2) dpair = a.dconst;
            else if (has_row && !(lane & 1)) dpair = a.d[row >> 1];
        }
        // lexicographic grid stencils: -1, 0, +1 in adjacent slots (launch-uniform test on kernel arguments)
        const int near = tab.near;
        if (near) {
            if (has_row) o = load_row_operands<OP, false, false>(a, row);  // x_i comes with the centre gather
            double xc = 0.0;
            if (near == 73) sum = sdia_tab_apply_near<7, 3>(tab, mrec, a.x, row, sl * 64, xlen, dv, xc, conform, cp);
            else if (near == 52) sum = sdia_tab_apply_near<5, 2>(tab, mrec, a.x, row, sl * 64, xlen, dv, xc, conform, cp);
            else sum = sdia_tab_apply_near<3, 1>(tab, mrec, a.x, row, sl * 64, xlen, dv, xc, conform, cp);
            if constexpr (op_needs_xi(OP)) o.xi = xc;
        } else {
        if (has_row) o = load_row_operands<OP, false>(a, row);
        dv = o.di;
        switch (tab.nd) {  // uniform over the whole launch
        case 8: sum = sdia_tab_apply<8>(tab, mrec, a.x, row, xlen, dv, conform, cp); break;
        case 7: sum = sdia_tab_apply<7>(tab, mrec, a.x, row, xlen, dv, conform, cp); break;
        case 6: sum = sdia_tab_apply<6>(tab, mrec, a.x, row, xlen, dv, conform, cp); break;
        case 5: sum = sdia_tab_apply<5>(tab, mrec, a.x, row, xlen, dv, conform, cp); break;
        case 4: sum = sdia_tab_apply<4>(tab, mrec, a.x, row, xlen, dv, conform, cp); break;
        case 3: sum = sdia_tab_apply<3>(tab, mrec, a.x, row, xlen, dv, conform, cp); break;
        case 2: sum = sdia_tab_apply<2>(tab, mrec, a.x, row, xlen, dv, conform, cp); break;
        default: sum = sdia_tab_apply<1>(tab, mrec, a.x, row, xlen, dv, conform, cp); break;
        }
        }
        if (conform) {
            o.di = dv;
        } else {  // rare: a slice off the level's stencil goes through its record / slot headers
            if constexpr (op_needs_xi(OP)) o.xi = a.x[row];
            sum = sdia_offtable_row<NT>(sl, row, lane, a.x, sd_ptr, sd_off, sd_mask, sd_vidx, sd_cval, sd_val, sd_rec, o.di);
        }
        if constexpr (OP == OP_RESID_PAIR) {
            // aggregates are the row pairs (2J, 2J+1) and a slice starts on an even row: the partner's residual comes
            // from the lane above.  Same expressions, same order as OP_RESID + restrict_agg_zero_kernel:
            // r = 1.0*b + (-1.0)*s ; b_c = (0 + r_2J) + r_2J+1 ; x_c = omega*b_c/d_c
            const double ri = has_row ? 1.0 * o.bi + (-1.0) * sum : 0.0;
            const double rn = lane_from_above(ri, 0.0);
            if (has_row && !(lane & 1)) {
                double bc = 0.0 + ri;
                if (row + 1 < nrow) bc = bc + rn;
                a.y[row >> 1] = bc;
                if (a.y2) a.y2[row >> 1] = a.omega * bc / dpair;
            }
        } else {
            if (has_row) acc = row_epilogue<OP>(a, row, sum, o);
        }
    }
    if constexpr (op_reduces(OP)) {
        const double t = block_sum(acc, red);
        if (threadIdx.x == 0) a.partial[a.partial_off + gid] = t;
    }
}

// LDS-tiled table path ("sdia_tile_kernel").  The table kernel above is bound by the bytes L2 delivers
// to the CUs: every row pulls x[r], x[r +- line], x[r +- plane] through L1 (~49 B/row of reads, every
// far gather an L1 miss; profiles/r01_pmc_diag/).  Here a workgroup of 16 waves owns T = 1024 S consecutive
// rows; every wave loads the centre values of its own slices (coalesced, kept in registers) and parks
// them in LDS, the first threads add the halo x[r0 - line, r0) and x[r0 + T, r0 + T + line) (north_star:
// "LDS staging of x-vector tiles"): after one barrier the +-1 and +-line neighbours of every row come out
// of LDS, only the +-plane neighbours (3D stencils) are still gathered from L2 -- issued, like b and the
// lane masks, before the barrier so their latency overlaps the staging.  Reads per row:
// 8 (1 + 2 line / T) + 16 + 8 (b) + 1 (masks) ~ 36 B instead of 49.  Same products, same order of additions
// as every other family.  Launch: whole level only (no slice lists); LDS = (T + lo + hi) doubles.
constexpr int kTileBlock = 1024;

template <int OP, int ND, int S, int TAG>
__global__ __launch_bounds__(kTileBlock) void sdia_tile_kernel(int nrow, int xlen, int ntile, int remap, SdTable tab,
                                                                const unsigned long long *__restrict__ sd_tmask, const int *__restrict__ sd_tconf,
                                                                const int *__restrict__ sd_ptr, const int *__restrict__ sd_off,
                                                                const unsigned long long *__restrict__ sd_mask, const int *__restrict__ sd_vidx,
                                                                const double *__restrict__ sd_cval, const double *__restrict__ sd_val,
                                                                const int *__restrict__ sd_rec, CsrArgs a)
{
    extern __shared__ double xt[];
    __shared__ double red[kTileBlock / 64];
    constexpr int C0 = ND / 2;
    constexpr int T = kTileBlock * S;
    constexpr int NW = kTileBlock / 64;
    int tile = remap ? xcd_remap(blockIdx.x, ntile, remap) : blockIdx.x;
    if (a.reverse && tile < ntile) tile = ntile - 1 - tile;
    if (tile >= ntile) return;  // whole workgroup leaves together
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lo = -tab.off[C0 - 2], hi = tab.off[C0 + 2];  // reach of the +-line neighbours (kernel arguments)
    const int r0 = tile * T;
    const double *__restrict__ x = a.x;
    int row[S];
    bool valid[S], has_row[S], conform[S];
    unsigned long long m[S][ND];
    RowOperands o[S];
    double xc[S], xm[S], xp[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int k = w + NW * s;
        const int sl = (r0 >> 6) + k;
        valid[s] = sl * 64 < nrow;  // wave-uniform
        const int slc = valid[s] ? sl : 0;
        int r = slc * 64 + lane;
        has_row[s] = valid[s] && r < nrow;
        if (r >= nrow) r = nrow - 1;
        row[s] = r;
        const unsigned long long *mrec = sd_tmask + (size_t)slc * 8;
#pragma unroll
        for (int u = 0; u < ND; ++u) m[s][u] = mrec[u];  // wave-uniform: scalar loads
        conform[s] = __builtin_amdgcn_readfirstlane(sd_tconf[slc]) != 0;
        if (has_row[s]) o[s] = load_row_operands<OP, false, false>(a, r);
        xm[s] = 0.0;
        xp[s] = 0.0;
        if constexpr (ND == 7) {  // +-plane neighbours: gathered through L1/L2, in flight across the barrier
            const int im = r + tab.off[0], ip = r + tab.off[6];
            xm[s] = x[(unsigned)im < (unsigned)xlen ? im : r];
            xp[s] = x[(unsigned)ip < (unsigned)xlen ? ip : r];
        }
        xc[s] = x[r];
        xt[lo + k * 64 + lane] = xc[s];
    }
    // halo of the window: x[r0 - lo, r0) in front, x[r0 + T, r0 + T + hi) behind (clamped into the vector;
    // entries outside it are never used: the lane masks of rows that would need them are clear)
    for (int i = threadIdx.x; i < lo + hi; i += kTileBlock) {
        const int idx = i < lo ? r0 - lo + i : r0 + T + (i - lo);
        const int pos = i < lo ? i : T + i;
        xt[pos] = x[idx < 0 ? 0 : (idx < xlen ? idx : xlen - 1)];
    }
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (!valid[s]) continue;  // wave-uniform
        const int r = row[s];
        double sum = 0.0;
        if (conform[s]) {
            const int li = lo + (r - r0);  // position of the row inside the window
            double xv[ND];
            if constexpr (ND == 7) {
                xv[0] = xm[s];
                xv[6] = xp[s];
            }
            xv[C0 - 2] = xt[li - lo];
            xv[C0 - 1] = xt[li - 1];
            xv[C0] = xc[s];
            xv[C0 + 1] = xt[li + 1];
            xv[C0 + 2] = xt[li + hi];
#pragma unroll
            for (int u = 0; u < ND; ++u) {
                const bool on = __builtin_amdgcn_inverse_ballot_w64(m[s][u]);
                const double t = tab.cval[u] * xv[u];
                sum = on ? sum + t : sum;
            }
            o[s].di = __builtin_amdgcn_inverse_ballot_w64(m[s][C0]) ? tab.cval[C0] : 0.0;
            if constexpr (op_needs_xi(OP)) o[s].xi = xc[s];
        } else {  // rare: a slice off the level's stencil goes through its record / slot headers
            if constexpr (op_needs_xi(OP)) o[s].xi = xc[s];
            sum = sdia_offtable_row<false>(r >> 6, r, lane, x, sd_ptr, sd_off, sd_mask, sd_vidx, sd_cval, sd_val, sd_rec, o[s].di);
        }
        if (has_row[s]) acc += row_epilogue<OP>(a, r, sum, o[s]);
    }
    if constexpr (op_reduces(OP)) {
        acc = wave_sum(acc);
        if (lane == 0) red[w] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < NW; ++q) t += red[q];
            a.partial[a.partial_off + tile] = t;
        }
    }
}

// ------------------------------------------------------------------ double sweep on box grids
// Two Jacobi sweeps in one pass over x, b and y (temporal blocking).  The single-sweep table kernel at 216^3 moves 24 B per
// row and sweep at the rate HBM delivers (profiles/r03_pmc); the only way below that is to not write the first sweep's result.
// A workgroup of 1024 threads owns TY grid lines of every plane of a chunk of CZ planes and marches through the planes:
//   region in a plane = lines j0-2 .. j0+TY+1 (contiguous in memory), thread t owns the points p = t + 1024 q, q < Q;
//   x0 of plane k sits in LDS (X0) for the in-plane neighbours, x0 of planes k-1 / k+1 at the own point in registers;
//   step k: x1 = J(x0) on plane k, lines 1 .. TY+2 of the region (one ring more than the tile: recomputed, not exchanged --
//           the same expression on the same operands gives the neighbour workgroup's bits);
//           x2 = J(x1) on plane k-1, lines 2 .. TY+1, from x1 of plane k-1 in LDS (X1) and of planes k-2 / k in registers; stored.
// Neighbours that do not exist are read as +0.0 -- a zero pad cell between the lines in LDS, zero lines and planes outside the
// grid: c * 0.0 = +-0.0 and sum + (+-0.0) == sum bit for bit (sum is never -0.0: it starts at +0.0, and x + y = -0.0 only for
// x = y = -0.0), which is exactly "skip the missing entry" of the table kernel, without a predicate.  Every row is computed with
// the table kernel's products in the table kernel's order, so y is bitwise what two OP_JACOBI launches produce.
// Reads per row and double sweep: x (TY+4)/TY * (CZ+3)/CZ, b (TY+2)/TY * (CZ+2)/CZ, one store: ~30 B instead of 48.
constexpr int kBoxBlock = 1024;

// a / b for a divisor known before the loop: the compiler's own fp64 division sequence -- v_div_scale of both operands, v_rcp_f64 and two
// Newton steps on the scaled divisor, q0 = a_s r, rem = fma(-b_s, q0, a_s), v_div_fmas, v_div_fixup -- with the part that depends on b
// alone, the refined reciprocal of the scaled divisor, computed once per thread instead of once per row and sweep (5 of the 12
// instructions, among them the quarter-rate v_rcp_f64).  Where v_div_scale would scale b differently for this a (denormal or zero
// operands, exponents ~2^1000 apart: not residual-sized numbers) the whole wave takes the plain division.  Same instructions on the
// same operands, hence bitwise the plain division's result (tools/micro/divtest: 1.7e8 random quotients incl. denormals, infinities
// and NaNs against the compiler's division and against the host's, no difference).
struct DivConst {
    double b, bs0, r0;
};
__device__ __forceinline__ DivConst make_div_const(double b)
{
    DivConst c;
    c.b = b;
    bool f;
    c.bs0 = __builtin_amdgcn_div_scale(1.0, b, false, &f);
    double r = __builtin_amdgcn_rcp(c.bs0);
    double e = __builtin_fma(-c.bs0, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-c.bs0, r, 1.0);
    r = __builtin_fma(r, e, r);
    c.r0 = r;
    return c;
}
__device__ __forceinline__ double div_const(double a, const DivConst &c)
{
    bool fd, fn;
    const double bs = __builtin_amdgcn_div_scale(a, c.b, false, &fd);
    const double as = __builtin_amdgcn_div_scale(a, c.b, true, &fn);
    if (__builtin_amdgcn_ballot_w64(bs != c.bs0) != 0ull) return a / c.b;  // wave-uniform
    const double q0 = as * c.r0;
    const double rem = __builtin_fma(-bs, q0, as);
    const double q = __builtin_amdgcn_div_fmas(rem, c.r0, q0, fn);
    return __builtin_amdgcn_div_fixup(q, c.b, a);
}

struct BoxArgs {
    int nx, ny, nz;
    int TY, CZ, ytiles;
    double c[7];  // -plane, -line, -1, 0, +1, +line, +plane
    double omega;
};

// ZERO: the leg starts from a zero guess -- x0 is not read: the first (matrix-free) sweep x1 = omega b / d is evaluated where x0 would be
// loaded, from the right-hand side the stages need anyway, so the launch performs the first THREE sweeps of the leg (y = J(J(omega b / d)))
template <int Q, int TAG, bool ZERO>
__global__ __launch_bounds__(kBoxBlock) void sdia_box2_kernel(BoxArgs g, const double *__restrict__ x, const double *__restrict__ b,
                                                               double *__restrict__ y)
{
    extern __shared__ double box_lds[];
    const int nx = g.nx, ny = g.ny, nz = g.nz, P = nx * ny, TY = g.TY;
    const int R0 = (TY + 4) * nx;
    const int pitch = nx + 1;
    const int cells = (TY + 4) * pitch + 1;
    double *X0 = box_lds;          // x0, plane k, whole region
    double *X1 = box_lds + cells;  // x1, plane k-1
    // workgroups go round-robin to the 8 XCDs: every XCD gets a contiguous run of (chunk, tile) pairs, so tiles that share
    // halo lines sit behind the same L2
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x, per = nwg / 8, rem = nwg % 8;
        const int c = wg % 8, r = wg / 8;
        wg = c * per + min(c, rem) + r;
    }
    const int tile = wg % g.ytiles, zc = wg / g.ytiles;
    const int j0 = tile * TY;
    const int z0 = zc * g.CZ, z1 = min(z0 + g.CZ, nz);
    const long base = (long)(j0 - 2) * nx;  // + k*P + p = global row
    const int tid = threadIdx.x;
    const double c0 = g.c[0], c1 = g.c[1], c2 = g.c[2], c3 = g.c[3], c4 = g.c[4], c5 = g.c[5], c6 = g.c[6], om = g.omega;
    const DivConst dc = make_div_const(c3);
    for (int i = tid; i < 2 * cells; i += kBoxBlock) box_lds[i] = 0.0;
    bool v0[Q], v1[Q], v2[Q];
    int sidx[Q];
    double xm[Q], xc[Q], xp[Q], bk[Q], bp[Q], x1m[Q], x1c[Q];
    double bq[Q];  // ZERO: b of plane k + 1 (it arrived as the source of xp)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int p = tid + kBoxBlock * q;
        const int lr = p / nx;
        const int jr = j0 - 2 + lr;
        sidx[q] = p + lr + 1;
        v0[q] = p < R0 && jr >= 0 && jr < ny;
        v1[q] = v0[q] && lr >= 1 && lr <= TY + 2;
        v2[q] = v0[q] && lr >= 2 && lr <= TY + 1;
        xm[q] = xc[q] = xp[q] = bk[q] = bp[q] = x1m[q] = x1c[q] = bq[q] = 0.0;
    }
    const int ks = z0 - 1;  // first plane of the first sweep (-1: does not exist)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int p = tid + kBoxBlock * q;
        if constexpr (ZERO) {
            if (v0[q]) {
                if (ks >= 1) xm[q] = div_const(om * b[(long)(ks - 1) * P + base + p], dc);
                if (ks >= 0) {
                    bk[q] = b[(long)ks * P + base + p];
                    xc[q] = div_const(om * bk[q], dc);
                }
                if (ks + 1 < nz) {
                    bq[q] = b[(long)(ks + 1) * P + base + p];
                    xp[q] = div_const(om * bq[q], dc);
                }
            }
        } else {
            if (v0[q]) {
                if (ks >= 1) xm[q] = x[(long)(ks - 1) * P + base + p];
                if (ks >= 0) xc[q] = x[(long)ks * P + base + p];
                if (ks + 1 < nz) xp[q] = x[(long)(ks + 1) * P + base + p];
            }
            if (v1[q] && ks >= 0) bk[q] = b[(long)ks * P + base + p];
        }
    }
    __syncthreads();
    for (int k = ks; k <= z1; ++k) {  // every thread of the workgroup runs the same z1 - ks + 1 steps
        const bool plane = k >= 0 && k < nz;  // uniform
        double xn[Q], bn[Q];  // operands of the next step: in flight across this one
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int p = tid + kBoxBlock * q;
            xn[q] = 0.0;
            bn[q] = 0.0;
            if (k + 1 <= z1) {
                if constexpr (ZERO) {
                    if (v0[q] && k + 2 < nz) xn[q] = b[(long)(k + 2) * P + base + p];  // raw b of plane k + 2 (x of that plane follows from it below)
                } else {
                    if (v0[q] && k + 2 < nz) xn[q] = x[(long)(k + 2) * P + base + p];
                    if (v1[q] && k + 1 < nz) bn[q] = b[(long)(k + 1) * P + base + p];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (v0[q]) X0[sidx[q]] = xc[q];
        __syncthreads();
        double x1k[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            x1k[q] = 0.0;
            if (plane && v1[q]) {
                const int s = sidx[q];
                double sum = 0.0;
                sum = sum + c0 * xm[q];
                sum = sum + c1 * X0[s - pitch];
                sum = sum + c2 * X0[s - 1];
                sum = sum + c3 * xc[q];
                sum = sum + c4 * X0[s + 1];
                sum = sum + c5 * X0[s + pitch];
                sum = sum + c6 * xp[q];
                const double h = 1.0 * bk[q] + (-1.0) * sum;
                x1k[q] = xc[q] + div_const(om * h, dc);
            }
        }
        const int k2 = k - 1;
        if (k2 >= z0 && k2 < z1) {  // uniform
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int p = tid + kBoxBlock * q;
                if (v2[q]) {
                    const int s = sidx[q];
                    double sum = 0.0;
                    sum = sum + c0 * x1m[q];
                    sum = sum + c1 * X1[s - pitch];
                    sum = sum + c2 * X1[s - 1];
                    sum = sum + c3 * x1c[q];
                    sum = sum + c4 * X1[s + 1];
                    sum = sum + c5 * X1[s + pitch];
                    sum = sum + c6 * x1k[q];
                    const double h = 1.0 * bp[q] + (-1.0) * sum;
                    y[(long)k2 * P + base + p] = x1c[q] + div_const(om * h, dc);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (v1[q]) X1[sidx[q]] = x1k[q];
            x1m[q] = x1c[q];
            x1c[q] = x1k[q];
            xm[q] = xc[q];
            xc[q] = xp[q];
            bp[q] = bk[q];
            if constexpr (ZERO) {
                bk[q] = bq[q];
                bq[q] = xn[q];
                xp[q] = 0.0;
                if (v0[q] && k + 1 <= z1 && k + 2 < nz) xp[q] = div_const(om * xn[q], dc);
            } else {
                xp[q] = xn[q];
                bk[q] = bn[q];
            }
        }
    }
}

// One stencil application per launch on the same marching-plane scheme (the launches of a box-grid level that carry an epilogue of their
// own: SpMV with the p.Ap dot, the last post-sweep with the z.r dot or with the prolongation, the residual with the pair restriction).
// x of plane k in LDS for the in-plane neighbours, planes k -+ 1 at the own point in registers: every x is read once (+ one halo line per
// side of the tile and one halo plane per end of the chunk) instead of being gathered by seven rows through L1 / L2; nothing is
// recomputed.  Products in table order, the table kernel's epilogue expressions: the stored vectors are bitwise the table kernel's.  The
// fused dot products are summed per workgroup of this kernel (one partial each) instead of per 256 rows: same terms, another order of
// additions -- the one place where the two paths differ, in the last bits of a reduction (reductions are held to 1e-12, section 2).
enum BoxEpi : int { BOX_SPMV_DOT = 0, BOX_JACOBI_DOT = 1, BOX_RESID_PAIRX = 2, BOX_JACOBI_PROLONG = 3, BOX_JACOBI = 4 };

struct Box1Args {
    const double *x, *b;     // input vector, right-hand side (not read by SPMV_DOT)
    double *y;               // SPMV_DOT: A x; JACOBI_DOT: the sweep's result; RESID_PAIRX: coarse right-hand side
    double *y2;              // RESID_PAIRX: coarse zero-guess sweep; JACOBI_PROLONG: the finer level's iterate
    const double *dc;        // RESID_PAIRX: coarse diagonal (nullptr: dconst)
    double dconst;
    const int *members;      // JACOBI_PROLONG: (first, second) fine rows per row; nullptr = rows (2i, 2i+1)
    int nfine;
    double *partial;         // one per workgroup (the reducing epilogues)
};

template <int Q, int EPI, int TAG>
__global__ __launch_bounds__(kBoxBlock) void sdia_box1_kernel(BoxArgs g, Box1Args a)
{
    extern __shared__ double box_lds[];
    __shared__ double red[kBoxBlock / 64];
    const int nx = g.nx, ny = g.ny, nz = g.nz, P = nx * ny, TY = g.TY;
    const int R0 = (TY + 2) * nx;
    const int pitch = nx + 1;
    const int cells = (TY + 2) * pitch + 1;
    double *X0 = box_lds;
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x, per = nwg / 8, rem = nwg % 8;
        const int c = wg % 8, r = wg / 8;
        wg = c * per + min(c, rem) + r;
    }
    const int tile = wg % g.ytiles, zc = wg / g.ytiles;
    const int j0 = tile * TY;
    const int z0 = zc * g.CZ, z1 = min(z0 + g.CZ, nz);
    const long base = (long)(j0 - 1) * nx;
    const int tid = threadIdx.x;
    const double c0 = g.c[0], c1 = g.c[1], c2 = g.c[2], c3 = g.c[3], c4 = g.c[4], c5 = g.c[5], c6 = g.c[6], om = g.omega;
    constexpr bool kNeedsB = EPI != BOX_SPMV_DOT;
    constexpr bool kDivides = EPI == BOX_JACOBI_DOT || EPI == BOX_JACOBI_PROLONG || EPI == BOX_JACOBI;
    DivConst dc3 = {1.0, 1.0, 1.0};
    if constexpr (kDivides) dc3 = make_div_const(c3);
    for (int i = tid; i < cells; i += kBoxBlock) X0[i] = 0.0;
    bool v0[Q], v1[Q];
    int sidx[Q];
    double xm[Q], xc[Q], xp[Q], bk[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int p = tid + kBoxBlock * q;
        const int lr = p / nx;
        const int jr = j0 - 1 + lr;
        sidx[q] = p + lr + 1;
        v0[q] = p < R0 && jr >= 0 && jr < ny;
        v1[q] = v0[q] && lr >= 1 && lr <= TY;
        xm[q] = xc[q] = xp[q] = bk[q] = 0.0;
        if (v0[q]) {
            if (z0 >= 1) xm[q] = a.x[(long)(z0 - 1) * P + base + p];
            xc[q] = a.x[(long)z0 * P + base + p];
            if (z0 + 1 < nz) xp[q] = a.x[(long)(z0 + 1) * P + base + p];
        }
        if constexpr (kNeedsB) {
            if (v1[q]) bk[q] = a.b[(long)z0 * P + base + p];
        }
    }
    __syncthreads();
    double acc = 0.0;
    for (int k = z0; k < z1; ++k) {  // every thread of the workgroup runs the same z1 - z0 steps
        double xn[Q], bn[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int p = tid + kBoxBlock * q;
            xn[q] = 0.0;
            bn[q] = 0.0;
            if (k + 1 < z1) {
                if (v0[q] && k + 2 < nz) xn[q] = a.x[(long)(k + 2) * P + base + p];
                if constexpr (kNeedsB) {
                    if (v1[q]) bn[q] = a.b[(long)(k + 1) * P + base + p];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (v0[q]) X0[sidx[q]] = xc[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int p = tid + kBoxBlock * q;
            const int s = sidx[q];
            const long row = (long)k * P + base + p;
            double sum = 0.0;
            if (v1[q]) {
                sum = sum + c0 * xm[q];
                sum = sum + c1 * X0[s - pitch];
                sum = sum + c2 * X0[s - 1];
                sum = sum + c3 * xc[q];
                sum = sum + c4 * X0[s + 1];
                sum = sum + c5 * X0[s + pitch];
                sum = sum + c6 * xp[q];
            }
            if constexpr (EPI == BOX_SPMV_DOT) {
                if (v1[q]) {
                    a.y[row] = sum;
                    acc += xc[q] * sum;
                }
            } else if constexpr (EPI == BOX_JACOBI_DOT || EPI == BOX_JACOBI) {
                if (v1[q]) {
                    const double h = 1.0 * bk[q] + (-1.0) * sum;
                    const double xnew = xc[q] + div_const(om * h, dc3);
                    a.y[row] = xnew;
                    if constexpr (EPI == BOX_JACOBI_DOT) acc += xnew * bk[q];
                }
            } else if constexpr (EPI == BOX_JACOBI_PROLONG) {
                if (v1[q]) {
                    const double h = 1.0 * bk[q] + (-1.0) * sum;
                    const double xnew = xc[q] + div_const(om * h, dc3);
                    if (a.members) {
                        const i2v m = *reinterpret_cast<const i2v *>(a.members + 2 * row);
                        a.y2[m.x] = 1.0 * xnew + a.y2[m.x];
                        if (m.y >= 0) a.y2[m.y] = 1.0 * xnew + a.y2[m.y];
                    } else {
                        const long f = 2 * row;
                        if (f + 1 < a.nfine) {
                            d2v *pp = reinterpret_cast<d2v *>(a.y2 + f);
                            d2v v = *pp;
                            v.x = 1.0 * xnew + v.x;
                            v.y = 1.0 * xnew + v.y;
                            *pp = v;
                        } else {
                            a.y2[f] = 1.0 * xnew + a.y2[f];
                        }
                    }
                }
            } else {  // BOX_RESID_PAIRX: nx is even, so a line starts on an even row and lane parity = row parity
                const double ri = v1[q] ? 1.0 * bk[q] + (-1.0) * sum : 0.0;
                const double rn = lane_from_above(ri, 0.0);
                if (v1[q] && !(tid & 1)) {
                    const long J = row >> 1;
                    double bc = 0.0 + ri;
                    bc = bc + rn;
                    a.y[J] = bc;
                    if (a.y2) a.y2[J] = om * bc / (a.dc ? a.dc[J] : a.dconst);  // (nullptr: the coarse leg starts from b itself)
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            xm[q] = xc[q];
            xc[q] = xp[q];
            xp[q] = xn[q];
            bk[q] = bn[q];
        }
    }
    if constexpr (EPI == BOX_SPMV_DOT || EPI == BOX_JACOBI_DOT) {
        acc = wave_sum(acc);
        const int lane = tid & 63, w = tid >> 6;
        if (lane == 0) red[w] = acc;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < kBoxBlock / 64; ++q) t += red[q];
            a.partial[blockIdx.x] = t;
        }
    }
}

// Residual + restriction (+ the coarse level's zero-guess sweep) on a box-grid level whose aggregates pair a grid point with its
// neighbour one line (AXIS 1) or one plane (AXIS 2) up, coarse points numbered lexicographically: one thread per aggregate computes
// both residuals from the seven-point stencil (every access is a unit-stride run along the grid line), adds them in
// transfer_residual's order and writes b_c and x_c = omega b_c / d_c.  r is never stored, R is never read.  Missing neighbours
// enter as +0.0 (see sdia_box2_kernel: adding +-0.0 leaves the sum's bits alone).
__device__ __forceinline__ double box_residual(const BoxArgs &g, const double *__restrict__ x, const double *__restrict__ b, int f, int i, int j,
                                               int k)
{
    const int nx = g.nx, P = g.nx * g.ny;
    const double xm2 = k > 0 ? x[f - P] : 0.0;
    const double xm1 = j > 0 ? x[f - nx] : 0.0;
    const double xm0 = i > 0 ? x[f - 1] : 0.0;
    const double xc = x[f];
    const double xp0 = i < nx - 1 ? x[f + 1] : 0.0;
    const double xp1 = j < g.ny - 1 ? x[f + nx] : 0.0;
    const double xp2 = k < g.nz - 1 ? x[f + P] : 0.0;
    const double bi = b[f];
    double sum = 0.0;
    sum = sum + g.c[0] * xm2;
    sum = sum + g.c[1] * xm1;
    sum = sum + g.c[2] * xm0;
    sum = sum + g.c[3] * xc;
    sum = sum + g.c[4] * xp0;
    sum = sum + g.c[5] * xp1;
    sum = sum + g.c[6] * xp2;
    return 1.0 * bi + (-1.0) * sum;
}

template <int AXIS>
__global__ __launch_bounds__(kBlock) void box_resid_pair_kernel(BoxArgs g, int nc, int rev, const double *__restrict__ x, const double *__restrict__ b,
                                                                const double *__restrict__ dc, double dconst, double *__restrict__ bc,
                                                                double *__restrict__ xc)
{
    const int J = blockIdx.x * kBlock + threadIdx.x;
    if (J >= nc) return;
    const int nx = g.nx, ny = g.ny;
    const int cny = AXIS == 1 ? ny / 2 : ny;
    const int i = J % nx, t = J / nx;
    const int cj = t % cny, ck = t / cny;
    const int j = AXIS == 1 ? 2 * cj : cj, k = AXIS == 2 ? 2 * ck : ck;
    const int f1 = i + nx * (j + ny * k);
    const int f2 = f1 + (AXIS == 1 ? nx : nx * ny);
    const int Jout = rev ? nc - 1 - J : J;  // (the matching of some levels numbers its aggregates from the far end of the box)
    const double dj = !xc ? 1.0 : (dc ? dc[Jout] : dconst);
    const double r1 = box_residual(g, x, b, f1, i, j, k);
    const double r2 = box_residual(g, x, b, f2, i, j + (AXIS == 1 ? 1 : 0), k + (AXIS == 2 ? 1 : 0));
    double s = 0.0 + r1;
    s = s + r2;
    bc[Jout] = s;
    if (xc) xc[Jout] = g.omega * s / dj;  // (nullptr: the coarse leg starts from b itself)
}

// ------------------------------------------------------------------ fp32 preconditioner kernels
// Same sliced-diagonal structure with float values and float vectors: 4 B per stored entry and
// 12 B per row for a fused sweep.  Used only inside the (opt-in) fp32 V-cycle; the fp64 parity
// path never touches them.
template <int L>
__device__ __forceinline__ float sdia32_chunk(const int *__restrict__ off, const unsigned long long *__restrict__ mask,
                                              const float *__restrict__ vp, const float *__restrict__ x, int d0, int row, int lane,
                                              float &dv, float sum)
{
    float v[L], xv[L];
    bool on[L];
    int oo[L];
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const int o = __builtin_amdgcn_readfirstlane(off[d0 + u]);
        oo[u] = o;
        const unsigned long long m = mask[d0 + u];
        on[u] = (m >> lane) & 1ull;
        v[u] = vp[(size_t)(d0 + u) * 64];
        xv[u] = x[on[u] ? row + o : row];
    }
#pragma unroll
    for (int u = 0; u < L; ++u) {
        const float t = v[u] * xv[u];
        sum = on[u] ? sum + t : sum;
        if (oo[u] == 0) dv = on[u] ? v[u] : dv;
    }
    return sum;
}

template <int OP>
__global__ __launch_bounds__(kBlock) void sdia_f32_kernel(int nrow, int nslice, int ngroups, int remap, const int *__restrict__ sd_ptr,
                                                           const int *__restrict__ sd_off, const unsigned long long *__restrict__ sd_mask,
                                                           const float *__restrict__ sd_val, const float *__restrict__ x,
                                                           const float *__restrict__ b, float *__restrict__ y, float omega)
{
    const int gid = remap ? xcd_remap(blockIdx.x, ngroups, remap) : blockIdx.x;
    if (gid >= ngroups) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sl = gid * (kBlock / 64) + w;
    if (sl >= nslice) return;
    int row = sl * 64 + lane;
    const bool has_row = row < nrow;
    if (!has_row) row = nrow - 1;
    const int s0 = __builtin_amdgcn_readfirstlane(sd_ptr[sl]) & kSdPtrMask;
    const int nd = (__builtin_amdgcn_readfirstlane(sd_ptr[sl + 1]) & kSdPtrMask) - s0;
    const float bi = b[row];
    const float xi = (OP == OP_JACOBI) ? x[row] : 0.f;
    const int *off = sd_off + s0;
    const unsigned long long *mask = sd_mask + s0;
    const float *vp = sd_val + (size_t)s0 * 64 + lane;
    float sum = 0.f, dv = 0.f;
    int d = 0;
    for (; d + 8 <= nd; d += 8) sum = sdia32_chunk<8>(off, mask, vp, x, d, row, lane, dv, sum);
    switch (nd - d) {
    case 7: sum = sdia32_chunk<7>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 6: sum = sdia32_chunk<6>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 5: sum = sdia32_chunk<5>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 4: sum = sdia32_chunk<4>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 3: sum = sdia32_chunk<3>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 2: sum = sdia32_chunk<2>(off, mask, vp, x, d, row, lane, dv, sum); break;
    case 1: sum = sdia32_chunk<1>(off, mask, vp, x, d, row, lane, dv, sum); break;
    default: break;
    }
    if (!has_row) return;
    const float h = bi - sum;
    if constexpr (OP == OP_JACOBI)
        y[row] = xi + omega * h / dv;
    else
        y[row] = h;
}

__global__ __launch_bounds__(kBlock) void jacobi_zero_f32_kernel(int n, const float *__restrict__ b, const float *__restrict__ d,
                                                                  float omega, float *__restrict__ x)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) x[i] = omega * b[i] / d[i];
}

// short rows (restriction of an aggregation: 1-2 entries): one thread per row
__global__ __launch_bounds__(kBlock) void csr_rows_f32_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                               const double *__restrict__ val, const float *__restrict__ x,
                                                               float *__restrict__ y, int add)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        float s = 0.f;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) s += (float)val[j] * x[col[j]];
        y[i] = add ? s + y[i] : s;
    }
}

// levels without a sliced-diagonal mirror (unstructured operators, small coarse levels): one thread per
// CSR row over a float copy of the values (8 B per entry instead of 12)
template <int OP>
__global__ __launch_bounds__(kBlock) void csr_f32_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                          const float *__restrict__ val, const float *__restrict__ diag,
                                                          const float *__restrict__ x, const float *__restrict__ b, float *__restrict__ y,
                                                          float omega)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        float s = 0.f;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) s += val[j] * x[col[j]];
        const float h = b[i] - s;
        if constexpr (OP == OP_JACOBI)
            y[i] = x[i] + omega * h / diag[i];
        else
            y[i] = h;
    }
}

__global__ __launch_bounds__(kBlock) void prolong_agg_f32_kernel(int n, const int *__restrict__ agg, const float *__restrict__ xc,
                                                                  float *__restrict__ xf)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) xf[i] = xc[agg[i]] + xf[i];
}

__global__ __launch_bounds__(kBlock) void gemv_f32_kernel(int n, const float *__restrict__ M, const float *__restrict__ b,
                                                           float *__restrict__ x)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (row >= n) return;
    const float *__restrict__ m = M + (size_t)row * n;
    float acc = 0.f;
    for (int j = lane; j < n; j += 64) acc += m[j] * b[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) x[row] = acc;
}

__global__ __launch_bounds__(kBlock) void cvt_d2f_kernel(long n, const double *__restrict__ in, float *__restrict__ out)
{
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) out[i] = (float)in[i];
}

__global__ __launch_bounds__(kBlock) void cvt_f2d_dot_kernel(int n, const float *__restrict__ z32, const double *__restrict__ r,
                                                              double *__restrict__ z, double *__restrict__ partial)
{
    __shared__ double red[kBlock / 64];
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const double zi = (double)z32[i];
        z[i] = zi;
        acc += zi * r[i];
    }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

template <int OP, int TAG>
int launch_csr_tagged(const DevCsr &A, const CsrArgs &a, bool nt, int remap, hipStream_t st, const KernelConfig &c)
{
    const CsrFamily fam = csr_family(A, c);
    if (fam == FAM_SDIA_TAB) {
        const int T = sdia_tile_rows(A, c);
        if (T > 0 && !a.slice_list) {  // LDS-tiled variant: whole-level launches of grid stencils
            const int ntile = (A.nrow + T - 1) / T;
            const int C0 = A.sd_tab.nd / 2;
            const size_t lds = (size_t)(T - A.sd_tab.off[C0 - 2] + A.sd_tab.off[C0 + 2]) * sizeof(double);
            const int grid = remap_grid(ntile, remap);
#define SPARSH_LAUNCH_TILE(ND_, S_) \
    hipLaunchKernelGGL((sdia_tile_kernel<OP, ND_, S_, TAG>), dim3(grid), dim3(kTileBlock), lds, st, A.nrow, A.ncol, ntile, remap, A.sd_tab, A.sd_tmask, A.sd_tconf, A.sd_ptr, A.sd_off, A.sd_mask, A.sd_vidx, A.sd_cval, A.sd_val, A.sd_rec, a)
            const int S = T / kTileBlock;
            if (A.sd_tab.nd == 7) {
                if (S == 1) SPARSH_LAUNCH_TILE(7, 1);
                else if (S == 2) SPARSH_LAUNCH_TILE(7, 2);
                else SPARSH_LAUNCH_TILE(7, 4);
            } else {
                if (S == 1) SPARSH_LAUNCH_TILE(5, 1);
                else if (S == 2) SPARSH_LAUNCH_TILE(5, 2);
                else SPARSH_LAUNCH_TILE(5, 4);
            }
#undef SPARSH_LAUNCH_TILE
            return ntile;
        }
        const int nwork = a.slice_list ? a.nlist : A.nslice;
        const int ngroups = (nwork + 3) / 4;
        if (ngroups <= 0) return 0;
        const int grid = remap_grid(ngroups, remap);
        if (nt)
            hipLaunchKernelGGL((sdia_tab_kernel<OP, true, TAG>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.ncol, nwork, ngroups, remap, A.sd_tab, a.slice_list, A.sd_tmask, A.sd_tconf, A.sd_ptr, A.sd_off, A.sd_mask, A.sd_vidx, A.sd_cval, A.sd_val, A.sd_rec, a);
        else
            hipLaunchKernelGGL((sdia_tab_kernel<OP, false, TAG>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.ncol, nwork, ngroups, remap, A.sd_tab, a.slice_list, A.sd_tmask, A.sd_tconf, A.sd_ptr, A.sd_off, A.sd_mask, A.sd_vidx, A.sd_cval, A.sd_val, A.sd_rec, a);
        return ngroups;
    }
    if (fam == FAM_SDIA) {
        const int ngroups = ((a.slice_list ? a.nlist : A.nslice) + 3) / 4;
        if (ngroups <= 0) return 0;
        const int grid = remap_grid(ngroups, remap);
        if (nt)
            hipLaunchKernelGGL((sdia_kernel<OP, true, TAG>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.nslice, ngroups, remap, A.sd_ptr, A.sd_off, A.sd_mask, A.sd_vidx, A.sd_cval, A.sd_val, A.sd_rec, a);
        else
            hipLaunchKernelGGL((sdia_kernel<OP, false, TAG>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.nslice, ngroups, remap, A.sd_ptr, A.sd_off, A.sd_mask, A.sd_vidx, A.sd_cval, A.sd_val, A.sd_rec, a);
        return ngroups;
    }
    if (fam == FAM_SELL) {
        const int ngroups = ((a.slice_list ? a.nlist : A.nslice) + 3) / 4;
        if (ngroups <= 0) return 0;
        const int grid = remap_grid(ngroups, remap);
        if (nt)
            hipLaunchKernelGGL((sell_kernel<OP, true, TAG>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.nslice, ngroups, remap, A.slice_ptr, A.rowptr, A.sell_col, A.sell_val, a);
        else
            hipLaunchKernelGGL((sell_kernel<OP, false, TAG>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.nslice, ngroups, remap, A.slice_ptr, A.rowptr, A.sell_col, A.sell_val, a);
        return ngroups;
    }
    if (fam == FAM_CSR_WAVE) {
        const int ngroups = (A.nwblk + 3) / 4;
        if (ngroups <= 0) return 0;
        const int grid = remap_grid(ngroups, remap);
#define SPARSH_LAUNCH_WAVE(NT_, VEC_) \
    hipLaunchKernelGGL((csr_wave_kernel<OP, NT_, VEC_, TAG>), dim3(grid), dim3(kBlock), 0, st, A.waveblk, A.nwblk, ngroups, remap, A.rowptr, A.col, A.val, a)
        if (nt && c.vec) SPARSH_LAUNCH_WAVE(true, 1);
        else if (nt) SPARSH_LAUNCH_WAVE(true, 0);
        else if (c.vec) SPARSH_LAUNCH_WAVE(false, 1);
        else SPARSH_LAUNCH_WAVE(false, 0);
#undef SPARSH_LAUNCH_WAVE
        return ngroups;
    }
    if (A.nblk <= 0) return 0;
    const int grid = remap_grid(A.nblk, remap);
    if (fam == FAM_CSR_ROWLANE16) {  // row-lane order, 16-bit delta-coded column indices
        if (nt)
            hipLaunchKernelGGL((csr_rowlane16_kernel<OP, true, TAG>), dim3(grid), dim3(kBlock), 0, st, A.rowblk, A.nblk, remap, A.rowptr, A.col, A.col16, A.cbase, A.val, a);
        else
            hipLaunchKernelGGL((csr_rowlane16_kernel<OP, false, TAG>), dim3(grid), dim3(kBlock), 0, st, A.rowblk, A.nblk, remap, A.rowptr, A.col, A.col16, A.cbase, A.val, a);
        return A.nblk;
    }
    if (fam == FAM_CSR_ROWLANE) {  // gathers in row-lane order (col/val staged in LDS)
        if (nt)
            hipLaunchKernelGGL((csr_rowlane_kernel<OP, true, TAG>), dim3(grid), dim3(kBlock), 0, st, A.rowblk, A.nblk, remap, A.rowptr, A.col, A.val, a);
        else
            hipLaunchKernelGGL((csr_rowlane_kernel<OP, false, TAG>), dim3(grid), dim3(kBlock), 0, st, A.rowblk, A.nblk, remap, A.rowptr, A.col, A.val, a);
        return A.nblk;
    }
#define SPARSH_LAUNCH_BLOCK(NT_, VEC_) \
    hipLaunchKernelGGL((csr_block_kernel<OP, NT_, VEC_, TAG>), dim3(grid), dim3(kBlock), 0, st, A.rowblk, A.nblk, remap, A.rowptr, A.col, A.val, a)
    if (nt && c.vec) SPARSH_LAUNCH_BLOCK(true, 1);
    else if (nt) SPARSH_LAUNCH_BLOCK(true, 0);
    else if (c.vec) SPARSH_LAUNCH_BLOCK(false, 1);
    else SPARSH_LAUNCH_BLOCK(false, 0);  // (vec = 3 below the HBM threshold runs the paired-load kernel)
#undef SPARSH_LAUNCH_BLOCK
    return A.nblk;
}

}  // namespace

namespace {
size_t box2_lds_bytes(int nx, int TY) { return (size_t)2 * ((size_t)(TY + 4) * (nx + 1) + 1) * sizeof(double); }
}  // namespace

// Plan of the double sweep: Q points per thread, TY lines per tile, CZ planes per chunk.  Cost model (checked against
// tools/micro/box2_proto on MI355X: 216^3 Q4/TY14/CZ14 57 us, Q3/TY10/CZ14 94 us, 108x216x216 Q4/TY33/CZ6 34 us, Q2/TY14/CZ14 38 us):
// a workgroup's time ~ (CZ + 2) steps x Q points, the launch takes ceil(workgroups / 256 CUs) rounds of it.  The Q = 2 instance needs
// 62 VGPRs and 31 KB of LDS, so two of its workgroups share a CU: 512 slots, each running at ~1/1.6 of the speed it has alone
// (108 x 216 x 216: Q2/TY14/CZ7 = 496 workgroups 32.9 us against 35.8 for CZ14 = 256 and 35.4 for Q3/TY24/CZ8 = 243).
bool box2_plan(DevCsr &A)
{
    A.box_q = A.box_ty = A.box_cz = 0;
    const int nx = A.box_nx, ny = A.box_ny, nz = A.box_nz;
    if (nx < 2 || ny < 1 || nz < 1) return false;
    long best = -1;
    for (int Q = 2; Q <= 4; ++Q) {
        int TY = std::min(ny, Q * kBoxBlock / nx - 4);
        while (TY >= 1 && box2_lds_bytes(nx, TY) > 65536) --TY;
        if (TY < 1) continue;
        const int ytiles = (ny + TY - 1) / TY;
        for (int zch = 1; zch <= nz; ++zch) {
            const int CZ = (nz + zch - 1) / zch;
            const int chunks = (nz + CZ - 1) / CZ;
            const long wgs = (long)ytiles * chunks;
            long cost;  // in tenths of a step of one point
            if (Q == 2 && wgs > 256) cost = ((wgs + 511) / 512) * (CZ + 2) * Q * 16;
            else cost = ((wgs + 255) / 256) * (CZ + 2) * Q * 10;
            if (best < 0 || cost < best) {
                best = cost;
                A.box_q = Q;
                A.box_ty = TY;
                A.box_cz = CZ;
            }
        }
    }
    return A.box_q > 0;
}

// plan of the single-stage kernel: region = TY + 2 lines, one LDS plane, nothing recomputed; a workgroup's time ~ (CZ + 1) steps x Q points.
// shared_cu: count 512 slots for the instances of <= 3 points per thread (<= 64 VGPRs, <= 32 KB of LDS: two workgroups per CU, each at
// ~1/1.6 speed) -- the alternative plan the setup times against the one-workgroup-per-CU plan (Engine::tune_box_kernels)
bool box1_plan(DevCsr &A, bool shared_cu)
{
    A.box1_q = A.box1_ty = A.box1_cz = 0;
    const int nx = A.box_nx, ny = A.box_ny, nz = A.box_nz;
    if (nx < 2 || ny < 1 || nz < 1) return false;
    long best = -1;
    for (int Q = 2; Q <= 4; ++Q) {
        int TY = std::min(ny, Q * kBoxBlock / nx - 2);
        while (TY >= 1 && ((size_t)(TY + 2) * (nx + 1) + 1) * sizeof(double) > 65536) --TY;
        if (TY < 1) continue;
        const int ytiles = (ny + TY - 1) / TY;
        for (int zch = 1; zch <= nz; ++zch) {
            const int CZ = (nz + zch - 1) / zch;
            const int chunks = (nz + CZ - 1) / CZ;
            const long wgs = (long)ytiles * chunks;
            long cost;
            if (shared_cu && Q <= 3 && wgs > 256) cost = ((wgs + 511) / 512) * (CZ + 1) * Q * 16;
            else cost = ((wgs + 255) / 256) * (CZ + 1) * Q * 10;
            if (best < 0 || cost < best) {
                best = cost;
                A.box1_q = Q;
                A.box1_ty = TY;
                A.box1_cz = CZ;
            }
        }
    }
    return A.box1_q > 0;
}

bool box1_applies(const DevCsr &A, const KernelConfig &c)
{
    return c.box1 != 0 && A.box1_on && A.box1_q > 0 && csr_family(A, c) == FAM_SDIA_TAB;
}

int launch_box1(const DevCsr &A, int epi, const CsrArgs &a, bool finest, hipStream_t st)
{
    BoxArgs g;
    g.nx = A.box_nx;
    g.ny = A.box_ny;
    g.nz = A.box_nz;
    g.TY = A.box1_ty;
    g.CZ = A.box1_cz;
    g.ytiles = (g.ny + g.TY - 1) / g.TY;
    for (int u = 0; u < 7; ++u) g.c[u] = A.sd_tab.cval[u];
    g.omega = a.omega;
    Box1Args b;
    b.x = a.x;
    b.b = a.b;
    b.y = a.y;
    b.y2 = a.y2;
    b.dc = a.d;
    b.dconst = a.dconst;
    b.members = a.members;
    b.nfine = a.nfine;
    b.partial = a.partial ? a.partial + a.partial_off : nullptr;
    const int chunks = (g.nz + g.CZ - 1) / g.CZ;
    const int nwg = g.ytiles * chunks;
    const dim3 grid(nwg), block(kBoxBlock);
    const size_t lds = ((size_t)(g.TY + 2) * (g.nx + 1) + 1) * sizeof(double);
#define SPARSH_LAUNCH_BOX1(Q_, E_)                                                                  \
    do {                                                                                            \
        if (finest) hipLaunchKernelGGL((sdia_box1_kernel<Q_, E_, 1>), grid, block, lds, st, g, b);  \
        else hipLaunchKernelGGL((sdia_box1_kernel<Q_, E_, 0>), grid, block, lds, st, g, b);         \
    } while (0)
#define SPARSH_LAUNCH_BOX1_Q(E_)                      \
    do {                                              \
        if (A.box1_q == 4) SPARSH_LAUNCH_BOX1(4, E_); \
        else if (A.box1_q == 3) SPARSH_LAUNCH_BOX1(3, E_); \
        else SPARSH_LAUNCH_BOX1(2, E_);               \
    } while (0)
    switch (epi) {
    case BOX_SPMV_DOT: SPARSH_LAUNCH_BOX1_Q(BOX_SPMV_DOT); break;
    case BOX_JACOBI_DOT: SPARSH_LAUNCH_BOX1_Q(BOX_JACOBI_DOT); break;
    case BOX_RESID_PAIRX: SPARSH_LAUNCH_BOX1_Q(BOX_RESID_PAIRX); break;
    case BOX_JACOBI: SPARSH_LAUNCH_BOX1_Q(BOX_JACOBI); break;
    default: SPARSH_LAUNCH_BOX1_Q(BOX_JACOBI_PROLONG); break;
    }
#undef SPARSH_LAUNCH_BOX1_Q
#undef SPARSH_LAUNCH_BOX1
    return nwg;
}

bool box2_applies(const DevCsr &A, const KernelConfig &c)
{
    return c.box2 != 0 && A.box_on && A.box_q > 0 && csr_family(A, c) == FAM_SDIA_TAB;
}

void launch_box2(const DevCsr &A, const double *x, const double *b, double *y, double omega, bool finest, hipStream_t st, bool from_zero)
{
    BoxArgs g;
    g.nx = A.box_nx;
    g.ny = A.box_ny;
    g.nz = A.box_nz;
    g.TY = A.box_ty;
    g.CZ = A.box_cz;
    g.ytiles = (g.ny + g.TY - 1) / g.TY;
    for (int u = 0; u < 7; ++u) g.c[u] = A.sd_tab.cval[u];
    g.omega = omega;
    const int chunks = (g.nz + g.CZ - 1) / g.CZ;
    const dim3 grid(g.ytiles * chunks), block(kBoxBlock);
    const size_t lds = box2_lds_bytes(g.nx, g.TY);
#define SPARSH_LAUNCH_BOX(Q_)                                                                                       \
    do {                                                                                                            \
        if (from_zero) {                                                                                            \
            if (finest) hipLaunchKernelGGL((sdia_box2_kernel<Q_, 1, true>), grid, block, lds, st, g, x, b, y);      \
            else hipLaunchKernelGGL((sdia_box2_kernel<Q_, 0, true>), grid, block, lds, st, g, x, b, y);             \
        } else {                                                                                                    \
            if (finest) hipLaunchKernelGGL((sdia_box2_kernel<Q_, 1, false>), grid, block, lds, st, g, x, b, y);     \
            else hipLaunchKernelGGL((sdia_box2_kernel<Q_, 0, false>), grid, block, lds, st, g, x, b, y);            \
        }                                                                                                           \
    } while (0)
    if (A.box_q == 4) SPARSH_LAUNCH_BOX(4);
    else if (A.box_q == 3) SPARSH_LAUNCH_BOX(3);
    else SPARSH_LAUNCH_BOX(2);
#undef SPARSH_LAUNCH_BOX
}

void launch_box_resid_pair(const DevCsr &A, int axis, const double *x, const double *b, const double *dc, double dconst, double omega,
                           double *bc, double *xc, hipStream_t st)
{
    BoxArgs g;
    g.nx = A.box_nx;
    g.ny = A.box_ny;
    g.nz = A.box_nz;
    g.TY = g.CZ = g.ytiles = 0;
    for (int u = 0; u < 7; ++u) g.c[u] = A.sd_tab.cval[u];
    g.omega = omega;
    const int nc = A.nrow / 2;
    if (nc <= 0) return;
    const dim3 grid((nc + kBlock - 1) / kBlock), block(kBlock);
    const int rev = axis < 0 ? 1 : 0;
    if (axis == 1 || axis == -1)
        hipLaunchKernelGGL(box_resid_pair_kernel<1>, grid, block, 0, st, g, nc, rev, x, b, dc, dconst, bc, xc);
    else
        hipLaunchKernelGGL(box_resid_pair_kernel<2>, grid, block, 0, st, g, nc, rev, x, b, dc, dconst, bc, xc);
}

bool resid_pair_applies(const DevCsr &A, const KernelConfig &c)
{
    return c.pair_restrict && csr_family(A, c) == FAM_SDIA_TAB && sdia_tile_rows(A, c) == 0 && A.nslice > 0;
}

void launch_resid_pair(const DevCsr &A, const CsrArgs &a, bool finest, hipStream_t st, const KernelConfig &c)
{
    bool nt;
    int remap;
    csr_placement(A, c, &nt, &remap);
    const int ngroups = (A.nslice + 3) / 4;
    const int grid = remap_grid(ngroups, remap);
    if (finest)
        hipLaunchKernelGGL((sdia_tab_kernel<OP_RESID_PAIR, false, 1>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.ncol, A.nslice, ngroups, remap, A.sd_tab, nullptr, A.sd_tmask, A.sd_tconf, A.sd_ptr, A.sd_off, A.sd_mask, A.sd_vidx, A.sd_cval, A.sd_val, A.sd_rec, a);
    else
        hipLaunchKernelGGL((sdia_tab_kernel<OP_RESID_PAIR, false, 0>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.ncol, A.nslice, ngroups, remap, A.sd_tab, nullptr, A.sd_tmask, A.sd_tconf, A.sd_ptr, A.sd_off, A.sd_mask, A.sd_vidx, A.sd_cval, A.sd_val, A.sd_rec, a);
}

CsrFamily csr_family(const DevCsr &A, const KernelConfig &c)
{
    if (c.kind == 3 && A.has_sdia() && c.table && A.sd_tmask) return FAM_SDIA_TAB;
    // a small level without a stencil table is latency-bound by the slot-header chain of sdia_kernel
    // (9842-row level of the 216^3 hierarchy: 5.5 us per sweep against 3.1 us for the sliced-ELL kernel)
    const bool small_prefers_ell = A.sell_val && A.nrow < 65536;
    if (c.kind == 3 && A.has_sdia() && !small_prefers_ell) return FAM_SDIA;
    if (c.kind >= 2 && A.sell_val) return FAM_SELL;
    if (c.kind == 1 && A.waveblk) return FAM_CSR_WAVE;
    // workgroup CSR-stream kernels: gathers in row-lane order when asked for, or (vec = 3) when the operator streams from HBM
    const size_t csr_bytes = (size_t)A.nnz * 12 + (size_t)A.nrow * 36;
    const bool streams = csr_bytes > (240u << 20);
    if ((c.vec == 4 || (c.vec == 3 && streams)) && A.col16) return FAM_CSR_ROWLANE16;  // 10 instead of 12 bytes per entry
    if (c.vec == 2 || c.vec == 4 || (c.vec == 3 && streams)) return FAM_CSR_ROWLANE;
    return FAM_CSR_BLOCK;
}

// Alternating sweep direction pays where a sweep streams several times what the 256 MB memory-side cache holds: the next sweep
// then starts on what is still cached instead of on what was evicted first.  Same-process A/B at 216^3 (tools/alt_dir_ab.py):
// CSR-stream level 0 (1.1 GB per sweep) 189 -> 178 us, general sliced diagonals (0.83 GB) 139.5 -> 135.8 us; the 5 M-row
// level 1 of both (0.4-0.56 GB) 1 % slower; the value-free table path (0.25 GB: fits) +-1 % depending on the box.  Hence the
// threshold.  Bytes per sweep of the launched layout: vectors + the value / index streams.
bool csr_alternates(const DevCsr &A, const KernelConfig &c)
{
    if (c.alt_dir == 0) return false;
    if (c.alt_dir >= 2) return true;
    size_t bytes = (size_t)A.nrow * 24;
    switch (csr_family(A, c)) {
    case FAM_SDIA_TAB: break;
    case FAM_SDIA: bytes += (size_t)A.sd_vblocks * 64 * 8; break;
    case FAM_SELL: bytes += (size_t)A.sell_entries * 12; break;
    case FAM_CSR_ROWLANE16: bytes += (size_t)A.nnz * 10; break;
    default: bytes += (size_t)A.nnz * 12; break;
    }
    return bytes > ((size_t)640 << 20);
}

// rows per workgroup of the LDS-tiled table kernel for this operator, 0 when it does not apply: grid
// stencils in lexicographic order (offsets -1, 0, +1 adjacent in the middle of a 5- or 7-entry table) whose
// +-line reach is small enough for the window [r0 - line, r0 + T + line) to fit 64 KiB of LDS with at most
// ~50 % of halo
int sdia_tile_rows(const DevCsr &A, const KernelConfig &c)
{
    if (!c.tile || !A.sd_tmask || (A.sd_tab.near != 73 && A.sd_tab.near != 52)) return 0;
    const int C0 = A.sd_tab.nd / 2;
    const int lo = -A.sd_tab.off[C0 - 2], hi = A.sd_tab.off[C0 + 2];
    if (lo < 2 || hi < 2) return 0;
    if (A.sd_tab.nd == 7 && (-A.sd_tab.off[0] <= lo || A.sd_tab.off[6] <= hi)) return 0;
    if (A.nrow < 65536) return 0;  // small levels are launch-latency bound either way: keep one slice per wave
    const int reach = lo + hi;
    const int T = reach <= 512 ? 1024 : (reach <= 1024 ? 2048 : (reach <= 2048 ? 4096 : 0));
    return T;
}

const char *csr_family_name(CsrFamily f)
{
    switch (f) {
    case FAM_SDIA_TAB: return "sdia_tab_kernel";
    case FAM_SDIA: return "sdia_kernel";
    case FAM_SELL: return "sell_kernel";
    case FAM_CSR_WAVE: return "csr_wave_kernel";
    case FAM_CSR_ROWLANE: return "csr_rowlane_kernel";
    case FAM_CSR_ROWLANE16: return "csr_rowlane16_kernel";
    default: return "csr_block_kernel";
    }
}

void csr_placement(const DevCsr &A, const KernelConfig &c, bool *nt_out, int *remap_out)
{
    bool nt = c.nt;
    int remap = c.remap;
    const CsrFamily fam = csr_family(A, c);
    if (c.auto_policy) {
        // measured on MI355X (profiles/r01_remap_sweep.txt, r01_remap_sweep_sdia.txt): what decides is
        // whether one sweep's working set -- the bytes of the layout actually used plus the
        // vectors -- stays in the 256 MiB Infinity Cache between consecutive sweeps.  If it does:
        // default cache policy, one contiguous eighth of the rows per XCD.  If not: non-temporal
        // matrix stream and all XCDs sweeping one neighbourhood (groups of 16 row blocks).
        size_t bytes;
        if (fam == FAM_SDIA || fam == FAM_SDIA_TAB)
            bytes = (size_t)A.sd_vblocks * 64 * 8 + (size_t)A.nrow * 24 +
                    (A.sd_tmask ? (size_t)A.nslice * 68 : (A.sd_rec ? (size_t)A.nslice * kSdRecInts * 4 : (size_t)A.sd_slots * 24));
        else if (fam == FAM_SELL)
            bytes = (size_t)A.sell_entries * 12 + (size_t)A.nrow * 28;
        else
            bytes = (size_t)A.nnz * 12 + (size_t)A.nrow * 36;
        if (fam == FAM_SDIA_TAB) {
            // table path: no matrix stream to keep out of the caches; a contiguous eighth per XCD
            // reads x 1.2x instead of 3.3x and wins inside the solve at every size (216^3: 395 vs
            // 380 it/s, finest-level sweep 58 vs 65 us; profiles/r01_table_placement_in_solve.txt)
            nt = false;
            remap = 1;
        } else if (bytes > (240u << 20)) {
            nt = true;
            remap = 16;
        } else {
            nt = false;
            remap = 1;
        }
    }
    *nt_out = nt;
    *remap_out = remap;
}

namespace {

template <int OP>
int launch_csr_op(const DevCsr &A, const CsrArgs &a, bool finest, hipStream_t st, const KernelConfig &c)
{
    bool nt;
    int remap;
    csr_placement(A, c, &nt, &remap);
    return finest ? launch_csr_tagged<OP, 1>(A, a, nt, remap, st, c) : launch_csr_tagged<OP, 0>(A, a, nt, remap, st, c);
}

// ------------------------------------------------------------------ elementwise

constexpr int kEwGridMax = 2048;  // grid-stride: ~8 workgroups per CU

inline int ew_grid(int n)
{
    int g = (n + kBlock * 2 - 1) / (kBlock * 2);
    if (g < 1) g = 1;
    return g > kEwGridMax ? kEwGridMax : g;
}

__global__ __launch_bounds__(kBlock) void jacobi_zero_kernel(int n, const double *__restrict__ b, const double *__restrict__ d,
                                                              double dconst, double omega, double *__restrict__ x)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        // x = 0 + omega*(b - 0)/d ; (b - 0) and (0 + t) are exact.  d == nullptr: constant diagonal dconst
        x[i] = omega * b[i] / (d ? d[i] : dconst);
    }
}

__global__ __launch_bounds__(kBlock) void prolong_agg_kernel(int n, const int *__restrict__ agg, const double *__restrict__ xc,
                                                              double *__restrict__ xf)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) xf[i] = 1.0 * xc[agg[i]] + xf[i];
}

// b_c[J] = sum of r over the members of aggregate J, in stored (ascending fine row) order: the
// restriction R = P^T of an aggregation P, whose values are all 1.0 -- the 8-byte value stream of the
// CSR kernel is skipped, 1.0 * r is r exactly
__global__ __launch_bounds__(kBlock) void restrict_agg_kernel(int nc, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                               const double *__restrict__ r, double *__restrict__ bc)
{
    for (int J = blockIdx.x * kBlock + threadIdx.x; J < nc; J += gridDim.x * kBlock) {
        const int j0 = rowptr[J], j1 = rowptr[J + 1];
        double sum = 0.0;
        for (int j = j0; j < j1; ++j) sum = sum + r[col[j]];
        bc[J] = sum;
    }
}

// the same restriction, fused with the zero-guess sweep of the coarse level it feeds: x_c = omega*b_c/d_c is what
// jacobi_zero_kernel would compute from b_c in a launch of its own (same expression, bitwise the same values)
__global__ __launch_bounds__(kBlock) void restrict_agg_zero_kernel(int nc, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                                    const double *__restrict__ r, double *__restrict__ bc,
                                                                    const double *__restrict__ dc, double dconst, double omega,
                                                                    double *__restrict__ xc)
{
    for (int J = blockIdx.x * kBlock + threadIdx.x; J < nc; J += gridDim.x * kBlock) {
        const int j0 = rowptr[J], j1 = rowptr[J + 1];
        double sum = 0.0;
        for (int j = j0; j < j1; ++j) sum = sum + r[col[j]];
        bc[J] = sum;
        xc[J] = omega * sum / (dc ? dc[J] : dconst);  // dc == nullptr: the coarse level's diagonal is the constant dconst
    }
}

__global__ __launch_bounds__(kBlock) void pack_kernel(int n, const int *__restrict__ idx, const double *__restrict__ vec,
                                                       double *__restrict__ out)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) out[i] = vec[idx[i]];
}

// vec[pos[k]] = buf[k]: scatter a received staging buffer to the ghost positions (deep-halo exchange)
__global__ __launch_bounds__(kBlock) void unpack_kernel(int n, const int *__restrict__ pos, const double *__restrict__ buf,
                                                         double *__restrict__ vec)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) vec[pos[i]] = buf[i];
}

// 4-byte-per-lane streaming copy: calibrates rocprof's FETCH_SIZE for int32 index streams
__global__ __launch_bounds__(kBlock) void copy_int_kernel(int n, const int *__restrict__ x, int *__restrict__ y)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) y[i] = x[i];
}

__global__ __launch_bounds__(kBlock) void fill_kernel(int n, double v, double *__restrict__ x)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) x[i] = v;
}

__global__ __launch_bounds__(kBlock) void copy_kernel(int n, const double *__restrict__ x, double *__restrict__ y)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) y[i] = x[i];
}

__global__ __launch_bounds__(kBlock) void axpby_kernel(int n, double a, const double *__restrict__ x, double b, double *__restrict__ y)
{
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) y[i] = a * x[i] + b * y[i];
}

__global__ __launch_bounds__(kBlock) void dot_kernel(int n, const double *__restrict__ x, const double *__restrict__ y,
                                                      double *__restrict__ partial)
{
    __shared__ double red[kBlock / 64];
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) acc += x[i] * y[i];
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(kBlock) void dot2_kernel(int n, const double *__restrict__ a, const double *__restrict__ b,
                                                       const double *__restrict__ c, const double *__restrict__ d,
                                                       double *__restrict__ p0, double *__restrict__ p1)
{
    __shared__ double red[kBlock / 64];
    double s0 = 0.0, s1 = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        s0 += a[i] * b[i];
        s1 += c[i] * d[i];
    }
    const double t0 = block_sum(s0, red);
    __syncthreads();
    const double t1 = block_sum(s1, red);
    if (threadIdx.x == 0) {
        p0[blockIdx.x] = t0;
        p1[blockIdx.x] = t1;
    }
}

__global__ __launch_bounds__(kBlock) void cg_update_kernel(int n, const double *__restrict__ scal, const double *__restrict__ p,
                                                            const double *__restrict__ Ap, double *__restrict__ x,
                                                            double *__restrict__ r, double *__restrict__ partial)
{
    __shared__ double red[kBlock / 64];
    const double alpha = scal[S_ALPHA], nalpha = scal[S_NALPHA];
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        if (x) x[i] = x[i] + alpha * p[i];        // cblas_daxpy(alpha, p, x)   (x == nullptr: left to xp_update_kernel)
        const double ri = r[i] + nalpha * Ap[i];  // cblas_daxpy(-alpha, Ap, r)
        r[i] = ri;
        acc += ri * ri;
    }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// cg_update + the zero-guess sweep of the V-cycle that follows on the new residual: z0 = omega * r / d (what jacobi_zero_kernel
// computes from r one launch later; same expression, so the same bits) -- r is not read a second time
template <bool NT>
__global__ __launch_bounds__(kBlock) void cg_update_zero_kernel(int n, const double *__restrict__ scal, const double *__restrict__ p,
                                                                 const double *__restrict__ Ap, double *__restrict__ x, double *__restrict__ r,
                                                                 double *__restrict__ partial, const double *__restrict__ d, double dconst,
                                                                 double omega, double *__restrict__ z0)
{
    // d == nullptr: every row of the level has the diagonal entry dconst (constant-coefficient stencils) -- one stream less
    __shared__ double red[kBlock / 64];
    const double alpha = scal[S_ALPHA], nalpha = scal[S_NALPHA];
    double acc = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        // NT: x, p, Ap and d are not touched again before the cycle is over -- streamed past the caches, so that r and z0, which the
        // first sweep of the cycle reads next, are what stays resident
        const double api = NT ? __builtin_nontemporal_load(Ap + i) : Ap[i];
        const double di = !d ? dconst : (NT ? __builtin_nontemporal_load(d + i) : d[i]);
        if (x) {  // (x == nullptr: left to xp_update_kernel)
            const double xi = NT ? __builtin_nontemporal_load(x + i) : x[i];
            const double pi = NT ? __builtin_nontemporal_load(p + i) : p[i];
            const double xn = xi + alpha * pi;
            if (NT) __builtin_nontemporal_store(xn, x + i);
            else x[i] = xn;
        }
        const double ri = r[i] + nalpha * api;
        r[i] = ri;
        z0[i] = omega * ri / di;
        acc += ri * ri;
    }
    const double t = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(kBlock) void p_update_kernel(int n, const double *__restrict__ scal, const double *__restrict__ z,
                                                           double *__restrict__ p)
{
    const double beta = scal[S_BETA];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) p[i] = 1.0 * z[i] + beta * p[i];
}

// PCG: x += alpha p moved from the residual update to the direction update at the end of the same iteration -- nothing reads x in
// between, alpha is still in its slot, and p is read once for both: x_i + alpha p_i and 1.0 z_i + beta p_i as before (one n-vector
// stream less per iteration)
__global__ __launch_bounds__(kBlock) void xp_update_kernel(int n, const double *__restrict__ scal, const double *__restrict__ z,
                                                            double *__restrict__ p, double *__restrict__ x)
{
    const double alpha = scal[S_ALPHA], beta = scal[S_BETA];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const double pi = p[i];
        const double xi = __builtin_nontemporal_load(x + i);
        __builtin_nontemporal_store(xi + alpha * pi, x + i);
        p[i] = 1.0 * z[i] + beta * pi;
    }
}

__global__ __launch_bounds__(kBlock) void bicg_s_kernel(int n, const double *__restrict__ scal, const double *__restrict__ r,
                                                         const double *__restrict__ Ap, double *__restrict__ s)
{
    const double alpha = scal[S_ALPHA];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) s[i] = r[i] - alpha * Ap[i];
}

__global__ __launch_bounds__(kBlock) void bicg_xr_kernel(int n, const double *__restrict__ scal, const double *__restrict__ p1,
                                                          const double *__restrict__ s1, const double *__restrict__ s,
                                                          const double *__restrict__ As, const double *__restrict__ r0,
                                                          double *__restrict__ x, double *__restrict__ r, double *__restrict__ q0,
                                                          double *__restrict__ q1)
{
    __shared__ double red[kBlock / 64];
    const double alpha = scal[S_ALPHA], omega1 = scal[S_OMEGA1];
    double a0 = 0.0, a1 = 0.0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        x[i] = x[i] + alpha * p1[i] + omega1 * s1[i];
        const double ri = s[i] - omega1 * As[i];
        r[i] = ri;
        a0 += ri * r0[i];
        a1 += ri * ri;
    }
    const double t0 = block_sum(a0, red);
    __syncthreads();
    const double t1 = block_sum(a1, red);
    if (threadIdx.x == 0) {
        q0[blockIdx.x] = t0;
        q1[blockIdx.x] = t1;
    }
}

__global__ __launch_bounds__(kBlock) void bicg_p_kernel(int n, const double *__restrict__ scal, const double *__restrict__ r,
                                                         const double *__restrict__ Ap, double *__restrict__ p)
{
    const double beta = scal[S_BETA], omega1 = scal[S_OMEGA1];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) p[i] = r[i] + beta * (p[i] - omega1 * Ap[i]);
}

// ------------------------------------------------------------------ coarse GEMV
// x = M b, M row-major n x n (explicit inverse of the coarsest operator): one wave per row,
// lanes stride the row with 16-byte loads; b (<= 64 KiB) is served by L1/L2.
__global__ __launch_bounds__(kBlock) void gemv_kernel(int n, const double *__restrict__ M, const double *__restrict__ b,
                                                       double *__restrict__ x)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (row >= n) return;  // whole wave leaves
    const double *__restrict__ m = M + (size_t)row * n;
    double acc = 0.0;
    const int n2 = n & ~1;
    if ((((size_t)row * n) & 1) == 0) {  // row start 16-byte aligned
        for (int j = lane * 2; j < n2; j += 128) {
            const double2 mv = *reinterpret_cast<const double2 *>(m + j);
            const double2 bv = *reinterpret_cast<const double2 *>(b + j);
            acc += mv.x * bv.x;
            acc += mv.y * bv.y;
        }
    } else {
        for (int j = lane * 2; j < n2; j += 128) {
            acc += m[j] * b[j];
            acc += m[j + 1] * b[j + 1];
        }
    }
    if (lane == 0 && n2 < n) acc += m[n2] * b[n2];
    acc = wave_sum(acc);
    if (lane == 0) x[row] = acc;
}

// ------------------------------------------------------------------ finalize
// One workgroup adds the per-workgroup partials in a fixed order and updates the scalar slots.
constexpr int kFinBlock = 1024;

template <int U>
__device__ __forceinline__ double fin_strided_sum(const double *__restrict__ p, int n, int tid)
{
    double a = 0.0;
    for (int base = tid; base < n; base += U * kFinBlock) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * kFinBlock;
            v[u] = i < n ? p[i] : 0.0;  // (a + 0.0 == a)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) a += v[u];
    }
    return a;
}

__global__ __launch_bounds__(kFinBlock) void finalize_kernel(int code, int mode, const double *__restrict__ p0,
                                                              const double *__restrict__ p1, int nblk, int nblk1, double *__restrict__ scal,
                                                              int slot_a, double *__restrict__ hist, int it, int *__restrict__ iter_ctr,
                                                              int hist_cap)
{
    __shared__ double red0[kFinBlock / 64], red1[kFinBlock / 64];
    double s0 = 0.0, s1 = 0.0;
    if (mode != 2) {
        // same order of additions as a plain strided loop; a batch of 20 loads per thread is issued before the first addition
        // (one workgroup pulling 39 366 partials of a 216^3 level: two rounds of memory latency instead of five)
        double a0 = fin_strided_sum<20>(p0, nblk, threadIdx.x);
        double a1 = p1 ? fin_strided_sum<20>(p1, nblk1, threadIdx.x) : 0.0;  // (the second array may come from a different kernel: its own count)
        a0 = wave_sum(a0);
        a1 = wave_sum(a1);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) {
            red0[w] = a0;
            red1[w] = a1;
        }
        __syncthreads();
        if (threadIdx.x != 0) return;
        for (int k = 0; k < kFinBlock / 64; ++k) {
            s0 += red0[k];
            s1 += red1[k];
        }
        if (mode == 1) {  // multi-GPU: leave the local sums for the all-reduce
            scal[S_SUM0] = s0;
            scal[S_SUM1] = s1;
            return;
        }
    } else {
        if (threadIdx.x != 0) return;
        s0 = scal[S_SUM0];
        s1 = scal[S_SUM1];
    }
    // residual-history slot: explicit (eager launches) or a device-side counter (graph replays,
    // where kernel arguments are frozen)
    int hslot = it;
    if (hist && it < 0) hslot = (*iter_ctr)++;
    if (hslot >= hist_cap) hslot = hist_cap - 1;
    switch (code) {
    case FIN_STORE: scal[slot_a] = s0; break;
    case FIN_SQRT: {
        const double v = sqrt(s0);
        scal[slot_a] = v;
        if (hist) hist[hslot] = v;
    } break;
    case FIN_PCG_ALPHA: {
        scal[S_PAP] = s0;
        const double alpha = scal[S_RZ] / s0;
        scal[S_ALPHA] = alpha;
        scal[S_NALPHA] = -alpha;
    } break;
    case FIN_PCG_BETA: {
        scal[S_ZR] = s0;
        scal[S_BETA] = s0 / scal[S_RZ];
        scal[S_RZ] = s0;
    } break;
    case FIN_PCG_BETA_RES: {  // FIN_PCG_BETA and the residual norm of the same iteration in one launch / one all-reduce
        scal[S_ZR] = s0;
        scal[S_BETA] = s0 / scal[S_RZ];
        scal[S_RZ] = s0;
        const double v = sqrt(s1);
        scal[S_RES] = v;
        if (hist) hist[hslot] = v;
    } break;
    case FIN_CG_ALPHA: {
        scal[S_PAP] = s0;
        const double alpha = scal[S_RR] / s0;
        scal[S_ALPHA] = alpha;
        scal[S_NALPHA] = -alpha;
    } break;
    case FIN_CG_BETA: {
        const double s = scal[S_RR];
        const double beta = s0 / s;
        scal[S_BETA] = beta;
        const double res = sqrt(s * beta);
        scal[S_RES] = res;
        scal[S_RR] = s0;
        if (hist) hist[hslot] = res;
    } break;
    case FIN_BICG_ALPHA: {
        scal[S_ALPHA1] = s0;
        scal[S_APR0] = s1;
        scal[S_ALPHA] = s0 / s1;
    } break;
    case FIN_BICG_OMEGA: {
        scal[S_ASS] = s0;
        scal[S_ASAS] = s1;
        // lucky breakdown: s = 0 after the half step (exact preconditioner, tiny systems) makes
        // As.As = 0; the reference divides 0/0 there and returns NaN.  omega = 0 keeps x and ends
        // the loop with r = s = 0.
        scal[S_OMEGA1] = (s1 == 0.0) ? 0.0 : s0 / s1;
    } break;
    case FIN_BICG_BETA: {
        double beta = s0 / scal[S_ALPHA1];
        beta = beta * (scal[S_ALPHA] / scal[S_OMEGA1]);
        scal[S_BETA] = beta;
        scal[S_RR0] = s0;
        const double res = sqrt(s1);
        scal[S_RES] = res;
        if (hist) hist[hslot] = res;
    } break;
    default: break;
    }
}

}  // namespace

// ------------------------------------------------------------------ host launchers

int build_rowblocks(int nrow, const int *rowptr, int *out)
{
    int nb = 0;
    out[0] = 0;
    int r = 0;
    while (r < nrow) {
        int rows = 0;
        long nnz = 0;
        while (r + rows < nrow && rows < kBlock) {
            const long len = (long)rowptr[r + rows + 1] - rowptr[r + rows];
            if (rows > 0 && nnz + len > kStreamNnz) break;
            nnz += len;
            ++rows;
            if (nnz > kStreamNnz) break;  // single long row: own block
        }
        r += rows;
        out[++nb] = r;
    }
    return nb;
}

int launch_csr(const DevCsr &A, CsrOp op, const CsrArgs &a, bool finest, hipStream_t st, const KernelConfig &cfg)
{
    switch (op) {
    case OP_SPMV: return launch_csr_op<OP_SPMV>(A, a, finest, st, cfg);
    case OP_RESID: return launch_csr_op<OP_RESID>(A, a, finest, st, cfg);
    case OP_JACOBI: return launch_csr_op<OP_JACOBI>(A, a, finest, st, cfg);
    case OP_ADD: return launch_csr_op<OP_ADD>(A, a, finest, st, cfg);
    case OP_SPMV_DOT: return launch_csr_op<OP_SPMV_DOT>(A, a, finest, st, cfg);
    case OP_RESNORM: return launch_csr_op<OP_RESNORM>(A, a, finest, st, cfg);
    case OP_JACOBI_DOT: return launch_csr_op<OP_JACOBI_DOT>(A, a, finest, st, cfg);
    case OP_JACOBI_PROLONG: return launch_csr_op<OP_JACOBI_PROLONG>(A, a, finest, st, cfg);
    case OP_RESID_PAIR: break;  // table kernel only: launch_resid_pair
    }
    return 0;
}

std::vector<int> rowblock_records(int nrow, const int *rowptr, int *nblk)
{
    std::vector<int> rb((size_t)nrow + 2);
    const int nb = build_rowblocks(nrow, rowptr, rb.data());
    std::vector<int> rec((size_t)std::max(nb, 1) * 4, 0);
    for (int k = 0; k < nb; ++k) {
        rec[(size_t)4 * k] = rb[k];
        rec[(size_t)4 * k + 1] = rb[k + 1];
        rec[(size_t)4 * k + 2] = rowptr[rb[k]];
        rec[(size_t)4 * k + 3] = rowptr[rb[k + 1]];
    }
    *nblk = nb;
    return rec;
}

int build_col16(const int *rowptr, const int *col, const int *rec, int nblk, unsigned short *col16, int *cbase)
{
    int n16 = 0;
#pragma omp parallel for schedule(static) reduction(+ : n16)
    for (int k = 0; k < nblk; ++k) {
        const int r0 = rec[(size_t)4 * k], r1 = rec[(size_t)4 * k + 1];
        const int j0 = rec[(size_t)4 * k + 2], j1 = rec[(size_t)4 * k + 3];
        bool fits = !(r1 - r0 == 1 && j1 - j0 > kStreamNnz) && j1 > j0;  // a long row is summed with strided accesses: keeps col[]
        int base = 0x7fffffff, top = -1;
        for (int r = r0; fits && r < r1; ++r) {
            const int s = rowptr[r], e = rowptr[r + 1];
            if (s == e) continue;
            base = std::min(base, col[s]);
            top = std::max(top, col[s]);
            for (int j = s + 1; j < e; ++j) {
                const long d = (long)col[j] - col[j - 1];
                if (d < 0 || d > 65535) fits = false;  // unsorted row or a gap too wide
            }
        }
        if (fits && (top < 0 || (long)top - base > 65535)) fits = false;
        for (int j = j0; j < j1; ++j) col16[j] = 0;
        cbase[k] = -1;
        if (!fits) continue;
        cbase[k] = base;
        ++n16;
        for (int r = r0; r < r1; ++r) {
            const int s = rowptr[r], e = rowptr[r + 1];
            for (int j = s; j < e; ++j) col16[j] = (unsigned short)(j == s ? col[j] - base : col[j] - col[j - 1]);
        }
    }
    return n16;
}

int build_waveblocks(int nrow, const int *rowptr, int *out)
{
    int nb = 0;
    out[0] = 0;
    int r = 0;
    while (r < nrow) {
        int rows = 0;
        long nnz = 0;
        while (r + rows < nrow && rows < 64) {
            const long len = (long)rowptr[r + rows + 1] - rowptr[r + rows];
            if (rows > 0 && nnz + len > kWaveNnz) break;
            nnz += len;
            ++rows;
            if (nnz > kWaveNnz) break;
        }
        r += rows;
        out[++nb] = r;
    }
    return nb;
}

void launch_jacobi_zero(int n, const double *b, const double *d, double dconst, double omega, double *x, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(jacobi_zero_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, b, d, dconst, omega, x);
}

void launch_prolong_agg(int n, const int *agg, const double *xc, double *xf, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(prolong_agg_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, agg, xc, xf);
}

void launch_restrict_agg(int nc, const int *rowptr, const int *col, const double *r, double *bc, hipStream_t st)
{
    if (nc <= 0) return;
    hipLaunchKernelGGL(restrict_agg_kernel, dim3(ew_grid(nc)), dim3(kBlock), 0, st, nc, rowptr, col, r, bc);
}

void launch_restrict_agg_zero(int nc, const int *rowptr, const int *col, const double *r, double *bc, const double *dc, double dconst,
                              double omega, double *xc, hipStream_t st)
{
    if (nc <= 0) return;
    hipLaunchKernelGGL(restrict_agg_zero_kernel, dim3(ew_grid(nc)), dim3(kBlock), 0, st, nc, rowptr, col, r, bc, dc, dconst, omega, xc);
}

void launch_gemv(int n, const double *M, const double *b, double *x, hipStream_t st)
{
    if (n <= 0) return;
    const int rows_per_blk = kBlock / 64;
    hipLaunchKernelGGL(gemv_kernel, dim3((n + rows_per_blk - 1) / rows_per_blk), dim3(kBlock), 0, st, n, M, b, x);
}

void launch_fill(int n, double v, double *x, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, v, x);
}

namespace {
// busy-waits `ticks` of the constant 100 MHz wall clock: an artificial transport latency on a stream (virtual-rank tests)
__global__ void spin_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

void launch_spin(double microseconds, hipStream_t st)
{
    if (microseconds <= 0.0) return;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, (long long)(microseconds * 100.0));
}

void launch_copy_int(int n, const int *x, int *y, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(copy_int_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, x, y);
}

void launch_copy(int n, const double *x, double *y, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(copy_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, x, y);
}

void launch_axpby(int n, double a, const double *x, double b, double *y, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, a, x, b, y);
}

void launch_dot(int n, const double *x, const double *y, double *partial, int *nblk, hipStream_t st)
{
    const int g = ew_grid(n);
    *nblk = g;
    hipLaunchKernelGGL(dot_kernel, dim3(g), dim3(kBlock), 0, st, n, x, y, partial);
}

void launch_dot2(int n, const double *a, const double *b, const double *c, const double *d, double *partial0, double *partial1,
                 int *nblk, hipStream_t st)
{
    const int g = ew_grid(n);
    *nblk = g;
    hipLaunchKernelGGL(dot2_kernel, dim3(g), dim3(kBlock), 0, st, n, a, b, c, d, partial0, partial1);
}

void launch_finalize(Fin code, const double *partial0, const double *partial1, int nblk, double *scal, int slot_a, double *hist,
                     int it, hipStream_t st, int mode, int *iter_ctr, int hist_cap, int nblk1)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(mode == 2 ? 64 : kFinBlock), 0, st, (int)code, mode, partial0, partial1, nblk,
                       nblk1 >= 0 ? nblk1 : nblk, scal, slot_a, hist, it, iter_ctr, hist_cap);
}

void launch_pack(int n, const int *idx, const double *vec, double *sendbuf, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(pack_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, idx, vec, sendbuf);
}

void launch_unpack(int n, const int *pos, const double *buf, double *vec, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(unpack_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, pos, buf, vec);
}

void launch_cg_update(int n, const double *scal, const double *p, const double *Ap, double *x, double *r, double *partial,
                      int *nblk, hipStream_t st)
{
    const int g = ew_grid(n);
    *nblk = g;
    hipLaunchKernelGGL(cg_update_kernel, dim3(g), dim3(kBlock), 0, st, n, scal, p, Ap, x, r, partial);
}

void launch_cg_update_zero(int n, const double *scal, const double *p, const double *Ap, double *x, double *r, double *partial,
                           int *nblk, const double *d, double dconst, double omega, double *z0, hipStream_t st, bool nt)
{
    const int g = ew_grid(n);
    *nblk = g;
    if (nt)
        hipLaunchKernelGGL(cg_update_zero_kernel<true>, dim3(g), dim3(kBlock), 0, st, n, scal, p, Ap, x, r, partial, d, dconst, omega, z0);
    else
        hipLaunchKernelGGL(cg_update_zero_kernel<false>, dim3(g), dim3(kBlock), 0, st, n, scal, p, Ap, x, r, partial, d, dconst, omega, z0);
}

void launch_xp_update(int n, const double *scal, const double *z, double *p, double *x, hipStream_t st)
{
    hipLaunchKernelGGL(xp_update_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, scal, z, p, x);
}

void launch_p_update(int n, const double *scal, const double *z, double *p, hipStream_t st)
{
    hipLaunchKernelGGL(p_update_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, scal, z, p);
}

void launch_bicg_s(int n, const double *scal, const double *r, const double *Ap, double *s, hipStream_t st)
{
    hipLaunchKernelGGL(bicg_s_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, scal, r, Ap, s);
}

void launch_bicg_xr(int n, const double *scal, const double *p1, const double *s1, const double *s, const double *As,
                    const double *r0, double *x, double *r, double *partial0, double *partial1, int *nblk, hipStream_t st)
{
    const int g = ew_grid(n);
    *nblk = g;
    hipLaunchKernelGGL(bicg_xr_kernel, dim3(g), dim3(kBlock), 0, st, n, scal, p1, s1, s, As, r0, x, r, partial0, partial1);
}

void launch_bicg_p(int n, const double *scal, const double *r, const double *Ap, double *p, hipStream_t st)
{
    hipLaunchKernelGGL(bicg_p_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, scal, r, Ap, p);
}

void launch_sdia_f32(const SdiaF32 &A, CsrOp op, const float *x, const float *b, float *y, float omega, hipStream_t st)
{
    const int ngroups = (A.nslice + 3) / 4;
    if (ngroups <= 0) return;
    // float layout working set vs the Infinity Cache (same rule as the fp64 policy)
    const size_t bytes = (size_t)A.slots * 64 * 4 + (size_t)A.nrow * 12;
    const int remap = bytes > (240u << 20) ? 16 : 1;
    const int grid = remap_grid(ngroups, remap);
    if (op == OP_JACOBI)
        hipLaunchKernelGGL((sdia_f32_kernel<OP_JACOBI>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.nslice, ngroups, remap, A.sd_ptr, A.sd_off, A.sd_mask, A.val, x, b, y, omega);
    else
        hipLaunchKernelGGL((sdia_f32_kernel<OP_RESID>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.nslice, ngroups, remap, A.sd_ptr, A.sd_off, A.sd_mask, A.val, x, b, y, omega);
}

void launch_csr_f32(const DevCsr &A, const float *val32, CsrOp op, const float *diag, const float *x, const float *b, float *y, float omega,
                    hipStream_t st)
{
    if (A.nrow <= 0) return;
    const int grid = (A.nrow + kBlock - 1) / kBlock;  // one row per thread: no grid-stride tail on large levels
    if (op == OP_JACOBI)
        hipLaunchKernelGGL((csr_f32_kernel<OP_JACOBI>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.rowptr, A.col, val32, diag, x, b, y, omega);
    else
        hipLaunchKernelGGL((csr_f32_kernel<OP_RESID>), dim3(grid), dim3(kBlock), 0, st, A.nrow, A.rowptr, A.col, val32, diag, x, b, y, omega);
}

void launch_jacobi_zero_f32(int n, const float *b, const float *d, float omega, float *x, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(jacobi_zero_f32_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, b, d, omega, x);
}

void launch_restrict_f32(int nc, const int *rowptr, const int *col, const double *val, const float *r, float *bc, hipStream_t st)
{
    if (nc <= 0) return;
    hipLaunchKernelGGL(csr_rows_f32_kernel, dim3(ew_grid(nc)), dim3(kBlock), 0, st, nc, rowptr, col, val, r, bc, 0);
}

void launch_prolong_csr_f32(int n, const int *rowptr, const int *col, const double *val, const float *xc, float *xf, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(csr_rows_f32_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, rowptr, col, val, xc, xf, 1);
}

void launch_prolong_agg_f32(int n, const int *agg, const float *xc, float *xf, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(prolong_agg_f32_kernel, dim3(ew_grid(n)), dim3(kBlock), 0, st, n, agg, xc, xf);
}

void launch_gemv_f32(int n, const float *M, const float *b, float *x, hipStream_t st)
{
    if (n <= 0) return;
    const int rows_per_blk = kBlock / 64;
    hipLaunchKernelGGL(gemv_f32_kernel, dim3((n + rows_per_blk - 1) / rows_per_blk), dim3(kBlock), 0, st, n, M, b, x);
}

void launch_cvt_d2f(long n, const double *in, float *out, hipStream_t st)
{
    if (n <= 0) return;
    long g = (n + kBlock * 4 - 1) / (kBlock * 4);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(cvt_d2f_kernel, dim3((unsigned)g), dim3(kBlock), 0, st, n, in, out);
}

void launch_cvt_f2d_dot(int n, const float *z32, const double *r64, double *z64, double *partial, int *nblk, hipStream_t st)
{
    const int g = ew_grid(n);
    *nblk = g;
    hipLaunchKernelGGL(cvt_f2d_dot_kernel, dim3(g), dim3(kBlock), 0, st, n, z32, r64, z64, partial);
}

}  // namespace sparsh
