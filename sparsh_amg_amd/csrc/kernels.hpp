// kernels.hpp -- launch interface of the hand-written gfx950 kernels (kernels.hip).
// Everything here works on device pointers and enqueues on the given HIP stream; nothing
// allocates, synchronises or touches the host (graph-capture safe).
#pragma once

#include <hip/hip_runtime_api.h>

#include <vector>

namespace sparsh {

// level-wide stencil of the sliced-diagonal layout (see DevCsr::sd_tab)
struct SdTable {
    int nd = 0;
    int near = 0;  // 73 / 52 / 31: nd = 7 / 5 / 3 with offsets -1, 0, +1 in the three middle slots; else 0
    int off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double cval[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// Device CSR + the row-block schedule of the CSR-stream kernels.
struct DevCsr {
    int nrow = 0, ncol = 0, nnz = 0;
    int *rowptr = nullptr;
    int *col = nullptr;
    double *val = nullptr;
    // row blocks of the workgroup CSR-stream kernel: one record of 4 ints per block {first row, end row, first
    // entry, end entry} (16-byte aligned); nnz of a block <= kStreamNnz unless it is a single long row
    int *rowblk = nullptr;
    int nblk = 0;
    // optional compressed column indices of the row blocks (csr_rowlane16_kernel): 16-bit deltas -- first entry of a row
    // relative to cbase[block], further entries relative to the previous column of the row; cbase[block] = -1 where a
    // delta does not fit (that block reads col[])
    unsigned short *col16 = nullptr;
    int *cbase = nullptr;
    int nblk16 = 0;  // blocks that use the 16-bit form
    // wave-granular schedule: wave-block k owns rows [waveblk[k], waveblk[k+1]) (<= 64 rows,
    // <= kWaveNnz products unless it is a single long row)
    int *waveblk = nullptr;
    int nwblk = 0;
    // optional sliced-ELL mirror (slices of 64 rows, column-major inside a slice) for operators
    // with near-uniform row length: lane = row, every access of a wave is one contiguous segment
    int nslice = 0;
    int *slice_ptr = nullptr;   // nslice+1 offsets (in entries) into sell_col / sell_val
    int *sell_col = nullptr;
    double *sell_val = nullptr;
    long sell_entries = 0;
    // optional sliced-diagonal mirror for stencil-like operators: per slice of 64 rows the distinct
    // offsets (col - row) are stored once, values sit column-major by offset slot, a 64-bit lane
    // mask per slot says which rows really hold an entry there.  No per-entry column index: 8 B
    // instead of 12 B per stored entry.
    int *sd_ptr = nullptr;              // nslice+1: first slot of every slice (| kSdConstBit, see below)
    int *sd_off = nullptr;              // per slot: col - row
    unsigned long long *sd_mask = nullptr;  // per slot: lanes (rows of the slice) that hold an entry
    double *sd_val = nullptr;           // value blocks (64 values, lane-major) of the slots that need one
    long sd_slots = 0;
    // constant slots: where every present entry of a slot carries the same value (constant-
    // coefficient stencils and their aggregated coarse operators) the value lives once in sd_cval
    // and the slot owns no block: sd_vidx = -1.  Otherwise sd_vidx = index of its block in sd_val.
    int *sd_vidx = nullptr;
    double *sd_cval = nullptr;
    long sd_vblocks = 0;                // blocks stored in sd_val
    // fixed-stride records (kSdRecInts ints per slice) for slices made of <= 8 constant slots: one
    // address computation and one scalar round trip give the kernel everything it needs about the
    // slice (offsets, lane masks, constants, slot count); other slices carry count -1 and go through
    // sd_ptr.  Layout: int off[8]; u64 mask[8]; double cval[8]; int count; pad.
    int *sd_rec = nullptr;
    // level-wide stencil table: when (nearly) all value-free slices draw their (offset, constant)
    // pairs from one set of <= 8, that set travels as a kernel argument and a slice only needs its
    // 8 lane masks (sd_tmask, 64 B per slice; sd_tconf = 1 where the slice conforms).  The wave then
    // issues its x gathers straight from the kernel arguments, in parallel with the mask load,
    // instead of after a record round trip.
    SdTable sd_tab;
    unsigned long long *sd_tmask = nullptr;
    int *sd_tconf = nullptr;
    bool has_sdia() const { return sd_ptr != nullptr; }
    // box grid: the table is the 7-point stencil (-plane, -line, -1, 0, +1, +line, +plane) of an nx x ny x nz grid in lexicographic
    // order, every slice conforms and a row lacks exactly the neighbours that would lie outside the box.  Such a level can run
    // two Jacobi sweeps in one pass over its vectors (sdia_box2_kernel); box_q/ty/cz = the launch plan (box2_plan), box_on =
    // the setup's verdict that the double sweep beats two single ones on this level (timed there, or forced)
    int box_nx = 0, box_ny = 0, box_nz = 0;
    int box_q = 0, box_ty = 0, box_cz = 0;
    bool box_on = false;
    // the same for the single-stage kernel of the launches that carry an epilogue (sdia_box1_kernel): plan and the setup's verdict
    int box1_q = 0, box1_ty = 0, box1_cz = 0;
    bool box1_on = false;
    // rank-local blocks: slices whose rows touch no halo column (interior) / some (boundary)
    int *int_list = nullptr, *bnd_list = nullptr;
    int nint = 0, nbnd = 0;
};

// per-handle choice of the SpMV-type kernel family (A/B measurements; defaults = the fastest measured).
// Lives in the Engine (sparsh_set_kernel_config / sparsh_set_const_slots act on a handle): no
// process-wide state.
struct KernelConfig {
    int kind = 3;       // 0 workgroup CSR-stream, 1 wave CSR-stream, 2 sliced ELL, 3 sliced diagonals; 2 and 3 fall
                        // back (3 -> 2 -> 0) where the operator does not qualify for the mirror
    int vec = 3;        // CSR-stream kernels: 0 one entry per load, 1 two entries per lane (16-B val / 8-B col loads), 2 (workgroup kernel)
                        // col/val staged in LDS and the x gathers issued in row-lane order (csr_rowlane_kernel), 3 = 2 for operators
                        // that stream from HBM (> 240 MB of CSR bytes), 1 below (a cache-resident ragged operator is faster on 1);
                        // 4 = 2 with 16-bit delta-coded column indices where the operator carries them (csr_rowlane16_kernel)
    bool auto_policy = true;  // choose nt / remap per operator from its size (overrides the two below)
    bool nt = true;     // non-temporal loads for the matrix stream
    int remap = 1;      // 0 none, 1 XCD x owns the x-th contiguous eighth, G>1 groups of G row blocks dealt round-robin to XCDs
    bool table = true;  // use the level-wide stencil table where a level has one
    bool tile = false;  // table levels of grid stencils: stage x tiles in LDS (sdia_tile_kernel) on whole-level launches
    bool const_slots = true;  // layout option read at setup: fold constant diagonals of a slice into one scalar
    bool place_search = true; // setup: choose by timing which buffers hold the finest level's iterate, its twin and the Krylov residual
    bool cg_nt = true;        // ... with x, p, Ap, d streamed past the caches (non-temporal): r and z0, which the cycle's first sweep reads,
                              // stay resident (+1.3 % it/s at 216^3 in a same-process A/B)
    bool fuse_cg_zero = true; // PCG: the cg_update kernel also writes the V-cycle's zero-guess sweep of level 0
    int alt_dir = 1;          // consecutive sweeps of a smoothing leg walk the level in alternating directions (CsrArgs::reverse):
                              // 0 never, 1 where a sweep streams more than 640 MB (2.5x the Infinity Cache), 2 always
    bool defer_x = true;       // PCG: x += alpha p rides in the direction update at the end of the iteration (xp_update_kernel) instead of in the
                               // residual update: p is read once for both
    bool zero_start = true;    // double-sweep levels: a leg that starts from a zero guess runs its first three sweeps as one launch that
                               // reads only the right-hand side (launch_box2 from_zero)
    int box1 = 1;              // box-grid levels: the launches with an epilogue of their own (SpMV + dot, last post-sweep + dot / + prolongation,
                               // residual + pair restriction) through the plane-marching kernel (sdia_box1_kernel): 0 never, 1 where the setup
                               // measured it faster than the table kernel (levels of >= 400 000 rows), 2 wherever a plan exists
    int box2 = 1;              // box-grid levels (DevCsr::box_nx): two Jacobi sweeps per launch (sdia_box2_kernel): 0 never, 1 where the
                               // setup measured it faster than two single sweeps (levels of >= 400 000 rows), 2 wherever a plan exists
    bool const_diag = true;    // levels whose diagonal is one constant: the vector kernels that divide by it (zero-guess sweeps fused
                               // into cg_update / the restriction) take it as an argument instead of streaming diag[]
    bool fuse_prolong = true;  // V-cycle: the last post-sweep of a level adds its result to the finer level's iterate itself
                               // (OP_JACOBI_PROLONG) instead of storing it for a prolongation launch
    bool pair_restrict = true; // V-cycle: on levels whose aggregates are the row pairs (2J, 2J+1) the residual kernel also
                               // restricts and writes the coarse level's zero-guess sweep (OP_RESID_PAIR)
    int idx16 = 1;      // layout option read at setup: build the 16-bit delta column form (DevCsr::col16) for 0 no operator,
                        // 1 operators whose default family is the CSR-stream kernel streaming from HBM, 2 every operator
};

// which kernel a launch of launch_csr on operator A runs under cfg (one decision, used by the
// launcher and by every report of it)
enum CsrFamily : int { FAM_CSR_BLOCK = 0, FAM_CSR_WAVE = 1, FAM_SELL = 2, FAM_SDIA = 3, FAM_SDIA_TAB = 4, FAM_CSR_ROWLANE = 5, FAM_CSR_ROWLANE16 = 6 };

constexpr int kBlock = 256;       // threads per workgroup (4 waves)
constexpr int kStreamNnz = 2048;  // products staged in LDS per workgroup (16 KiB)
constexpr int kWaveNnz = 512;     // products staged in LDS per wave (4 KiB) in the wave-granular kernel
constexpr int kSdRecInts = 48;        // 192 B per slice record
constexpr int kSdConstBit = 1 << 30;  // set in sd_ptr[s] when every slot of slice s is a constant slot
constexpr int kSdPlainBit = 1 << 29;  // set in sd_ptr[s] when no slot of slice s is (its value blocks are consecutive)
constexpr int kSdPtrMask = (1 << 29) - 1;
constexpr int kCsrPad = 4;        // zeroed entries appended to col/val so paired loads stay in bounds

// epilogue selector of the CSR-stream kernel: what happens to the row sum s_i = (A x)_i
enum CsrOp : int {
    OP_SPMV = 0,      // y_i = s_i
    OP_RESID = 1,     // y_i = b_i - s_i                          (store_residual)
    OP_JACOBI = 2,    // y_i = x_i + omega*(b_i - s_i)/d_i        (one fused Jacobi sweep, y != x)
    OP_ADD = 3,       // y_i = s_i + y_i                          (transfer_solution, beta = 1)
    OP_SPMV_DOT = 4,  // y_i = s_i ; partial += x_i*s_i           (Ap and p.Ap of CG)
    OP_RESNORM = 5,   // partial += (s_i - b_i)^2                 (residual norm, nothing stored)
    OP_JACOBI_DOT = 6, // Jacobi sweep ; partial += y_i*b_i        (last post-sweep of PCG: z.r)
    OP_JACOBI_PROLONG = 8, // Jacobi sweep whose result is not stored but added to the finer level's iterate: y2_f = 1.0*xn_i + y2_f
                           // for the (at most two) fine rows f of aggregate i -- the last post-sweep of a level and
                           // transfer_solution into the level above in one launch (aggregation P only: every fine row has one owner)
    OP_RESID_PAIR = 7  // levels whose aggregates are the row pairs (2J, 2J+1): y_J = (b - s)_2J + (b - s)_2J+1 and
                       // y2_J = omega*y_J/d_J -- residual, restriction and the coarse level's zero-guess sweep in one
                       // launch (table kernel only: launch_resid_pair)
};

struct CsrArgs {
    const double *x = nullptr;   // input vector (gathered)
    const double *b = nullptr;   // rhs (RESID/JACOBI/RESNORM)
    const double *d = nullptr;   // diagonal (JACOBI)
    double *y = nullptr;         // output
    double *y2 = nullptr;        // second output (RESID_PAIR: the coarse level's zero-guess sweep; d is then the coarse diagonal;
                                 // JACOBI_PROLONG: the finer level's iterate)
    double dconst = 0.0;         // RESID_PAIR with d == nullptr: the coarse level's constant diagonal
    const int *members = nullptr; // JACOBI_PROLONG: two fine rows per row (second -1 for a single); nullptr = rows (2i, 2i+1)
    int nfine = 0;                // JACOBI_PROLONG: rows of the finer level
    double omega = 0.0;
    double *partial = nullptr;   // per-block partial sums (reductions), size >= nblk
    // optional subset launch of the sliced kernels (multi-GPU overlap): process only the slices
    // listed, write reduction partials from index partial_off on
    const int *slice_list = nullptr;
    int nlist = 0;
    int partial_off = 0;
    // walk the row blocks / slices from the last to the first: consecutive sweeps of a smoothing leg alternate, so a sweep starts on
    // the part of the iterate the previous one wrote last (still in the memory-side cache); placement only, results unchanged
    int reverse = 0;
};

// host-side builder of the row-block schedule (returns number of blocks; out sized nrow+1 max)
int build_rowblocks(int nrow, const int *rowptr, int *out);
int build_waveblocks(int nrow, const int *rowptr, int *out);
// the row-block schedule as the 4-int records DevCsr::rowblk holds (4 * *nblk ints)
std::vector<int> rowblock_records(int nrow, const int *rowptr, int *nblk);
// 16-bit delta form of the column indices for the row blocks `rec` (DevCsr::col16 / cbase); col16 holds nnz entries,
// cbase nblk; returns the number of blocks that took the 16-bit form
int build_col16(const int *rowptr, const int *col, const int *rec, int nblk, unsigned short *col16, int *cbase);

// returns the number of per-workgroup partial sums the launch writes (reducing ops)
// `finest`: launch on the finest level (selects a separately named kernel instance for profilers)
int launch_csr(const DevCsr &A, CsrOp op, const CsrArgs &a, bool finest, hipStream_t st, const KernelConfig &cfg);
CsrFamily csr_family(const DevCsr &A, const KernelConfig &cfg);
// Double Jacobi sweep y = J(J(x)) on a box-grid level (DevCsr::box_nx > 0), bitwise what two OP_JACOBI launches give.
// box2_plan fills box_q/ty/cz (false: no plan -- lines too long for the LDS region); box2_applies = the level runs it under cfg
bool box2_plan(DevCsr &A);
bool box2_applies(const DevCsr &A, const KernelConfig &cfg);
// from_zero: the leg starts from x = 0: x is not read, y = J(J(omega b / d)) = the first three sweeps of the leg
void launch_box2(const DevCsr &A, const double *x, const double *b, double *y, double omega, bool finest, hipStream_t st, bool from_zero = false);
// box-grid level whose aggregates pair a point with its neighbour one line (axis +-1) or one plane (axis +-2) up: residual, restriction and
// the coarse zero-guess sweep in one launch (dc == nullptr: constant coarse diagonal dconst); axis < 0: aggregates numbered from the far
// end of the coarse box (J = nc - 1 - lexicographic index)
void launch_box_resid_pair(const DevCsr &A, int axis, const double *x, const double *b, const double *dc, double dconst, double omega,
                           double *bc, double *xc, hipStream_t st);
// Single-stage plane-marching kernel on a box-grid level (sdia_box1_kernel); epi: 0 y = A x + partial x.Ax, 1 Jacobi sweep into y + partial
// y.b, 2 residual + pair restriction (aggregates = row pairs; y = coarse rhs, y2 = coarse zero-guess sweep, d / dconst = coarse diagonal),
// 3 Jacobi sweep added to the finer iterate y2 (members / nfine as OP_JACOBI_PROLONG), 4 plain Jacobi sweep into y (the odd sweep of a leg
// that runs double sweeps).  Returns the number of partial sums written.
bool box1_plan(DevCsr &A, bool shared_cu = false);
bool box1_applies(const DevCsr &A, const KernelConfig &cfg);
int launch_box1(const DevCsr &A, int epi, const CsrArgs &a, bool finest, hipStream_t st);
// OP_RESID_PAIR over the whole of A (a.y = coarse rhs, a.y2 = coarse iterate, a.d = coarse diagonal); applies to operators
// that run the table kernel under cfg -- resid_pair_applies says whether launch_resid_pair may be called
bool resid_pair_applies(const DevCsr &A, const KernelConfig &cfg);
void launch_resid_pair(const DevCsr &A, const CsrArgs &a, bool finest, hipStream_t st, const KernelConfig &cfg);
// whether consecutive sweeps over A alternate their walking direction under cfg (KernelConfig::alt_dir)
bool csr_alternates(const DevCsr &A, const KernelConfig &cfg);
const char *csr_family_name(CsrFamily f);
int sdia_tile_rows(const DevCsr &A, const KernelConfig &cfg);  // rows per workgroup of the LDS-tiled table kernel, 0 = not used
// placement the launcher picks for A under cfg: non-temporal matrix stream, XCD remap mode
void csr_placement(const DevCsr &A, const KernelConfig &cfg, bool *nt, int *remap);

// x_i = omega*b_i/d_i : first Jacobi sweep from a zero guess (bitwise equal to the full sweep)
// (here and below: d == nullptr means every row's diagonal entry is dconst -- constant-coefficient stencils: the stream is not read)
void launch_jacobi_zero(int n, const double *b, const double *d, double dconst, double omega, double *x, hipStream_t st);
// xf_i = xc[agg_i] + xf_i : prolongation for an aggregation P (one unit entry per row)
void launch_prolong_agg(int n, const int *agg, const double *xc, double *xf, hipStream_t st);
// bc[J] = sum of r over aggregate J (R = P^T of an aggregation P: all values 1.0, not read)
void launch_restrict_agg(int nc, const int *rowptr, const int *col, const double *r, double *bc, hipStream_t st);
// the same, and xc[J] = omega*bc[J]/dc[J]: the coarse level's zero-guess sweep in the same launch
void launch_restrict_agg_zero(int nc, const int *rowptr, const int *col, const double *r, double *bc, const double *dc, double dconst,
                              double omega, double *xc, hipStream_t st);
// x = A^{-1} b with the explicit row-major inverse (coarsest level)
void launch_gemv(int n, const double *M, const double *b, double *x, hipStream_t st);

// ---- BLAS-1 (daxpby/daxpbyc kernels, cublasDaxpy/Ddot/Dnrm2, thrust::fill of the reference)
void launch_fill(int n, double v, double *x, hipStream_t st);
void launch_copy(int n, const double *x, double *y, hipStream_t st);
void launch_copy_int(int n, const int *x, int *y, hipStream_t st);  // PMC calibration of 4-byte streams
void launch_spin(double microseconds, hipStream_t st);              // occupies the stream for that long (injected transport latency, tests)
// y = a*x + b*y with host scalars
void launch_axpby(int n, double a, const double *x, double b, double *y, hipStream_t st);
// partial sums of x.y (nblocks returned through *nblk)
void launch_dot(int n, const double *x, const double *y, double *partial, int *nblk, hipStream_t st);

// ---- fp32 preconditioner hierarchy (opt-in; SURVEY §8f-4 "mixed precision"): the V-cycle runs on a
// float copy of the sliced-diagonal layout with float vectors, the Krylov loop stays fp64.
struct SdiaF32 {
    int nrow = 0, nslice = 0;
    const int *sd_ptr = nullptr;                  // shared with the fp64 mirror
    const int *sd_off = nullptr;
    const unsigned long long *sd_mask = nullptr;
    float *val = nullptr;                         // per slot 64 floats
    long slots = 0;
};
// op: OP_JACOBI (y = x + omega (b - A x)/d) or OP_RESID (y = b - A x); all vectors float
void launch_sdia_f32(const SdiaF32 &A, CsrOp op, const float *x, const float *b, float *y, float omega, hipStream_t st);
// same two ops for a level that has no sliced-diagonal mirror (thread per CSR row over a float copy of the values)
void launch_csr_f32(const DevCsr &A, const float *val32, CsrOp op, const float *diag, const float *x, const float *b, float *y, float omega,
                    hipStream_t st);
void launch_jacobi_zero_f32(int n, const float *b, const float *d, float omega, float *x, hipStream_t st);
void launch_restrict_f32(int nc, const int *rowptr, const int *col, const double *val, const float *r, float *bc, hipStream_t st);
void launch_prolong_agg_f32(int n, const int *agg, const float *xc, float *xf, hipStream_t st);
void launch_prolong_csr_f32(int n, const int *rowptr, const int *col, const double *val, const float *xc, float *xf, hipStream_t st);
void launch_gemv_f32(int n, const float *M, const float *b, float *x, hipStream_t st);
void launch_cvt_d2f(long n, const double *in, float *out, hipStream_t st);
// z64 = (double) z32 ; partial += z64 * r64   (preconditioned residual back to fp64 + fused z.r)
void launch_cvt_f2d_dot(int n, const float *z32, const double *r64, double *z64, double *partial, int *nblk, hipStream_t st);

// device-resident scalar slots used by the Krylov loops
enum Slot : int {
    S_RZ = 0, S_PAP, S_ALPHA, S_NALPHA, S_BETA, S_RR, S_RES, S_ZR,
    S_ALPHA1, S_APR0, S_ASS, S_ASAS, S_OMEGA1, S_RR0, S_TMP, S_SUM0, S_SUM1, S_COUNT
};

// finalize codes: reduce partial arrays, then one thread updates the scalar slots
enum Fin : int {
    FIN_STORE = 0,     // scal[slot_a] = sum0
    FIN_SQRT = 1,      // scal[slot_a] = sqrt(sum0) ; hist[it] = that
    FIN_PCG_ALPHA = 2, // pAp=sum0 ; alpha = rz/pAp ; nalpha = -alpha
    FIN_PCG_BETA = 3,  // zr=sum0 ; beta = zr/rz ; rz = zr
    FIN_CG_BETA = 4,   // rr_new=sum0 ; beta = rr_new/rr ; res = sqrt(rr*beta) ; rr = rr_new ; hist
    FIN_CG_ALPHA = 5,  // pAp=sum0 ; alpha = rr/pAp ; nalpha = -alpha
    FIN_BICG_ALPHA = 6,// alpha1 = sum0 (r.r0) ; apr0 = sum1 (Ap.r0) ; alpha = alpha1/apr0
    FIN_BICG_OMEGA = 7,// ass = sum0 (As.s) ; asas = sum1 (As.As) ; omega1 = ass/asas
    FIN_BICG_BETA = 8, // rr0 = sum0 (r.r0) ; rr = sum1 (r.r) ; beta = rr0/alpha1*(alpha/omega1) ; res = sqrt(rr) ; hist
    FIN_PCG_BETA_RES = 9  // FIN_PCG_BETA on sum0 (z.r) and res = sqrt(sum1) (r.r) ; hist
};
// it >= 0: residual-history slot; it < 0: take the slot from the device counter *iter_ctr (graph replays).
// mode 0: reduce the partials and apply `code` (single GPU); mode 1: reduce only, local sums to
// scal[S_SUM0], scal[S_SUM1] (then all-reduced across ranks); mode 2: apply `code` to those sums
void launch_finalize(Fin code, const double *partial0, const double *partial1, int nblk, double *scal, int slot_a,
                     double *hist, int it, hipStream_t st, int mode = 0, int *iter_ctr = nullptr, int hist_cap = 1 << 30,
                     int nblk1 = -1);  // nblk1: length of partial1 when it differs from nblk
// sendbuf[k] = vec[idx[k]] : pack the entries the peers need (halo exchange)
void launch_pack(int n, const int *idx, const double *vec, double *sendbuf, hipStream_t st);
// vec[pos[k]] = buf[k] : scatter received ghost-layer entries to their local positions (deep-halo exchange)
void launch_unpack(int n, const int *pos, const double *buf, double *vec, hipStream_t st);

// Krylov vector updates with device-resident coefficients (no host round trip)
// PCG/CG: x += alpha p ; r += (-alpha) Ap ; partial += r_i^2
void launch_cg_update(int n, const double *scal, const double *p, const double *Ap, double *x, double *r, double *partial,
                      int *nblk, hipStream_t st);
// the same, and z0 = omega * r / d on the new residual: the zero-guess sweep of the V-cycle that follows (PCG)
void launch_cg_update_zero(int n, const double *scal, const double *p, const double *Ap, double *x, double *r, double *partial,
                           int *nblk, const double *d, double dconst, double omega, double *z0, hipStream_t st, bool nt = false);
// p = 1.0*z + beta*p
void launch_p_update(int n, const double *scal, const double *z, double *p, hipStream_t st);
// x += alpha p ; p = 1.0*z + beta*p  (PCG with the x update moved here: launch_cg_update / _zero are then called with x = nullptr)
void launch_xp_update(int n, const double *scal, const double *z, double *p, double *x, hipStream_t st);
// two dots at once: partial0 += a.b, partial1 += c.d
void launch_dot2(int n, const double *a, const double *b, const double *c, const double *d, double *partial0, double *partial1,
                 int *nblk, hipStream_t st);
// BiCGStab: s = r - alpha*Ap
void launch_bicg_s(int n, const double *scal, const double *r, const double *Ap, double *s, hipStream_t st);
// BiCGStab: x = x + alpha*p1 + omega1*s1 ; r = s - omega1*As ; partial0 += r.r0 ; partial1 += r.r
void launch_bicg_xr(int n, const double *scal, const double *p1, const double *s1, const double *s, const double *As,
                    const double *r0, double *x, double *r, double *partial0, double *partial1, int *nblk, hipStream_t st);
// BiCGStab: p = r + beta*(p - omega1*Ap)
void launch_bicg_p(int n, const double *scal, const double *r, const double *Ap, double *p, hipStream_t st);

}  // namespace sparsh
