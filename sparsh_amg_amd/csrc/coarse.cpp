// coarse.cpp -- host side of the device coarse direct solver (see coarse.hpp).
#include "coarse.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace sparsh {

namespace {

HostCsr make_csr(int n, std::vector<int> &&rp, std::vector<int> &&ci, std::vector<double> &&v)
{
    HostCsr M;
    M.nrow = M.ncol = n;
    M.rp_store = std::move(rp);
    M.col_store = std::move(ci);
    M.val_store = std::move(v);
    M.adopt();
    return M;
}

}  // namespace

bool bt_make_plan(const HostCsr &A, int target_blocks, int max_block, BtPlan &P, std::string &err, int block_hint)
{
    const int n = A.nrow;
    P = BtPlan();
    P.n = n;
    P.perm = rcm_order(A);
    std::vector<int> inv((size_t)n);
    for (int i = 0; i < n; ++i) inv[P.perm[i]] = i;
    int bw = 0;
    for (int i = 0; i < n; ++i)
        for (int j = A.rowptr[i]; j < A.rowptr[i + 1]; ++j) bw = std::max(bw, std::abs(inv[i] - inv[A.col[j]]));
    P.bw = bw;
    auto up64 = [](int x) { return (x + 63) / 64 * 64; };
    // any B >= bandwidth makes the permuted operator block tridiagonal.  Narrow-band operators (2D
    // grids) take wider blocks, up to 1024, so the chain of dependent steps stays short.
    int B = std::max(up64(std::max(bw, 1)), std::min(up64((n + target_blocks - 1) / std::max(1, target_blocks)), 1024));
    if (block_hint > 0) B = std::max(up64(std::max(bw, 1)), up64(block_hint));
    if (B > max_block) {
        err = "coarsest level too wide for the block-tridiagonal device solve: bandwidth " + std::to_string(bw) + " after RCM (limit " +
              std::to_string(max_block) + "); lower coarse_limit so the hierarchy is extended instead";
        return false;
    }
    if ((size_t)((n + B - 1) / B) * B * B * sizeof(double) > ((size_t)16 << 30)) {
        err = "coarsest level too large for the block-tridiagonal device solve: " + std::to_string(n) + " rows in blocks of " + std::to_string(B) +
              " would need more than 16 GB of factors; lower coarse_limit so the hierarchy is extended instead";
        return false;
    }
    P.B = B;
    P.nb = (n + B - 1) / B;
    P.mid = P.nb / 2;
    // permuted rows with sorted columns, split into diag / out / in pieces
    std::vector<int> rp[3];
    std::vector<int> ci[3];
    std::vector<double> v[3];
    for (int q = 0; q < 3; ++q) rp[q].assign((size_t)n + 1, 0);
    std::vector<std::pair<int, double>> row;
    auto piece_of = [&](int r, int c) {  // 0 diag, 1 out, 2 in
        const int br = r / B, bc = c / B;
        if (bc == br) return 0;
        if (br == P.mid) return 1;
        const int outer = br < P.mid ? br - 1 : br + 1;
        return bc == outer ? 1 : 2;
    };
    for (int r = 0; r < n; ++r) {
        const int o = P.perm[r];
        row.clear();
        for (int j = A.rowptr[o]; j < A.rowptr[o + 1]; ++j) row.emplace_back(inv[A.col[j]], A.val[j]);
        std::sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        for (const auto &e : row) {
            const int q = piece_of(r, e.first);
            ci[q].push_back(e.first);
            v[q].push_back(e.second);
        }
        for (int q = 0; q < 3; ++q) rp[q][(size_t)r + 1] = (int)ci[q].size();
    }
    P.diag = make_csr(n, std::move(rp[0]), std::move(ci[0]), std::move(v[0]));
    P.out = make_csr(n, std::move(rp[1]), std::move(ci[1]), std::move(v[1]));
    P.in = make_csr(n, std::move(rp[2]), std::move(ci[2]), std::move(v[2]));
    P.inT = transpose(P.in);
    return true;
}

void CoarseSolver::release()
{
    nd_.release();
    for (void *p : allocs_) (void)hipFree(p);
    allocs_.clear();
    inv_ = sinv_ = z_ = nullptr;
    g_ = h_ = y_ = xw_ = nullptr;
    wdesc_ = nullptr;
    for (int c = 0; c < 2; ++c) {
        L_[c] = U_[c] = nullptr;
        in_idx_[c] = out_idx_[c] = nullptr;
        chain_m_[c] = 0;
    }
    unrolled_ = false;
    tri_bytes_ = 0;
    windowed_ = false;
    win_ = 0;
    perm_ = nullptr;
    out_ = in_ = BtDevCsr();
    out_ell_ = in_ell_ = BtDevEll();
    steps_.clear();
    n_ = 0;
}

namespace {

template <class T>
T *dev_alloc(std::vector<void *> &allocs, size_t count, std::string &err)
{
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
        err = "hipMalloc of " + std::to_string(count * sizeof(T)) + " bytes failed (coarse direct solver)";
        return nullptr;
    }
    allocs.push_back(p);
    return static_cast<T *>(p);
}

template <class T>
T *dev_upload(std::vector<void *> &allocs, const T *src, size_t count, std::string &err)
{
    T *d = dev_alloc<T>(allocs, count, err);
    if (d && count && hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        err = "hipMemcpy H2D failed (coarse direct solver)";
        return nullptr;
    }
    return d;
}

// ELL copy of a coupling piece for the solve kernel (at most 8 entries per row, else the CSR form is used)
bool upload_ell(std::vector<void *> &allocs, const HostCsr &M, BtDevEll &E, std::string &err)
{
    const int n = M.nrow;
    int k = 0;
    for (int i = 0; i < n; ++i) k = std::max(k, M.rowptr[i + 1] - M.rowptr[i]);
    E = BtDevEll();
    if (k == 0 || k > 8) return true;
    std::vector<int> ci((size_t)k * n);
    std::vector<double> v((size_t)k * n, 0.0);
    for (int i = 0; i < n; ++i) {
        int q = 0;
        for (int j = M.rowptr[i]; j < M.rowptr[i + 1]; ++j, ++q) {
            ci[(size_t)q * n + i] = M.col[j];
            v[(size_t)q * n + i] = M.val[j];
        }
        for (; q < k; ++q) ci[(size_t)q * n + i] = i;
    }
    E.ci = dev_upload(allocs, ci.data(), ci.size(), err);
    E.v = dev_upload(allocs, v.data(), v.size(), err);
    if (!E.ci || !E.v) return false;
    E.k = k;
    return true;
}

bool upload_piece(std::vector<void *> &allocs, const HostCsr &M, BtDevCsr &D, std::string &err)
{
    D.rp = dev_upload(allocs, M.rowptr, (size_t)M.nrow + 1, err);
    D.ci = dev_upload(allocs, M.col, (size_t)M.nnz(), err);
    D.v = dev_upload(allocs, M.val, (size_t)M.nnz(), err);
    return D.rp && D.ci && D.v;
}

}  // namespace

bool CoarseSolver::setup_dense(int n, const double *inv_host, std::string &err)
{
    release();
    inv_ = dev_upload(allocs_, inv_host, (size_t)n * n, err);
    if (!inv_) return false;
    n_ = n;
    dense_ = true;
    return true;
}

namespace {
int up64w(int x) { return (x + 63) / 64 * 64; }
}  // namespace

bool CoarseSolver::setup_bt(const HostCsr &A, hipStream_t st, std::string &err, int *why_failed)
{
    release();
    int why_local = 0;
    int &why = why_failed ? *why_failed : why_local;
    why = 3;
    const auto t0 = std::chrono::steady_clock::now();
    if (!bt_make_plan(A, 32, 6144, plan_, err, block_hint_)) {
        why = 1;
        return false;
    }
    const BtPlan &P = plan_;
    const int n = P.n, B = P.B, nb = P.nb, mid = P.mid;
    BtDevCsr diag, inT;
    std::vector<void *> tmp;  // setup-only device buffers
    auto drop_tmp = [&]() {
        for (void *p : tmp) (void)hipFree(p);
        tmp.clear();
    };
    bool ok = upload_piece(allocs_, P.out, out_, err) && upload_piece(allocs_, P.in, in_, err) && upload_piece(tmp, P.diag, diag, err) &&
              upload_piece(tmp, P.inT, inT, err) && upload_ell(allocs_, P.out, out_ell_, err) && upload_ell(allocs_, P.in, in_ell_, err);
    perm_ = ok ? dev_upload(allocs_, P.perm.data(), (size_t)n, err) : nullptr;
    z_ = ok ? dev_alloc<double>(allocs_, (size_t)n, err) : nullptr;
    sinv_ = ok ? dev_alloc<double>(allocs_, (size_t)nb * B * B, err) : nullptr;
    double *S = ok ? dev_alloc<double>(tmp, (size_t)B * B, err) : nullptr;
    double *S2 = ok ? dev_alloc<double>(tmp, (size_t)B * B, err) : nullptr;
    double *c0 = ok ? dev_alloc<double>(tmp, (size_t)B, err) : nullptr;
    double *c1 = ok ? dev_alloc<double>(tmp, (size_t)B, err) : nullptr;
    int *piv = ok ? dev_alloc<int>(tmp, (size_t)B, err) : nullptr;
    int *cmap = ok ? dev_alloc<int>(tmp, (size_t)B, err) : nullptr;
    int *sing = ok ? dev_alloc<int>(tmp, 1, err) : nullptr;
    if (!(ok && perm_ && z_ && sinv_ && S && S2 && c0 && c1 && piv && cmap && sing)) {
        drop_tmp();
        release();
        return false;
    }
    (void)hipMemsetAsync(sinv_, 0, (size_t)nb * B * B * sizeof(double), st);
    (void)hipMemsetAsync(z_, 0, (size_t)n * sizeof(double), st);
    (void)hipMemsetAsync(sing, 0, sizeof(int), st);
    const size_t bstride = (size_t)B * B;
    auto factor_block = [&](int i, int o1, int o2) {
        const int r0 = i * B, bs = P.block_rows(i);
        bt_launch_diag(r0, bs, B, diag.rp, diag.ci, diag.v, S, st);
        for (int o : {o1, o2})
            if (o >= 0 && o < nb) bt_launch_schur(r0, bs, o * B, P.block_rows(o), B, out_, inT, sinv_ + (size_t)o * bstride, S, st);
        bt_launch_invert(bs, B, S, S2, c0, c1, piv, cmap, sing, sinv_ + (size_t)i * bstride, st);
    };
    for (int i = 0; i < mid; ++i) factor_block(i, i - 1, -1);          // top chain
    for (int i = nb - 1; i > mid; --i) factor_block(i, i + 1, -1);     // bottom chain
    factor_block(mid, mid - 1, mid + 1);                                // middle block
    // interface form of the solve where the band is narrow against the block (see coarse_kernels.hip): the products
    // G_i = S_i^-1 A[i,outer], H_i = S_i^-1 A[i,inner] on the neighbour's window columns, and the window of every block
    const int W = up64w(std::max(P.bw, 1));
    if (allow_windowed_ && 2 * W <= B && nb >= 3) {
        HostCsr outT = transpose(P.out);
        BtDevCsr outT_d;
        std::vector<int> wd((size_t)4 * nb, 0);
        auto top_win = [&](int j, int &w0, int &wn) {
            wn = std::min(W, P.block_rows(j));
            w0 = j * B;
        };
        auto bot_win = [&](int j, int &w0, int &wn) {
            wn = std::min(W, P.block_rows(j));
            w0 = j * B + P.block_rows(j) - wn;
        };
        for (int i = 0; i < nb; ++i) {
            int *d = &wd[(size_t)4 * i];
            if (i < mid) {
                if (i >= 1) bot_win(i - 1, d[0], d[1]);
                top_win(i + 1, d[2], d[3]);
            } else if (i > mid) {
                if (i + 1 < nb) top_win(i + 1, d[0], d[1]);
                bot_win(i - 1, d[2], d[3]);
            } else {
                if (mid >= 1) bot_win(mid - 1, d[0], d[1]);
                if (mid + 1 < nb) top_win(mid + 1, d[2], d[3]);
            }
        }
        bool okw = upload_piece(tmp, outT, outT_d, err);
        g_ = okw ? dev_alloc<double>(allocs_, (size_t)nb * B * W, err) : nullptr;
        h_ = okw ? dev_alloc<double>(allocs_, (size_t)nb * B * W, err) : nullptr;
        y_ = okw ? dev_alloc<double>(allocs_, (size_t)n, err) : nullptr;
        xw_ = okw ? dev_alloc<double>(allocs_, (size_t)n, err) : nullptr;
        wdesc_ = okw ? dev_upload(allocs_, wd.data(), wd.size(), err) : nullptr;
        if (!(okw && g_ && h_ && y_ && xw_ && wdesc_)) {
            drop_tmp();
            release();
            return false;
        }
        (void)hipMemsetAsync(g_, 0, (size_t)nb * B * W * sizeof(double), st);
        (void)hipMemsetAsync(h_, 0, (size_t)nb * B * W * sizeof(double), st);
        (void)hipMemsetAsync(y_, 0, (size_t)n * sizeof(double), st);
        (void)hipMemsetAsync(xw_, 0, (size_t)n * sizeof(double), st);
        for (int i = 0; i < nb; ++i) {
            const int *d = &wd[(size_t)4 * i];
            const double *Si = sinv_ + (size_t)i * bstride;
            if (d[1] > 0) bt_launch_winprod(i * B, P.block_rows(i), B, Si, outT_d, d[0], d[1], W, g_ + (size_t)i * B * W, st);
            if (d[3] > 0) bt_launch_winprod(i * B, P.block_rows(i), B, Si, i == mid ? outT_d : inT, d[2], d[3], W, h_ + (size_t)i * B * W, st);
        }
        windowed_ = true;
        win_ = W;
        // the two chains unrolled into block-triangular matrices (coarse_kernels.hip): chain position k = 0 is the block at
        // the chain's end, k = m - 1 the block next to the middle one
        const int mc[2] = {mid, nb - 1 - mid};
        size_t tri = 0;
        for (int c = 0; c < 2; ++c) tri += ((size_t)mc[c] * W * mc[c] * W + (size_t)(mc[c] + 1) * W * (mc[c] + 1) * W) * sizeof(double);
        if (unroll_chain_ && tri <= ((size_t)1 << 30)) {
            bool okt = true;
            for (int c = 0; c < 2 && okt; ++c) {
                const int m = mc[c];
                chain_m_[c] = m;
                auto blk_of = [&](int k) { return c == 0 ? k : nb - 1 - k; };
                // inward-facing window: top chain = last rows, bottom chain = first rows; outward-facing the other one
                auto win_rows = [&](int blk, bool inward, int &first, int &nt) {
                    const int bs = P.block_rows(blk);
                    nt = std::min(W, bs);
                    const bool bottom = inward == (c == 0);
                    first = blk * B + (bottom ? bs - nt : 0);
                };
                std::vector<int> iin((size_t)m * W, -1), iout((size_t)(m + 1) * W, -1);
                for (int k = 0; k < m; ++k) {
                    int f, nt;
                    win_rows(blk_of(k), true, f, nt);
                    for (int q = 0; q < nt; ++q) iin[(size_t)k * W + q] = f + q;
                    win_rows(blk_of(k), false, f, nt);
                    for (int q = 0; q < nt; ++q) iout[(size_t)k * W + q] = f + q;
                }
                {   // the middle block's window that faces this chain
                    const int bs = P.block_rows(mid), nt = std::min(W, bs);
                    const int f = mid * B + (c == 0 ? 0 : bs - nt);
                    for (int q = 0; q < nt; ++q) iout[(size_t)m * W + q] = f + q;
                }
                const int ldl = m * W, ldu = (m + 1) * W;
                L_[c] = dev_alloc<double>(allocs_, (size_t)std::max(ldl, 1) * std::max(ldl, 1), err);
                U_[c] = dev_alloc<double>(allocs_, (size_t)ldu * ldu, err);
                in_idx_[c] = dev_upload(allocs_, iin.data(), iin.size(), err);
                out_idx_[c] = dev_upload(allocs_, iout.data(), iout.size(), err);
                okt = L_[c] && U_[c] && in_idx_[c] && out_idx_[c];
                if (!okt) break;
                (void)hipMemsetAsync(L_[c], 0, (size_t)std::max(ldl, 1) * std::max(ldl, 1) * sizeof(double), st);
                (void)hipMemsetAsync(U_[c], 0, (size_t)ldu * ldu * sizeof(double), st);
                for (int k = 0; k < m; ++k) {  // L[k][k] = I ; L[k][0 .. k) = -C_k L[k-1][0 .. k)
                    int f, nt;
                    win_rows(blk_of(k), true, f, nt);
                    double *Lk = L_[c] + (size_t)k * W * ldl;
                    bt_launch_identity(Lk + (size_t)k * W, ldl, nt, st);
                    if (k >= 1) bt_launch_chain_mul(g_ + (size_t)f * W, nt, W, L_[c] + (size_t)(k - 1) * W * ldl, ldl, k * W, Lk, ldl, st);
                }
                // U[m][m] = I (the middle block's window) ; U[k][k] = I ; U[k][(k+1) .. m] = -D_k U[k+1][(k+1) .. m]
                bt_launch_identity(U_[c] + (size_t)m * W * ldu + (size_t)m * W, ldu, W, st);
                for (int k = m - 1; k >= 0; --k) {
                    int f, nt;
                    win_rows(blk_of(k), false, f, nt);
                    double *Uk = U_[c] + (size_t)k * W * ldu;
                    bt_launch_identity(Uk + (size_t)k * W, ldu, nt, st);
                    bt_launch_chain_mul(h_ + (size_t)f * W, nt, W, U_[c] + (size_t)(k + 1) * W * ldu + (size_t)(k + 1) * W, ldu, (m - k) * W,
                                        Uk + (size_t)(k + 1) * W, ldu, st);
                }
            }
            if (!okt) {
                drop_tmp();
                release();
                return false;
            }
            unrolled_ = true;
            tri_bytes_ = tri;
        }
    }
    int sing_h = 0;
    const bool copied = hipMemcpyAsync(&sing_h, sing, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
    const hipError_t e = hipStreamSynchronize(st);
    drop_tmp();
    if (!copied || e != hipSuccess) {
        err = std::string("device factorisation of the coarsest level failed: ") + hipGetErrorString(e);
        release();
        return false;
    }
    if (sing_h) {
        why = 2;
        err = "coarsest-level matrix is singular (zero pivot in a diagonal block of the block-tridiagonal factorisation)";
        release();
        return false;
    }
    // schedule of a solve: inward steps pair block s of the top chain with block nb-1-s of the bottom
    // chain, then the middle block (final), then the same pairs outward
    auto add = [&](std::vector<int> blks, int mode, int final_) {
        Step s{};
        s.nblk = (int)blks.size();
        for (int q = 0; q < s.nblk; ++q) {
            s.blk[q] = blks[q];
            s.r0[q] = blks[q] * B;
            s.bs[q] = P.block_rows(blks[q]);
        }
        s.mode = mode;
        s.final_ = final_;
        steps_.push_back(s);
    };
    const int ntop = mid, nbot = nb - 1 - mid;
    for (int s = 0; s < std::max(ntop, nbot); ++s) {
        std::vector<int> blks;
        if (s < ntop) blks.push_back(s);
        if (s < nbot) blks.push_back(nb - 1 - s);
        add(blks, 0, 0);
    }
    add({mid}, 0, 1);
    for (int s = std::max(ntop, nbot) - 1; s >= 0; --s) {
        // outward: distance from the middle grows; pair top block mid-1-t with bottom block mid+1+t
        const int t = std::max(ntop, nbot) - 1 - s;
        std::vector<int> blks;
        if (mid - 1 - t >= 0) blks.push_back(mid - 1 - t);
        if (mid + 1 + t < nb) blks.push_back(mid + 1 + t);
        if (!blks.empty()) add(blks, 1, 1);
    }
    // the host pieces are no longer needed (sizes stay in plan_)
    plan_.diag = plan_.out = plan_.in = plan_.inT = HostCsr();
    n_ = n;
    dense_ = false;
    why = 0;
    factor_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

bool CoarseSolver::probe(const HostCsr &A, std::string &err)
{
    if (form_ == 1) {
        BtPlan P;
        return bt_make_plan(A, 32, 6144, P, err, block_hint_);
    }
    return nd_.plan(A, nd_prm_, err);
}

bool CoarseSolver::setup_nd(const HostCsr &A, hipStream_t st, std::string &err, int *why_failed)
{
    release();
    if (!nd_.setup(A, nd_prm_, st, err, why_failed)) return false;
    n_ = A.nrow;
    dense_ = false;
    factor_seconds = nd_.plan_seconds + nd_.factor_seconds;
    return true;
}

void CoarseSolver::solve(const double *b, double *x, hipStream_t st) const
{
    if (n_ <= 0) return;
    if (nd_.ready()) {
        nd_.solve(b, x, st);
        return;
    }
    if (dense_) {
        launch_gemv(n_, inv_, b, x, st);
        return;
    }
    const size_t bstride = (size_t)plan_.B * plan_.B;
    if (windowed_) {
        const int B = plan_.B, nb = plan_.nb;
        bt_launch_prepass(nb, B, n_, bstride, sinv_, perm_, b, y_, st);
        if (unrolled_) {
            const int ldl[2] = {chain_m_[0] * win_, chain_m_[1] * win_}, ldu[2] = {(chain_m_[0] + 1) * win_, (chain_m_[1] + 1) * win_};
            bt_launch_tri_gemv(L_, in_idx_, ldl, ldl, win_, 1, y_, z_, st);           // z on the inward-facing windows
            bt_launch_win_mid(nb, B, n_, win_, plan_.mid, wdesc_, g_, h_, y_, z_, xw_, st);  // z on the outward-facing ones, middle block
            bt_launch_tri_gemv(U_, out_idx_, ldu, ldl, win_, 0, z_, xw_, st);         // x on the outward-facing windows
            bt_launch_win_final(nb, B, n_, win_, plan_.mid, wdesc_, g_, h_, y_, z_, xw_, perm_, x, st);
            return;
        }
        for (const Step &s : steps_)
            bt_launch_win_step(s.blk, s.nblk, s.mode == 1 ? 1 : (s.final_ ? 2 : 0), B, n_, win_, wdesc_, g_, h_, y_, z_, xw_, st);
        bt_launch_win_final(nb, B, n_, win_, plan_.mid, wdesc_, g_, h_, y_, z_, xw_, perm_, x, st);
        return;
    }
    for (const Step &s : steps_)
        bt_launch_solve_step(s.r0, s.bs, s.blk, s.nblk, s.mode, s.final_, plan_.B, bstride, sinv_, perm_, s.mode == 0 ? out_ : in_,
                             s.mode == 0 ? out_ell_ : in_ell_, n_, b, z_, x, st);
}

}  // namespace sparsh
