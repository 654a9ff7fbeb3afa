// nd_solver.cpp -- host driver of the nested-dissection multifrontal coarse solver (see nd_solver.hpp, nd_plan.hpp).
#include "nd_solver.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>

#include "coarse.hpp"

namespace sparsh {

namespace {

template <class T>
T *nd_alloc(std::vector<void *> &allocs, size_t count, std::string &err)
{
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
        err = "hipMalloc of " + std::to_string(count * sizeof(T)) + " bytes failed (nested-dissection coarse solver)";
        return nullptr;
    }
    allocs.push_back(p);
    return static_cast<T *>(p);
}

template <class T>
T *nd_upload(std::vector<void *> &allocs, const std::vector<T> &src, std::string &err)
{
    T *d = nd_alloc<T>(allocs, src.size(), err);
    if (d && !src.empty() && hipMemcpy(d, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) {
        err = "hipMemcpy H2D failed (nested-dissection coarse solver)";
        return nullptr;
    }
    return d;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

void NdSolver::release()
{
    for (void *p : allocs_) (void)hipFree(p);
    allocs_.clear();
    Bm_ = Lm_ = w_ = nullptr;
    perm_ = idx_ = seg_ptr_ = nullptr;
    segs_ = nullptr;
    nodes_ = nullptr;
    lev_.clear();
    n_ = nlevels_ = nnodes_ = leaf_ = max_np_ = launches_ = 0;
    factor_bytes_ = 0;
}

bool NdSolver::plan(const HostCsr &A, const NdParams &prm, std::string &err)
{
    const double t0 = now_s();
    pending_.reset(new NdPlan());
    pending_key_ = A.val;
    if (!nd_make_plan(A, prm, *pending_, err)) {
        pending_.reset();
        return false;
    }
    plan_seconds = now_s() - t0;
    return true;
}

bool NdSolver::setup(const HostCsr &A, const NdParams &prm, hipStream_t st, std::string &err, int *why_failed)
{
    release();
    int why_local = 0;
    int &why = why_failed ? *why_failed : why_local;
    why = 1;
    if (!(pending_ && pending_key_ == A.val && pending_->n == A.nrow && pending_->leaf == prm.leaf) && !plan(A, prm, err)) return false;
    const std::unique_ptr<NdPlan> owned = std::move(pending_);
    const NdPlan &P = *owned;
    why = 3;
    const int n = P.n, nn = (int)P.nodes.size();
    std::vector<void *> tmp;  // setup-only device buffers
    auto fail = [&]() {
        for (void *p : tmp) (void)hipFree(p);
        release();
        return false;
    };
    // ---- persistent arrays
    std::vector<NdDevNode> dn((size_t)nn);
    std::vector<int> idx(P.idx_ints);
    for (int k = 0; k < nn; ++k) {
        const NdNode &nd = P.nodes[k];
        dn[k] = NdDevNode{(long long)nd.foff, (long long)nd.boff, (long long)nd.loff, (long long)nd.ioff, nd.first, nd.np, nd.nu, (int)nd.rel, nd.parent, 0};
        if (nd.rel > (size_t)0x7fffffff) {
            err = "nested dissection: update sets too large for 32-bit offsets";
            why = 1;
            return fail();
        }
        int *ix = idx.data() + nd.ioff;
        for (int t = 0; t < nd.np; ++t) ix[t] = nd.first + t;                              // c part of w
        for (int i = 0; i < nd.nu; ++i) ix[nd.np + i] = n + P.upd_idx[nd.upd + i];          // x part of w
    }
    nodes_ = nd_upload(allocs_, dn, err);
    idx_ = nd_upload(allocs_, idx, err);
    perm_ = nd_upload(allocs_, P.perm, err);
    seg_ptr_ = nd_upload(allocs_, P.seg_ptr, err);
    segs_ = nd_upload(allocs_, P.segs, err);
    Bm_ = nd_alloc<double>(allocs_, P.b_doubles, err);
    Lm_ = nd_alloc<double>(allocs_, P.l_doubles, err);
    w_ = nd_alloc<double>(allocs_, (size_t)2 * n, err);
    if (!(nodes_ && idx_ && perm_ && seg_ptr_ && segs_ && Bm_ && Lm_ && w_)) return fail();
    // level row lists; a level whose rows are long gets one workgroup per row
    lev_.assign((size_t)P.nlevels, Level());
    for (int l = 0; l < P.nlevels; ++l) {
        std::vector<int> rows, rnode;
        long long fw = 0, bw = 0;
        for (int k : P.level_nodes[l]) {
            const NdNode &nd = P.nodes[k];
            for (int r = 0; r < nd.np; ++r) {
                rows.push_back(nd.first + r);
                rnode.push_back(k);
                long long t = 0;
                for (int s = P.seg_ptr[nd.first + r]; s < P.seg_ptr[nd.first + r + 1]; ++s) t += P.segs[s].p;
                fw = std::max(fw, t);
            }
            bw = std::max(bw, (long long)nd.np + nd.nu);
        }
        Level &L = lev_[l];
        L.nrows = (int)rows.size();
        L.rows = nd_upload(allocs_, rows, err);
        L.rnode = nd_upload(allocs_, rnode, err);
        if (!L.rows || !L.rnode) return fail();
        // one wave streams 4 x 512 B per step: beyond ~2000 entries per row the whole workgroup takes the row
        L.wide_fwd = fw > 2048 && L.nrows < 16384;
        L.wide_bwd = bw > 2048 && L.nrows < 16384;
    }
    // ---- setup-only arrays
    double *fronts = nd_alloc<double>(tmp, P.front_doubles, err);
    long long *a_dst = nd_upload(tmp, P.a_dst, err);
    double *a_val = nd_upload(tmp, P.a_val, err);
    int *rel = nd_upload(tmp, P.rel_idx, err);
    int *sing = nd_alloc<int>(tmp, 1, err);
    if (!(fronts && a_dst && a_val && rel && sing)) return fail();
    (void)hipMemsetAsync(fronts, 0, P.front_doubles * sizeof(double), st);
    (void)hipMemsetAsync(sing, 0, sizeof(int), st);
    nd_launch_scatter((long long)P.a_dst.size(), a_dst, a_val, fronts, st);
    // scratch of the whole-chip inversion for pivot blocks above kNdSmallPivot rows
    double *S2 = nullptr, *c0 = nullptr, *c1 = nullptr;
    int *piv = nullptr, *cmap = nullptr;
    if (P.max_np > kNdSmallPivot) {
        size_t big = 0;
        for (const NdNode &nd : P.nodes)
            if (nd.np > kNdSmallPivot) big = std::max(big, (size_t)nd.np * (nd.np + nd.nu));
        S2 = nd_alloc<double>(tmp, big, err);
        c0 = nd_alloc<double>(tmp, (size_t)P.max_np, err);
        c1 = nd_alloc<double>(tmp, (size_t)P.max_np, err);
        piv = nd_alloc<int>(tmp, (size_t)P.max_np, err);
        cmap = nd_alloc<int>(tmp, (size_t)P.max_np, err);
        if (!(S2 && c0 && c1 && piv && cmap)) return fail();
    }
    const double t1 = now_s();
    for (int l = 0; l < P.nlevels; ++l) {
        const std::vector<int> &ln = P.level_nodes[l];
        // extend-add: children of this level's nodes, one pass per child slot
        for (int s = 0; s < P.max_children; ++s) {
            std::vector<int> ch;
            int max_nu = 0;
            for (int k = 0; k < nn; ++k) {
                const NdNode &nd = P.nodes[k];
                if (nd.parent >= 0 && nd.slot == s && P.nodes[nd.parent].level == l && nd.nu > 0) {
                    ch.push_back(k);
                    max_nu = std::max(max_nu, nd.nu);
                }
            }
            if (ch.empty()) continue;
            int *chd = nd_upload(tmp, ch, err);
            if (!chd) return fail();
            nd_launch_extend_add(nodes_, chd, (int)ch.size(), max_nu, rel, fronts, st);
        }
        // pivot-block inverses
        std::vector<int> small;
        for (int k : ln)
            if (P.nodes[k].np <= kNdSmallPivot) small.push_back(k);
        if (!small.empty()) {
            int *sl = nd_upload(tmp, small, err);
            if (!sl) return fail();
            nd_launch_invert(nodes_, sl, (int)small.size(), fronts, Bm_, sing, st);
        }
        for (int k : ln) {
            const NdNode &nd = P.nodes[k];
            if (nd.np <= kNdSmallPivot) continue;
            const int ld = nd.np + nd.nu;
            bt_launch_invert(nd.np, ld, fronts + nd.foff, S2, c0, c1, piv, cmap, sing, Bm_ + nd.boff, st);
        }
        // -D^-1 F12 -> B_k[:, np:], F21 D^-1 -> Lh_k ; then F22 += F21 (-D^-1 F12)
        std::vector<NdGemm> g1, g2;
        std::vector<int> t1v, t2v;
        auto add = [](std::vector<NdGemm> &gs, std::vector<int> &ts, NdGemm g) {
            g.tiles_n = (g.N + 63) / 64;
            const int tiles = ((g.M + 63) / 64) * g.tiles_n;
            for (int t = 0; t < tiles; ++t) {
                ts.push_back((int)gs.size());
                ts.push_back(t);
            }
            gs.push_back(g);
        };
        for (int k : ln) {
            const NdNode &nd = P.nodes[k];
            if (nd.nu == 0) continue;
            const int p = nd.np, u = nd.nu, ld = p + u;
            double *F = fronts + nd.foff, *Bk = Bm_ + nd.boff, *Lk = Lm_ + nd.loff;
            add(g1, t1v, NdGemm{Bk, F + p, Bk + p, ld, ld, ld, p, u, p, -1.0, 0, 0});
            add(g1, t1v, NdGemm{F + (size_t)p * ld, Bk, Lk, ld, ld, p, u, p, p, 1.0, 0, 0});
            if (nd.parent >= 0) add(g2, t2v, NdGemm{F + (size_t)p * ld, Bk + p, F + (size_t)p * ld + p, ld, ld, ld, u, u, p, 1.0, 1, 0});
        }
        for (int pass = 0; pass < 2; ++pass) {
            const std::vector<NdGemm> &gs = pass ? g2 : g1;
            const std::vector<int> &ts = pass ? t2v : t1v;
            if (gs.empty()) continue;
            NdGemm *gd = nd_upload(tmp, gs, err);
            int *td = nd_upload(tmp, ts, err);
            if (!gd || !td) return fail();
            nd_launch_gemm(gd, td, (int)ts.size() / 2, st);
        }
    }
    int sing_h = 0;
    const bool copied = hipMemcpyAsync(&sing_h, sing, sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess;
    const hipError_t e = hipStreamSynchronize(st);
    for (void *p : tmp) (void)hipFree(p);
    tmp.clear();
    if (!copied || e != hipSuccess) {
        err = std::string("device factorisation of the coarsest level failed: ") + hipGetErrorString(e);
        release();
        return false;
    }
    if (sing_h) {
        why = 2;
        err = "coarsest-level matrix is singular (zero pivot in a pivot block of the multifrontal factorisation)";
        release();
        return false;
    }
    n_ = n;
    nlevels_ = P.nlevels;
    nnodes_ = nn;
    leaf_ = P.leaf;
    max_np_ = P.max_np;
    factor_bytes_ = P.factor_bytes();
    launches_ = 1 + (P.nlevels - 1) + P.nlevels;
    factor_seconds = now_s() - t1;
    why = 0;
    return true;
}

void NdSolver::solve(const double *b, double *x, hipStream_t st) const
{
    if (n_ <= 0) return;
    nd_launch_permute(n_, perm_, b, w_, st);
    for (int l = 1; l < nlevels_; ++l) nd_launch_forward(lev_[l].rows, lev_[l].nrows, lev_[l].wide_fwd, seg_ptr_, segs_, Lm_, w_, st);
    for (int l = nlevels_ - 1; l >= 0; --l)
        nd_launch_backward(lev_[l].rows, lev_[l].rnode, lev_[l].nrows, lev_[l].wide_bwd, n_, nodes_, idx_, Bm_, w_, perm_, x, st);
}

}  // namespace sparsh
