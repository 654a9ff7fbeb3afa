// nd_solver.cpp -- host driver of the nested-dissection multifrontal coarse solver (see nd_solver.hpp, nd_plan.hpp).
#include "nd_solver.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

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
    Bm_ = Lf_ = w_ = nullptr;
    bidx_ = fidx_ = nullptr;
    fwd_.clear();
    bwd_.clear();
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
    const double t_up = now_s();
    // ---- persistent arrays: the rows of both passes, their gather lists, the work vector
    Bm_ = nd_alloc<double>(allocs_, P.b_doubles, err);
    Lf_ = nd_alloc<double>(allocs_, P.l_doubles, err);
    w_ = nd_alloc<double>(allocs_, (size_t)2 * n, err);
    bidx_ = nd_upload(allocs_, P.bidx, err);
    fidx_ = nd_upload(allocs_, P.fidx, err);
    if (!(Bm_ && Lf_ && w_ && bidx_ && fidx_)) return fail();
    fwd_.assign((size_t)P.nlevels, Pass());
    bwd_.assign((size_t)P.nlevels, Pass());
    for (int l = 0; l < P.nlevels; ++l)
        for (int pass = 0; pass < 2; ++pass) {
            const NdPass &src = pass ? P.bwd[l] : P.fwd[l];
            Pass &dst = pass ? bwd_[l] : fwd_[l];
            dst.nrows = (int)src.rows.size();
            dst.nwide = src.nwide;
            if (dst.nrows == 0) continue;
            dst.rows = nd_upload(allocs_, src.rows, err);
            if (!dst.rows) return fail();
        }
    // node descriptors of the factorisation kernels
    std::vector<NdDevNode> dn((size_t)nn);
    for (int k = 0; k < nn; ++k) {
        const NdNode &nd = P.nodes[k];
        dn[k] = NdDevNode{(long long)nd.foff, (long long)nd.boff, (long long)nd.loff, (long long)nd.ioff, nd.first, nd.np, nd.nu, (int)nd.rel, nd.parent, 0};
        if (nd.rel > (size_t)0x7fffffff) {
            err = "nested dissection: update sets too large for 32-bit offsets";
            why = 1;
            return fail();
        }
    }
    NdDevNode *nodes_ = nd_upload(tmp, dn, err);
    double *Lm_ = nd_alloc<double>(tmp, P.l_doubles, err);  // Lh as the products leave it (row-major per source node); re-laid into Lf_ at the end
    NdSegment *segs_d = nd_upload(tmp, P.segs, err);
    if (!(nodes_ && Lm_ && segs_d)) return fail();
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
    // scratch of the batched whole-chip inversion (pivot blocks above kNdTinyPivot rows): per level the second buffers of its blocks
    size_t gj_doubles = 0, gj_rows = 0;
    for (int l = 0; l < P.nlevels; ++l) {
        size_t d = 0, r = 0;
        for (int k : P.level_nodes[l])
            if (P.nodes[k].np > kNdTinyPivot) {
                d += (size_t)P.nodes[k].np * P.nodes[k].np;
                r += (size_t)P.nodes[k].np;
            }
        gj_doubles = std::max(gj_doubles, d);
        gj_rows = std::max(gj_rows, r);
    }
    double *gj_buf = nullptr, *gj_col = nullptr;
    int *gj_int = nullptr;
    if (gj_rows > 0) {
        gj_buf = nd_alloc<double>(tmp, gj_doubles, err);
        gj_col = nd_alloc<double>(tmp, 2 * gj_rows, err);
        gj_int = nd_alloc<int>(tmp, 2 * gj_rows, err);
        if (!(gj_buf && gj_col && gj_int)) return fail();
    }
    const double t1 = now_s();
    // SPARSH_ND_TIMING=1: synchronise after every phase and print where the factorisation time goes
    const bool timing = std::getenv("SPARSH_ND_TIMING") != nullptr;
    double t_ext = 0.0, t_inv = 0.0, t_big = 0.0, t_gemm = 0.0, t_last = 0.0;
    auto lap = [&](double &acc) {  // time since the previous lap goes to `acc`
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const double t = now_s();
        acc += t - t_last;
        t_last = t;
    };
    if (timing) {
        (void)hipStreamSynchronize(st);
        t_last = now_s();
        std::printf("nd timing: plan %.3f s, uploads + scatter %.3f s\n", plan_seconds, t_last - t_up);
    }
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
        lap(t_ext);
        // pivot-block inverses
        std::vector<int> small;
        std::vector<NdGjNode> big;
        std::vector<int> wg_node;
        {
            size_t dpos = 0, rpos = 0;
            for (int k : ln) {
                const NdNode &nd = P.nodes[k];
                if (nd.np <= kNdTinyPivot) {
                    small.push_back(k);
                    continue;
                }
                NdGjNode g;
                g.a = fronts + nd.foff;
                g.lda = nd.np + nd.nu;
                g.b = gj_buf + dpos;
                g.ldb = nd.np;
                g.c0 = gj_col + 2 * rpos;
                g.c1 = gj_col + 2 * rpos + nd.np;
                g.piv = gj_int + 2 * rpos;
                g.cmap = gj_int + 2 * rpos + nd.np;
                g.out = Bm_ + nd.boff;
                g.ldo = nd.np + nd.nu;
                g.p = nd.np;
                g.wg0 = (int)wg_node.size();
                const int nwg = (nd.np + 7) / 8;
                for (int q = 0; q < nwg; ++q) wg_node.push_back((int)big.size());
                big.push_back(g);
                dpos += (size_t)nd.np * nd.np;
                rpos += (size_t)nd.np;
            }
        }
        if (!small.empty()) {
            int *sl = nd_upload(tmp, small, err);
            if (!sl) return fail();
            nd_launch_invert(nodes_, sl, (int)small.size(), fronts, Bm_, sing, st);
        }
        lap(t_inv);
        if (!big.empty()) {
            int max_p = 0;
            for (const NdGjNode &g : big) max_p = std::max(max_p, g.p);
            NdGjNode *bd = nd_upload(tmp, big, err);
            int *wd = nd_upload(tmp, wg_node, err);
            if (!bd || !wd) return fail();
            nd_launch_gj_batched(bd, (int)big.size(), wd, (int)wg_node.size(), max_p, sing, st);
        }
        lap(t_big);
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
        lap(t_gemm);
    }
    if (timing) std::printf("nd timing: extend-add %.3f s, one-workgroup inversions %.3f s, whole-chip inversions %.3f s, products %.3f s\n", t_ext, t_inv, t_big, t_gemm);
    nd_launch_repack((long long)P.segs.size(), segs_d, Lm_, Lf_, st);
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
    // everything a solve streams: the factor rows, their gather indices (one int32 per forward element, one list per node for the
    // backward pass) and the row records
    factor_bytes_ = P.factor_bytes() + (P.fidx.size() + P.bidx.size()) * sizeof(int) + (size_t)2 * n * sizeof(NdRow);
    launches_ = 2 * P.nlevels - 1;
    factor_seconds = now_s() - t1;
    why = 0;
    return true;
}

void NdSolver::solve(const double *b, double *x, hipStream_t st, NdProlong pr) const
{
    if (n_ <= 0) return;
    for (int l = 1; l < nlevels_; ++l) nd_launch_pass(true, fwd_[l].rows, fwd_[l].nrows, fwd_[l].nwide, n_, Lf_, fidx_, w_, b, x, st);
    for (int l = nlevels_ - 1; l >= 0; --l) nd_launch_pass(false, bwd_[l].rows, bwd_[l].nrows, bwd_[l].nwide, n_, Bm_, bidx_, w_, b, x, st, pr);
}

}  // namespace sparsh
