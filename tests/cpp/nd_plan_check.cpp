// Host check of the nested-dissection multifrontal plan (csrc/nd_plan.cpp: nd_make_plan): the ordering, update
// sets, front positions, extend-add maps, forward segments and backward gather lists are run through a plain host
// emulation of what the device kernels of csrc/nd_kernels.hip do (scatter, extend-add by child slot, Gauss-Jordan
// inverse of the pivot block with partial pivoting, the three products, forward pull, backward product) and the
// result is checked against A x = b.  No GPU needed.
//   argv: kind(grid2|grid3|box|rand|blocks|arrow|diag) size leaf [merge_rows [top_merge_rows]]      (box: size = nx * 1000000 + ny * 1000 + nz)
//   env ND_PLAN_ONLY=1: statistics of the plan only (large cases)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "nd_plan.hpp"

using namespace sparsh;

static bool invert(double *M, int n, int ld, double *out)  // as nd_invert_kernel
{
    std::vector<int> piv((size_t)n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(M[(size_t)i * ld + k]) > std::fabs(M[(size_t)p * ld + k])) p = i;
        piv[k] = p;
        if (!(std::fabs(M[(size_t)p * ld + k]) > 0.0)) return false;
        if (p != k)
            for (int j = 0; j < n; ++j) std::swap(M[(size_t)k * ld + j], M[(size_t)p * ld + j]);
        const double rp = 1.0 / M[(size_t)k * ld + k];
        std::vector<double> prow((size_t)n), fcol((size_t)n);
        for (int j = 0; j < n; ++j) {
            prow[j] = (j == k) ? rp : M[(size_t)k * ld + j] * rp;
            fcol[j] = M[(size_t)j * ld + k];
        }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                M[(size_t)i * ld + j] = i == k ? prow[j] : ((j == k) ? -fcol[i] * rp : M[(size_t)i * ld + j] - fcol[i] * prow[j]);
    }
    std::vector<int> cm((size_t)n);
    for (int j = 0; j < n; ++j) cm[j] = j;
    for (int k = n - 1; k >= 0; --k) std::swap(cm[k], cm[piv[k]]);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) out[(size_t)i * ld + j] = M[(size_t)i * ld + cm[j]];
    return true;
}

static void gemm(const double *a, int lda, const double *b, int ldb, double *c, int ldc, int M, int N, int K, double alpha, int beta)
{
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc += a[(size_t)i * lda + k] * b[(size_t)k * ldb + j];
            c[(size_t)i * ldc + j] = beta ? c[(size_t)i * ldc + j] + alpha * acc : alpha * acc;
        }
}

int main(int argc, char **argv)
{
    const std::string kind = argc > 1 ? argv[1] : "grid3";
    const int m = argc > 2 ? std::atoi(argv[2]) : 10;
    const int leaf = argc > 3 ? std::atoi(argv[3]) : 32;
    const int merge = argc > 4 ? std::atoi(argv[4]) : -1;
    std::vector<int> rp(1, 0), ci;
    std::vector<double> v;
    int n = 0;
    if (kind == "grid2" || kind == "grid3" || kind == "box") {
        const int dim = kind == "grid2" ? 2 : 3;
        const int nx = kind == "box" ? m / 1000000 : m, ny = kind == "box" ? (m / 1000) % 1000 : m, nz = kind == "box" ? m % 1000 : (dim == 3 ? m : 1);
        n = nx * ny * nz;
        auto idx = [&](int x, int y, int z) { return (z * ny + y) * nx + x; };
        for (int z = 0; z < nz; ++z)
            for (int y = 0; y < ny; ++y)
                for (int x = 0; x < nx; ++x) {
                    std::vector<std::pair<int, double>> e;
                    e.emplace_back(idx(x, y, z), 2.0 * dim + 0.01 * ((x * 7 + y * 3 + z) % 5));
                    if (x > 0) e.emplace_back(idx(x - 1, y, z), -1.0);
                    if (x < nx - 1) e.emplace_back(idx(x + 1, y, z), -1.05);
                    if (y > 0) e.emplace_back(idx(x, y - 1, z), -1.0);
                    if (y < ny - 1) e.emplace_back(idx(x, y + 1, z), -0.95);
                    if (dim == 3 && z > 0) e.emplace_back(idx(x, y, z - 1), -1.0);
                    if (dim == 3 && z < nz - 1) e.emplace_back(idx(x, y, z + 1), -1.0);
                    std::sort(e.begin(), e.end());
                    for (auto &p : e) {
                        ci.push_back(p.first);
                        v.push_back(p.second);
                    }
                    rp.push_back((int)ci.size());
                }
    } else {
        // rand: random sparse pattern, structurally nonsymmetric, diagonally dominant; blocks: the same but in 3 disconnected
        // pieces of different size (one of them a single row) -- several roots, components inside the dissection
        n = m;
        std::mt19937 rng(12345);
        std::vector<std::vector<std::pair<int, double>>> rows((size_t)n);
        auto piece = [&](int i) { return kind == "blocks" ? (i == 0 ? 0 : (i < n / 3 ? 1 : 2)) : 0; };
        if (kind == "arrow")  // tridiagonal + a full last row and column: diameter 2, no level structure to cut -> one dense block
            for (int i = 0; i + 1 < n; ++i) {
                rows[i].emplace_back(n - 1, -0.01);
                rows[n - 1].emplace_back(i, -0.02);
            }
        for (int i = 0; i < n; ++i) {
            if (kind == "diag") break;  // only the diagonal: every row a component (and a leaf) of its own
            for (int q = 0; q < 3; ++q) {
                const int span = 1 + (int)(rng() % 40);
                int j = i + (int)(rng() % (2 * span + 1)) - span;
                j = std::max(0, std::min(n - 1, j));
                if (j == i || piece(j) != piece(i)) continue;
                rows[i].emplace_back(j, -0.3 - 0.001 * (rng() % 100));
                if (rng() % 2) rows[j].emplace_back(i, -0.2);
            }
        }
        for (int i = 0; i < n; ++i) {
            auto &r = rows[i];
            std::sort(r.begin(), r.end());
            std::vector<std::pair<int, double>> u;
            double off = 0.0;
            for (auto &e : r) {
                if (!u.empty() && u.back().first == e.first)
                    u.back().second += e.second;
                else
                    u.push_back(e);
            }
            for (auto &e : u) off += std::fabs(e.second);
            u.emplace_back(i, off + 1.0 + 0.1 * (i % 3));
            std::sort(u.begin(), u.end());
            for (auto &e : u) {
                ci.push_back(e.first);
                v.push_back(e.second);
            }
            rp.push_back((int)ci.size());
        }
    }
    HostCsr A = HostCsr::alias(n, n, rp.data(), ci.data(), v.data());
    NdParams prm;
    prm.leaf = leaf;
    if (merge >= 0) prm.merge_rows = merge;
    if (argc > 5) prm.top_merge_rows = std::atoi(argv[5]);
    NdPlan P;
    std::string err;
    if (!nd_make_plan(A, prm, P, err)) {
        std::printf("plan failed: %s\n", err.c_str());
        return 2;
    }
    const int nn = (int)P.nodes.size();
    // ---- structural checks
    {
        std::vector<char> seen((size_t)n, 0);
        for (int i = 0; i < n; ++i) {
            if (P.perm[i] < 0 || P.perm[i] >= n || seen[P.perm[i]] || P.inv[P.perm[i]] != i) return std::printf("bad permutation\n"), 3;
            seen[P.perm[i]] = 1;
        }
        int covered = 0;
        for (int k = 0; k < nn; ++k) {
            const NdNode &nd = P.nodes[k];
            covered += nd.np;
            if (nd.parent >= 0 && (nd.parent <= k || P.nodes[nd.parent].level <= nd.level)) return std::printf("bad tree order\n"), 3;
            for (int i = 0; i < nd.nu; ++i) {
                const int w = P.upd_idx[nd.upd + i];
                if (w < nd.first + nd.np || (i && w <= P.upd_idx[nd.upd + i - 1])) return std::printf("bad update set\n"), 3;
            }
        }
        if (covered != n) return std::printf("pivot ranges do not cover the rows\n"), 3;
    }
    if (std::getenv("ND_PLAN_ONLY")) {
        long long seg_len = 0;
        for (const NdSegment &g : P.segs) seg_len += g.p;
        std::printf("n %d nodes %d levels %d (launches per solve %d) max_pivot %d max_children %d factor_MB %.1f (B %.1f, L %.1f) front_MB %.1f\n", n, nn, P.nlevels,
                    2 * P.nlevels, P.max_np, P.max_children, P.factor_bytes() / 1e6, P.b_doubles * 8 / 1e6, P.l_doubles * 8 / 1e6, P.front_bytes() / 1e6);
        for (int l = 0; l < P.nlevels; ++l) {
            long long rows = 0, bbytes = 0, lbytes = 0;
            int mp = 0, mu = 0;
            for (int k : P.level_nodes[l]) {
                const NdNode &nd = P.nodes[k];
                rows += nd.np;
                bbytes += 8ll * nd.np * (nd.np + nd.nu);
                lbytes += 8ll * nd.nu * nd.np;
                mp = std::max(mp, nd.np);
                mu = std::max(mu, nd.nu);
            }
            std::printf("  level %2d: %5zu nodes %7lld rows, max pivot %4d max update %5d, B %.2f MB, L %.2f MB\n", l, P.level_nodes[l].size(), rows, mp, mu, bbytes / 1e6,
                        lbytes / 1e6);
        }
        return 0;
    }
    // ---- numeric factorisation, level by level, as the device does it
    std::vector<double> fronts(P.front_doubles, 0.0), Bm(P.b_doubles, 0.0), Lm(P.l_doubles, 0.0);
    for (size_t q = 0; q < P.a_dst.size(); ++q) fronts[(size_t)P.a_dst[q]] = P.a_val[q];
    for (int l = 0; l < P.nlevels; ++l) {
        for (int s = 0; s < P.max_children; ++s)
            for (int k = 0; k < nn; ++k) {
                const NdNode &c = P.nodes[k];
                if (c.parent < 0 || c.slot != s || P.nodes[c.parent].level != l) continue;
                const NdNode &p = P.nodes[c.parent];
                const int ldc = c.np + c.nu, ldp = p.np + p.nu;
                for (int i = 0; i < c.nu; ++i)
                    for (int j = 0; j < c.nu; ++j)
                        fronts[p.foff + (size_t)P.rel_idx[c.rel + i] * ldp + P.rel_idx[c.rel + j]] += fronts[c.foff + (size_t)(c.np + i) * ldc + c.np + j];
            }
        for (int k : P.level_nodes[l]) {
            const NdNode &nd = P.nodes[k];
            const int p = nd.np, u = nd.nu, ld = p + u;
            double *F = fronts.data() + nd.foff, *Bk = Bm.data() + nd.boff, *Lk = Lm.data() + nd.loff;
            if (!invert(F, p, ld, Bk)) return std::printf("singular pivot block\n"), 4;
            if (u == 0) continue;
            gemm(Bk, ld, F + p, ld, Bk + p, ld, p, u, p, -1.0, 0);
            gemm(F + (size_t)p * ld, ld, Bk, ld, Lk, p, u, p, p, 1.0, 0);
            gemm(F + (size_t)p * ld, ld, Bk + p, ld, F + (size_t)p * ld + p, ld, u, u, p, 1.0, 1);
        }
    }
    // ---- solve, as the device does it: Lh re-laid per target row, then both passes as gathered dots over vec = [c | x | b]
    std::vector<double> Lf(P.l_doubles, 0.0);
    for (const NdSegment &g : P.segs)
        for (int t = 0; t < g.p; ++t) Lf[(size_t)g.dst + t] = Lm[(size_t)g.moff + t];
    std::vector<double> b((size_t)n), x((size_t)n, 0.0), w((size_t)2 * n, std::nan(""));
    for (int i = 0; i < n; ++i) b[i] = 1.0 + 0.37 * ((i * 31) % 17);
    auto vec = [&](int g) { return g < 2 * n ? w[g] : b[g - 2 * n]; };
    auto gdot = [&](const NdRow &R, const std::vector<double> &M, const std::vector<int> &idx) {
        double acc = 0.0;
        for (int t = 0; t < R.len; ++t) acc += M[(size_t)R.moff + t] * vec(idx[(size_t)R.ioff + t]);
        return acc;
    };
    size_t rows_seen = 0;
    for (int l = 1; l < P.nlevels; ++l)
        for (const NdRow &R : P.fwd[l].rows) {
            if (R.bsrc != P.perm[R.out] || P.nodes[P.node_of_row[R.out]].level != l) return std::printf("bad forward row record\n"), 5;
            w[R.out] = b[R.bsrc] - gdot(R, Lf, P.fidx);
        }
    for (int l = P.nlevels - 1; l >= 0; --l)
        for (const NdRow &R : P.bwd[l].rows) {
            const double a = gdot(R, Bm, P.bidx);
            w[n + R.out] = a;
            x[R.bsrc] = a;
            ++rows_seen;
        }
    if (rows_seen != (size_t)n) return std::printf("backward rows do not cover the system\n"), 5;
    double rn = 0.0, bn = 0.0;
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int j = rp[i]; j < rp[i + 1]; ++j) s -= v[j] * x[ci[j]];
        rn += s * s;
        bn += b[i] * b[i];
    }
    const double rel = std::sqrt(rn / bn);
    long long seg_len = 0;
    for (const NdSegment &g : P.segs) seg_len += g.p;
    std::printf("n %d nodes %d levels %d max_pivot %d max_children %d factor_MB %.2f front_MB %.2f relerr %.3e\n", n, nn, P.nlevels, P.max_np,
                P.max_children, P.factor_bytes() / 1e6, P.front_bytes() / 1e6, rel);
    return rel < 1e-12 ? 0 : 1;
}
