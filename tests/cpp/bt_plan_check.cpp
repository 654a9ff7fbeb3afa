// Host check of the block-tridiagonal coarse-solver plan (csrc/coarse.cpp: bt_make_plan): the
// pieces (diag / out / in / inT), the block partition and the inward/outward schedule are run
// through a plain dense host emulation of what the device kernels do (Schur assembly from `out` and
// `inT`, explicit block inverses, twisted solve) and the result is checked against A x = b.
// No GPU needed.  argv: n_side dim(2|3) [target_blocks]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "coarse.hpp"

using namespace sparsh;

static void invert(std::vector<double> &M, int n)  // Gauss-Jordan with partial pivoting, in place
{
    std::vector<int> piv((size_t)n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(M[(size_t)i * n + k]) > std::fabs(M[(size_t)p * n + k])) p = i;
        piv[k] = p;
        if (p != k)
            for (int j = 0; j < n; ++j) std::swap(M[(size_t)k * n + j], M[(size_t)p * n + j]);
        const double rp = 1.0 / M[(size_t)k * n + k];
        for (int j = 0; j < n; ++j) M[(size_t)k * n + j] = (j == k) ? rp : M[(size_t)k * n + j] * rp;
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            const double f = M[(size_t)i * n + k];
            for (int j = 0; j < n; ++j) M[(size_t)i * n + j] = (j == k) ? -f * rp : M[(size_t)i * n + j] - f * M[(size_t)k * n + j];
        }
    }
    for (int k = n - 1; k >= 0; --k)
        if (piv[k] != k)
            for (int i = 0; i < n; ++i) std::swap(M[(size_t)i * n + k], M[(size_t)i * n + piv[k]]);
}

int main(int argc, char **argv)
{
    const int m = argc > 1 ? std::atoi(argv[1]) : 12;
    const int dim = argc > 2 ? std::atoi(argv[2]) : 3;
    const int target = argc > 3 ? std::atoi(argv[3]) : 8;
    // grid Laplacian with a mild nonsymmetric perturbation (the reference factors with mtype 11, general)
    const int n = dim == 3 ? m * m * m : m * m;
    std::vector<int> rp(1, 0), ci;
    std::vector<double> v;
    auto idx = [&](int x, int y, int z) { return (z * m + y) * m + x; };
    for (int z = 0; z < (dim == 3 ? m : 1); ++z)
        for (int y = 0; y < m; ++y)
            for (int x = 0; x < m; ++x) {
                std::vector<std::pair<int, double>> e;
                e.emplace_back(idx(x, y, z), 2.0 * dim + 0.01 * ((x * 7 + y * 3 + z) % 5));
                if (x > 0) e.emplace_back(idx(x - 1, y, z), -1.0);
                if (x < m - 1) e.emplace_back(idx(x + 1, y, z), -1.05);
                if (y > 0) e.emplace_back(idx(x, y - 1, z), -1.0);
                if (y < m - 1) e.emplace_back(idx(x, y + 1, z), -0.95);
                if (dim == 3 && z > 0) e.emplace_back(idx(x, y, z - 1), -1.0);
                if (dim == 3 && z < m - 1) e.emplace_back(idx(x, y, z + 1), -1.0);
                std::sort(e.begin(), e.end());
                for (auto &t : e) {
                    ci.push_back(t.first);
                    v.push_back(t.second);
                }
                rp.push_back((int)ci.size());
            }
    HostCsr A = HostCsr::alias(n, n, rp.data(), ci.data(), v.data());
    BtPlan P;
    std::string err;
    if (!bt_make_plan(A, target, 1 << 20, P, err)) {
        std::printf("PLAN FAILED %s\n", err.c_str());
        return 1;
    }
    const int B = P.B, nb = P.nb, mid = P.mid;
    if (B < P.bw || B % 64 != 0 || nb != (n + B - 1) / B) return 2;
    // every entry lands in exactly one piece, neighbours only
    if (P.diag.nnz() + P.out.nnz() + P.in.nnz() != A.nnz() || P.inT.nnz() != P.in.nnz()) return 3;
    std::vector<std::vector<double>> Sinv((size_t)nb);
    auto bs_of = [&](int i) { return P.block_rows(i); };
    auto factor = [&](int i, std::vector<int> outers) {
        const int bs = bs_of(i), r0 = i * B;
        std::vector<double> S((size_t)bs * bs, 0.0);
        for (int r = 0; r < bs; ++r)
            for (int j = P.diag.rowptr[r0 + r]; j < P.diag.rowptr[r0 + r + 1]; ++j) S[(size_t)r * bs + (P.diag.col[j] - r0)] = P.diag.val[j];
        for (int o : outers) {
            if (o < 0 || o >= nb) continue;
            const int o0 = o * B, obs = bs_of(o);
            for (int r = 0; r < bs; ++r) {
                std::vector<double> T((size_t)obs, 0.0);
                bool any = false;
                for (int j = P.out.rowptr[r0 + r]; j < P.out.rowptr[r0 + r + 1]; ++j) {
                    const int k = P.out.col[j];
                    if (k < o0 || k >= o0 + obs) continue;
                    any = true;
                    for (int q = 0; q < obs; ++q) T[q] += P.out.val[j] * Sinv[o][(size_t)(k - o0) * obs + q];
                }
                if (!any) continue;
                for (int c = 0; c < bs; ++c) {
                    double acc = 0.0;
                    for (int j = P.inT.rowptr[r0 + c]; j < P.inT.rowptr[r0 + c + 1]; ++j) {
                        const int mm = P.inT.col[j];
                        if (mm >= o0 && mm < o0 + obs) acc += T[mm - o0] * P.inT.val[j];
                    }
                    S[(size_t)r * bs + c] -= acc;
                }
            }
        }
        invert(S, bs);
        Sinv[i] = std::move(S);
    };
    for (int i = 0; i < mid; ++i) factor(i, {i - 1});
    for (int i = nb - 1; i > mid; --i) factor(i, {i + 1});
    factor(mid, {mid - 1, mid + 1});
    // solve with b = A * xref
    std::vector<double> xref((size_t)n), b((size_t)n, 0.0), z((size_t)n, 0.0), x((size_t)n, 0.0);
    for (int i = 0; i < n; ++i) xref[i] = std::sin(0.37 * i) + 0.5;
    for (int i = 0; i < n; ++i)
        for (int j = rp[i]; j < rp[i + 1]; ++j) b[i] += v[j] * xref[ci[j]];
    auto step = [&](int i, int mode, bool fin) {
        const int bs = bs_of(i), r0 = i * B;
        const HostCsr &M = mode == 0 ? P.out : P.in;
        std::vector<double> w((size_t)bs);
        for (int r = 0; r < bs; ++r) {
            double acc = 0.0;
            for (int j = M.rowptr[r0 + r]; j < M.rowptr[r0 + r + 1]; ++j) acc += M.val[j] * z[M.col[j]];
            w[r] = mode == 0 ? b[P.perm[r0 + r]] - acc : acc;
        }
        for (int r = 0; r < bs; ++r) {
            double s = 0.0;
            for (int c = 0; c < bs; ++c) s += Sinv[i][(size_t)r * bs + c] * w[c];
            z[r0 + r] = mode == 0 ? s : z[r0 + r] - s;
            if (fin) x[P.perm[r0 + r]] = z[r0 + r];
        }
    };
    const int ntop = mid, nbot = nb - 1 - mid, depth = std::max(ntop, nbot);
    for (int s = 0; s < depth; ++s) {
        if (s < ntop) step(s, 0, false);
        if (s < nbot) step(nb - 1 - s, 0, false);
    }
    step(mid, 0, true);
    for (int t = 0; t < depth; ++t) {
        if (mid - 1 - t >= 0) step(mid - 1 - t, 1, true);
        if (mid + 1 + t < nb) step(mid + 1 + t, 1, true);
    }
    double err2 = 0.0, ref2 = 0.0;
    for (int i = 0; i < n; ++i) {
        err2 += (x[i] - xref[i]) * (x[i] - xref[i]);
        ref2 += xref[i] * xref[i];
    }
    std::printf("BT n %d bw %d B %d nb %d mid %d relerr %.3e\n", n, P.bw, B, nb, mid, std::sqrt(err2 / ref2));
    return std::sqrt(err2 / ref2) < 1e-10 ? 0 : 4;
}
