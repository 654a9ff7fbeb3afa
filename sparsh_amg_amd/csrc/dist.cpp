// dist.cpp -- partition planning (see dist.hpp).
#include "dist.hpp"

#include <algorithm>
#include <numeric>

namespace sparsh {

int Partition::owner(int i) const
{
    if (replicated) return 0;
    const int k = (int)(std::upper_bound(starts.begin(), starts.end(), i) - starts.begin()) - 1;
    return rank_of_range[std::min(std::max(k, 0), nranks - 1)];
}

Partition Partition::whole(int n, int nranks)
{
    Partition p;
    p.n = n;
    p.nranks = nranks;
    p.replicated = true;
    p.starts.assign((size_t)nranks + 1, 0);
    for (int k = 1; k <= nranks; ++k) p.starts[k] = n;  // range 0 holds everything; lo()/hi() overridden below
    p.rank_of_range.resize((size_t)nranks);
    p.range_of_rank.assign((size_t)nranks, 0);
    std::iota(p.rank_of_range.begin(), p.rank_of_range.end(), 0);
    return p;
}

Partition make_partition(int n, int nranks)
{
    Partition p;
    p.n = n;
    p.nranks = nranks;
    p.starts.resize((size_t)nranks + 1);
    for (int k = 0; k <= nranks; ++k) {
        long s = (long)n * k / nranks;
        s = (s + 32) / 64 * 64;
        if (s > n) s = n;
        p.starts[k] = (int)s;
    }
    p.starts[0] = 0;
    p.starts[nranks] = n;
    for (int k = 1; k <= nranks; ++k) p.starts[k] = std::max(p.starts[k], p.starts[k - 1]);
    p.rank_of_range.resize((size_t)nranks);
    p.range_of_rank.resize((size_t)nranks);
    std::iota(p.rank_of_range.begin(), p.rank_of_range.end(), 0);
    std::iota(p.range_of_rank.begin(), p.range_of_rank.end(), 0);
    return p;
}

Partition coarse_partition(const HostCsr &R, const Partition &fine)
{
    Partition p = make_partition(R.nrow, fine.nranks);
    // does coarse range k mostly aggregate fine rows of rank k (identity) or of rank G-1-k (reversed)?
    long same = 0, rev = 0;
    const int G = fine.nranks;
    const int step = std::max(1, R.nrow / 4096);
    for (int J = 0; J < R.nrow; J += step) {
        if (R.rowptr[J + 1] == R.rowptr[J]) continue;
        const int fr = fine.owner(R.col[R.rowptr[J]]);
        const int k = (int)(std::upper_bound(p.starts.begin(), p.starts.end(), J) - p.starts.begin()) - 1;
        same += (fr == k);
        rev += (fr == G - 1 - k);
    }
    if (rev > same) {
        for (int k = 0; k < G; ++k) p.rank_of_range[k] = G - 1 - k;
    }
    for (int k = 0; k < G; ++k) p.range_of_rank[p.rank_of_range[k]] = k;
    return p;
}

LocalOp extract_local(const HostCsr &M, const Partition &rowsP, const Partition &colsP, int rank)
{
    LocalOp L;
    const int rlo = rowsP.replicated ? 0 : rowsP.lo(rank), rhi = rowsP.replicated ? M.nrow : rowsP.hi(rank);
    const int nrows = rhi - rlo;
    const int j0 = M.rowptr[rlo], j1 = M.rowptr[rhi];
    HostCsr &A = L.M;
    A.nrow = nrows;
    A.rp_store.resize((size_t)nrows + 1);
    for (int i = 0; i <= nrows; ++i) A.rp_store[i] = M.rowptr[rlo + i] - j0;
    A.col_store.assign(M.col + j0, M.col + j1);
    A.val_store.assign(M.val + j0, M.val + j1);
    A.grow0 = rlo;
    if (colsP.replicated) {  // the whole input vector is present on every rank: keep global columns
        A.ncol = M.ncol;
        L.plan.nloc = M.ncol;
        L.plan.nhalo = 0;
        A.adopt();
        return L;
    }
    const int clo = colsP.lo(rank), chi = colsP.hi(rank);
    const int nloc = chi - clo;
    // remote columns of my rows, ascending
    std::vector<int> remote;
    for (int c : A.col_store)
        if (c < clo || c >= chi) remote.push_back(c);
    std::sort(remote.begin(), remote.end());
    remote.erase(std::unique(remote.begin(), remote.end()), remote.end());
    L.plan.nloc = nloc;
    L.plan.nhalo = (int)remote.size();
    L.plan.halo_global = remote;
    // receive segments: runs of equal owner (ascending global order == ascending range order)
    for (size_t q = 0; q < remote.size();) {
        const int peer = colsP.owner(remote[q]);
        size_t e = q;
        while (e < remote.size() && colsP.owner(remote[e]) == peer) ++e;
        L.plan.recv.push_back({peer, (int)q, (int)(e - q), -1});
        q = e;
    }
    A.gcol_store = A.col_store;  // global columns, before renumbering
    // renumber columns
    for (int &c : A.col_store) {
        if (c >= clo && c < chi)
            c -= clo;
        else
            c = nloc + (int)(std::lower_bound(remote.begin(), remote.end(), c) - remote.begin());
    }
    A.ncol = nloc + L.plan.nhalo;
    A.adopt();
    // send lists: what every peer's rows reference inside my column range (ascending, unique) --
    // the same run the peer derives for me on its side
    const int G = rowsP.nranks;
    for (int h = 0; h < G; ++h) {
        if (h == rank) continue;
        const int hlo = rowsP.lo(h), hhi = rowsP.hi(h);
        std::vector<int> need;
        for (int j = M.rowptr[hlo]; j < M.rowptr[hhi]; ++j) {
            const int c = M.col[j];
            if (c >= clo && c < chi) need.push_back(c);
        }
        // columns h owns itself are never "needed"; by construction [clo,chi) is mine, so all count
        std::sort(need.begin(), need.end());
        need.erase(std::unique(need.begin(), need.end()), need.end());
        if (need.empty()) continue;
        HaloSeg seg{h, (int)L.plan.send_idx.size(), (int)need.size(), -1};
        if (need.back() - need.front() + 1 == (int)need.size()) seg.start = need.front() - clo;  // contiguous run
        L.plan.send.push_back(seg);
        for (int c : need) L.plan.send_idx.push_back(c - clo);
    }
    return L;
}

}  // namespace sparsh
