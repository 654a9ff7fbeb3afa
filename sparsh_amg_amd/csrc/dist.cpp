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

namespace {

// ghost layers 1..K of the row block [lo, hi) of A: layer d = columns referenced by the rows of layer d-1 that are
// in no earlier layer.  Returned layer-major, ascending global index inside a layer; layer_end[d] counts the
// entries of layers 1..d.
void ghost_layers(const HostCsr &A, int lo, int hi, int K, std::vector<int> &ghosts, std::vector<int> &layer_end)
{
    std::vector<char> seen((size_t)A.nrow, 0);  // whole-level marker: planning runs once per level and rank
    for (int i = lo; i < hi; ++i) seen[i] = 1;
    ghosts.clear();
    layer_end.assign((size_t)K + 1, 0);
    std::vector<int> frontier, next;
    for (int i = lo; i < hi; ++i) frontier.push_back(i);
    // only rows near the block boundary can reach outside: scanning all own rows once is O(nnz of the block)
    for (int d = 1; d <= K; ++d) {
        next.clear();
        for (int i : frontier)
            for (int j = A.rowptr[i]; j < A.rowptr[i + 1]; ++j) {
                const int c = A.col[j];
                if (!seen[c]) {
                    seen[c] = 1;
                    next.push_back(c);
                }
            }
        std::sort(next.begin(), next.end());
        ghosts.insert(ghosts.end(), next.begin(), next.end());
        layer_end[d] = (int)ghosts.size();
        frontier.swap(next);
    }
}

}  // namespace

DeepLocal extract_local_deep(const HostCsr &A, const Partition &P, int rank, int K, const std::vector<int> &depths)
{
    DeepLocal L;
    L.K = K;
    const int lo = P.lo(rank), hi = P.hi(rank);
    const int nloc = hi - lo;
    const int npad = (nloc + 63) / 64 * 64;  // ghost rows start on a slice boundary: launches over "own rows" touch no ghost row
    L.nloc = nloc;
    L.npad = npad;
    std::vector<int> ghosts, gl_end;
    ghost_layers(A, lo, hi, K, ghosts, gl_end);
    L.layer_end.resize((size_t)K + 1);
    for (int d = 0; d <= K; ++d) L.layer_end[d] = npad + gl_end[d];
    L.global_of.assign((size_t)npad + ghosts.size(), -1);
    for (int i = 0; i < nloc; ++i) L.global_of[i] = lo + i;
    for (size_t g = 0; g < ghosts.size(); ++g) L.global_of[(size_t)npad + g] = ghosts[g];
    // global -> local for the ghosts (sorted copy with positions)
    std::vector<std::pair<int, int>> gpos(ghosts.size());
    for (size_t g = 0; g < ghosts.size(); ++g) gpos[g] = {ghosts[g], npad + (int)g};
    std::sort(gpos.begin(), gpos.end());
    auto local_of = [&](int c) {
        if (c >= lo && c < hi) return c - lo;
        auto it = std::lower_bound(gpos.begin(), gpos.end(), std::make_pair(c, -1));
        return it->second;  // present by construction for every column of a row of layers <= K-1
    };
    // local operator: rows of layers 0..K-1 (padding rows are empty)
    HostCsr &M = L.M;
    const int nrows = L.layer_end[K - 1];
    M.nrow = nrows;
    M.ncol = L.layer_end[K];
    M.rp_store.assign((size_t)nrows + 1, 0);
    for (int r = 0; r < nrows; ++r) {
        const int g = L.global_of[r];
        M.rp_store[(size_t)r + 1] = M.rp_store[r] + (g >= 0 ? A.rowptr[g + 1] - A.rowptr[g] : 0);
    }
    M.col_store.resize((size_t)M.rp_store[nrows]);
    M.val_store.resize((size_t)M.rp_store[nrows]);
    M.gcol_store.resize((size_t)M.rp_store[nrows]);
    M.grow_store.assign(L.global_of.begin(), L.global_of.begin() + nrows);
    M.grow0 = lo;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < nrows; ++r) {
        const int g = L.global_of[r];
        if (g < 0) continue;
        int q = M.rp_store[r];
        for (int j = A.rowptr[g]; j < A.rowptr[g + 1]; ++j, ++q) {
            M.gcol_store[q] = A.col[j];
            M.col_store[q] = local_of(A.col[j]);
            M.val_store[q] = A.val[j];
        }
    }
    M.adopt();
    // exchange plans: what I receive = my ghost layers <= depth, grouped by owner; what I send to peer h = the
    // entries of MY block inside h's ghost layers <= depth, in the order h stores them grouped by owner
    const int G = P.nranks;
    std::vector<std::vector<int>> peer_ghosts((size_t)G), peer_end((size_t)G);
    for (int h = 0; h < G; ++h) {
        if (h == rank) continue;
        ghost_layers(A, P.lo(h), P.hi(h), K, peer_ghosts[h], peer_end[h]);
    }
    for (int depth : depths) {
        DeepPlan pl;
        pl.depth = depth;
        const int ng = gl_end[depth];
        pl.nrecv = ng;
        // receive: staging order = by owner (ascending rank), inside an owner by (layer, global) = my ghost order
        std::vector<std::vector<int>> by_owner((size_t)G);
        for (int g = 0; g < ng; ++g) by_owner[P.owner(ghosts[g])].push_back(g);
        for (int h = 0; h < G; ++h) {
            if (by_owner[h].empty()) continue;
            pl.recv.push_back({h, (int)pl.recv_pos.size(), (int)by_owner[h].size(), -1});
            for (int g : by_owner[h]) pl.recv_pos.push_back(npad + g);
        }
        // send: for every peer h, its ghosts of layers <= depth that I own, in h's ghost order
        for (int h = 0; h < G; ++h) {
            if (h == rank) continue;
            const int nh = peer_end[h][depth];
            HaloSeg seg{h, (int)pl.send_idx.size(), 0, -1};
            for (int g = 0; g < nh; ++g) {
                const int c = peer_ghosts[h][g];
                if (c >= lo && c < hi) pl.send_idx.push_back(c - lo);
            }
            seg.cnt = (int)pl.send_idx.size() - seg.off;
            if (seg.cnt > 0) pl.send.push_back(seg);
        }
        L.plans.push_back(std::move(pl));
    }
    return L;
}

}  // namespace sparsh
