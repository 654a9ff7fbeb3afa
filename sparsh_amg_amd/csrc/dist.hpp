// dist.hpp -- row-block partition of the AMG hierarchy over the GPUs of one node (new design:
// the reference is single-GPU).  Host-side planning only; every rank runs the
// same deterministic host setup on the whole matrix and extracts its own part, so planning needs
// no communication.
#pragma once

#include <vector>

#include "host_setup.hpp"

namespace sparsh {

// Partition of one vector space (the rows of one level) into nranks contiguous ranges.
// Range k is [starts[k], starts[k+1]); it belongs to rank rank_of_range[k].  On levels produced by
// a backward HEM sweep the coarse numbering runs against the fine one, so the ranges
// are handed out in reverse rank order there: a rank's coarse rows stay the aggregates of its own
// fine rows and restriction / prolongation remain almost entirely local.
struct Partition {
    int n = 0, nranks = 1;
    std::vector<int> starts;         // nranks + 1
    std::vector<int> rank_of_range;  // nranks
    std::vector<int> range_of_rank;  // nranks
    int lo(int rank) const { return starts[range_of_rank[rank]]; }
    int hi(int rank) const { return starts[range_of_rank[rank] + 1]; }
    int owner(int i) const;  // rank owning global index i
    static Partition whole(int n, int nranks);  // every rank owns everything (replicated level)
    bool replicated = false;
};

// Balanced contiguous ranges, boundaries rounded to multiples of 64 rows (sliced-ELL slices and
// wave-blocks never straddle ranks).
Partition make_partition(int n, int nranks);
// Partition of the coarse space of prolongator P (fine rows partitioned by `fine`): balanced
// ranges in coarse order, rank order chosen (identity or reversed) to follow the fine ownership.
Partition coarse_partition(const HostCsr &R, const Partition &fine);

struct HaloSeg {
    int peer = 0, off = 0, cnt = 0;
    int start = -1;  // send segments: >= 0 when the entries are the contiguous run [start, start+cnt) of
                     // the own vector (slab partitions: a whole boundary plane) -- sent in place, no packing
};

// Exchange plan of one operator: which entries of the input vector this rank must receive
// (appended after its nloc own entries, grouped by source in ascending global order) and which
// of its own entries every peer needs.
struct HaloPlan {
    int nloc = 0;   // own entries of the input vector
    int nhalo = 0;  // received entries
    std::vector<HaloSeg> recv;   // off relative to the halo region
    std::vector<HaloSeg> send;   // off into send_idx
    std::vector<int> send_idx;   // local indices to pack
    std::vector<int> halo_global;  // global index of each halo entry (tests)
};

// Rows of M owned by `rank` (row space partitioned by rowsP) with columns renumbered into
// [own entries of the column space | halo]; entry order inside each row is preserved, so row sums
// stay bitwise identical to the global operator.  If colsP.replicated the columns stay global and
// the plan is empty.
struct LocalOp {
    HostCsr M;
    HaloPlan plan;
};
LocalOp extract_local(const HostCsr &M, const Partition &rowsP, const Partition &colsP, int rank);

// ---- deep halo (communication-avoiding smoothing) -------------------------------------------------
// A rank that knows its input vector on its own rows and on the K layers of rows around them (layer d = rows at
// graph distance d from the own block, through the columns of A) can run K-1 Jacobi sweeps without talking to
// anybody: sweep s updates its own rows and the layers <= K-s, reading layers <= K-s+1 of the previous iterate.
// After nu sweeps with K = nu+1 the own rows AND layer 1 hold exactly what the global sweep would have produced,
// so the residual that follows needs no exchange either.  One exchange per smoothing leg instead of one per sweep.
//
// Local numbering: [own rows | padding up to a multiple of 64 | layer 1 | ... | layer K], ascending global index
// inside a layer (the padding rows are empty: a launch over the "own" slices never touches a ghost row).  The
// local operator has the rows of layers 0..K-1 (layer K is referenced, never updated); entry order inside every
// row is the global one, so row sums stay bitwise identical to the one-rank operator.
struct DeepPlan {  // exchange of the ghost layers <= depth of one vector
    int depth = 0;
    int nrecv = 0;                 // ghost entries received (= layer_end[depth] - nloc)
    std::vector<HaloSeg> recv;     // per peer: off / cnt into the receive staging buffer
    std::vector<int> recv_pos;     // staging entry k belongs at local index recv_pos[k]
    std::vector<HaloSeg> send;     // per peer: off / cnt into send_idx
    std::vector<int> send_idx;     // own local indices to pack
};

struct DeepLocal {
    int K = 0;
    int nloc = 0;                  // own rows
    int npad = 0;                  // own rows rounded up to a multiple of 64 = local index of the first ghost
    HostCsr M;                     // rows of layers 0..K-1, columns of layers 0..K (local numbering)
    std::vector<int> layer_end;    // layer_end[d] = number of local indices in layers 0..d (d = 0..K); layer_end[0] = npad
    std::vector<int> global_of;    // global index of every local index (own, -1 for padding, all ghost layers)
    std::vector<DeepPlan> plans;   // one per requested depth
    const DeepPlan *plan(int depth) const
    {
        for (const DeepPlan &p : plans)
            if (p.depth == depth) return &p;
        return nullptr;
    }
};
// A: square level operator, P: its row partition.  depths: which exchange plans to build (each <= K).
DeepLocal extract_local_deep(const HostCsr &A, const Partition &P, int rank, int K, const std::vector<int> &depths);

}  // namespace sparsh
