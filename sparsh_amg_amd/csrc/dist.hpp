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

}  // namespace sparsh
