// nd_plan.hpp -- host plan of the nested-dissection multifrontal direct solver for large coarsest levels
// (stands where the reference calls PARDISO: analyse + factor once, src/AMG_coarse_level_solver.cpp:9-62; one
// solve per V-cycle, :64-76.  PARDISO itself is a supernodal LU over a nested-dissection ordering; this is the
// same family of method laid out for the device: few dependent steps, dense blocks).
//
// The symmetrised graph of the operator is dissected recursively (separator = one level of a breadth-first level
// structure rooted at a pseudo-peripheral vertex, thinned to the vertices that really touch the far side).  Every
// tree node k owns a contiguous range of pivot rows P_k in the new numbering (descendants first, separator last)
// and an update set U_k (sorted new indices > P_k: the ancestor rows its elimination touches).  The numeric
// factorisation (device, nd_kernels.hip) is multifrontal: front F_k over P_k + U_k, D_k = F11 inverted
// explicitly, and what a solve needs is kept per node as dense row-major blocks
//     Lh_k = F21 D_k^-1         (u x p)   forward:   c[U_k] -= Lh_k c[P_k]      (pulled by the target rows)
//     B_k  = [D_k^-1 | -D_k^-1 F12]  (p x (p + u))   backward:  x[P_k] = B_k [c[P_k]; x[U_k]]
// Both passes of a solve are then the same operation, row by row: a dot product of a contiguous row of numbers with
// a vector gathered through an index list,  out = sum_t M[moff + t] * vec[idx[ioff + t]]  -- forward rows are the rows of
// Lh re-laid per TARGET row (all contributions to c[r] side by side), backward rows are the rows of B_k.  `vec` is one
// index space over three vectors: [0, n) c (forward result, new numbering), [n, 2n) x (new numbering), [2n, 3n) the
// caller's right-hand side in ITS numbering (rows of leaves read it directly, so no permutation pass is needed).
// A solve is one launch per tree level up (forward, levels >= 1) and one per level down (backward): 2 * height - 1
// dependent launches, against ~ n / bandwidth for the block-tridiagonal chain.
//
// Pure host code, no HIP: tests/cpp/nd_plan_check.cpp runs the plan with a host emulation of the kernels.
#pragma once

#include <cstddef>
#include <string>
#include <vector>

#include "host_setup.hpp"

namespace sparsh {

struct NdNode {
    int first = 0, np = 0;   // pivot rows [first, first + np) in the new numbering
    int nu = 0;              // size of the update set
    int parent = -1;
    int level = 0;           // height above the leaves (leaf = 0; parent = 1 + max over children)
    int slot = 0;            // position among the children of its parent (extend-add passes run slot by slot)
    size_t upd = 0;          // U_k = upd_idx[upd .. upd + nu)
    size_t foff = 0;         // front (np + nu)^2, row-major, ld = np + nu                    (setup only)
    size_t boff = 0;         // B_k, np x (np + nu)
    size_t loff = 0;         // Lh_k, nu x np
    size_t ioff = 0;         // gather list of the backward product: np + nu ints
    size_t rel = 0;          // position of U_k[i] in the parent's front (rel_idx[rel + i])
};

struct NdSegment {           // one contribution to a forward target row: Lh[moff .. moff + p) (a row of Lh of the source node) times
    long long moff;          // c[first .. first + p); `dst` = where that row goes in the per-target-row layout
    long long dst;
    int first, p;
};

struct NdRow {               // one row of a solve pass: out = sum_t M[moff + t] * vec[idx[ioff + t]], t < len
    long long moff, ioff;
    int len;
    int out;                 // new index of the row
    int bsrc;                // the row's index in the caller's numbering (perm[out])
    int pad;
};

struct NdPass {              // rows of one tree level for one pass; the first `nwide` rows are long (one workgroup each)
    std::vector<NdRow> rows;
    int nwide = 0;
};

struct NdPlan {
    int n = 0, leaf = 0;
    int nlevels = 0, max_children = 0, max_np = 0;
    std::vector<int> perm, inv;          // perm[new] = old, inv[old] = new
    std::vector<NdNode> nodes;           // children before parents
    std::vector<int> upd_idx, rel_idx;
    std::vector<int> node_of_row;        // new row -> node
    std::vector<std::vector<int>> level_nodes;  // nodes of each level
    size_t front_doubles = 0, b_doubles = 0, l_doubles = 0, idx_ints = 0;
    // entries of the permuted operator with their position in the fronts buffer (duplicates summed, sorted by position)
    std::vector<long long> a_dst;
    std::vector<double> a_val;
    // forward: segments of every target row (CSR over new rows; rows of leaves have none), the per-target-row layout they
    // are re-laid into (fwd_ptr over new rows) and the vector index of every element of it
    std::vector<int> seg_ptr;
    std::vector<NdSegment> segs;
    std::vector<long long> fwd_ptr;
    std::vector<int> fidx;
    std::vector<int> bidx;               // backward gather list of every node (np + nu ints at ioff)
    std::vector<NdPass> fwd, bwd;        // per level (fwd[0] is empty)
    int wide_len = 384;                  // rows longer than this get a whole workgroup
    size_t factor_bytes() const { return (b_doubles + l_doubles) * sizeof(double); }
    size_t front_bytes() const { return front_doubles * sizeof(double); }
};

struct NdParams {
    int leaf = 64;                 // largest subgraph kept as one dense pivot block
    int merge_rows = 384;          // separators of successive bisections are eliminated as ONE pivot block while their total stays below
                                   // (measured on MI355X, profiles/r03_nd_leaf_merge_sweep.txt: 192 -> 384 saves 2 tree levels = 4 launches per
                                   // solve for +10 % of factor bytes; 768+ pays in the one-workgroup pivot-block inversion at setup)
    int top_merge_rows = 2048;     // the same cap for the root; halved per tree depth until it meets merge_rows
    int max_pivot = 8000;          // largest pivot block (separator or unsplittable subgraph) accepted: the batched Gauss-Jordan keeps one
                                   // scaled pivot row of that many doubles in LDS (64 KB of dynamic LDS per workgroup, minus its reduction scratch)
    double max_dense_fraction = 0.25;  // refuse when the factors (above 256 MB) reach this share of the dense inverse's n^2 numbers: no separators worth the name
    size_t max_factor_bytes = (size_t)12 << 30;
    size_t max_front_bytes = (size_t)32 << 30;
};

// false (err set) when the graph has a piece that cannot be dissected within the limits
bool nd_make_plan(const HostCsr &A, const NdParams &prm, NdPlan &plan, std::string &err);

}  // namespace sparsh
