// nd_plan.cpp -- nested-dissection ordering, symbolic factorisation and solve schedule (see nd_plan.hpp).
#include "nd_plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>

namespace sparsh {

namespace {

// Symmetrised pattern of A without the diagonal.
struct Graph {
    int n = 0;
    std::vector<int> xadj, adj;
};

Graph build_graph(const HostCsr &A)
{
    Graph g;
    const int n = A.nrow;
    g.n = n;
    std::vector<int> deg((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i)
        for (int j = A.rowptr[i]; j < A.rowptr[i + 1]; ++j) {
            const int c = A.col[j];
            if (c == i) continue;
            ++deg[i + 1];
            ++deg[c + 1];
        }
    std::vector<int> xa((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) xa[i + 1] = xa[i] + deg[i + 1];
    std::vector<int> tmp((size_t)xa[n]);
    std::vector<int> pos(xa.begin(), xa.end() - 1);
    for (int i = 0; i < n; ++i)
        for (int j = A.rowptr[i]; j < A.rowptr[i + 1]; ++j) {
            const int c = A.col[j];
            if (c == i) continue;
            tmp[pos[i]++] = c;
            tmp[pos[c]++] = i;
        }
    // sort + unique per vertex (an entry present in both triangles was added twice)
    g.xadj.assign((size_t)n + 1, 0);
    g.adj.reserve(tmp.size());
    for (int i = 0; i < n; ++i) {
        int *b = tmp.data() + xa[i], *e = tmp.data() + xa[i + 1];
        std::sort(b, e);
        e = std::unique(b, e);
        g.adj.insert(g.adj.end(), b, e);
        g.xadj[i + 1] = (int)g.adj.size();
    }
    return g;
}

struct Dissector {
    const Graph &g;
    const NdParams &prm;
    NdPlan &P;
    std::string &err;
    std::vector<int> owner;  // id of the vertex set a vertex currently belongs to (-1: numbered)
    std::vector<int> dist;
    int next_set = 1, next_index = 0;
    bool failed = false;

    Dissector(const Graph &g_, const NdParams &p_, NdPlan &P_, std::string &e_) : g(g_), prm(p_), P(P_), err(e_), owner((size_t)g_.n, 0), dist((size_t)g_.n, 0) {}

    // breadth-first levels of the set `sid` from `root`; order = visit order, lvl_ptr = start of every level in it
    void bfs(int root, int sid, std::vector<int> &order, std::vector<int> &lvl_ptr)
    {
        order.clear();
        lvl_ptr.clear();
        const int tag = next_set++;  // visited = owner switched to tag; switched back by the caller through relabel()
        order.push_back(root);
        owner[root] = tag;
        dist[root] = 0;
        lvl_ptr.push_back(0);
        size_t head = 0;
        int cur = 0;
        while (head < order.size()) {
            const int v = order[head];
            if (dist[v] != cur) {
                cur = dist[v];
                lvl_ptr.push_back((int)head);
            }
            ++head;
            for (int j = g.xadj[v]; j < g.xadj[v + 1]; ++j) {
                const int w = g.adj[j];
                if (owner[w] == sid) {
                    owner[w] = tag;
                    dist[w] = cur + 1;
                    order.push_back(w);
                }
            }
        }
        lvl_ptr.push_back((int)order.size());
        for (int v : order) owner[v] = sid;
    }

    int degree_in(int v, int sid) const
    {
        int d = 0;
        for (int j = g.xadj[v]; j < g.xadj[v + 1]; ++j) d += owner[g.adj[j]] == sid;
        return d;
    }

    int make_node(const std::vector<int> &verts, const std::vector<int> &children)
    {
        NdNode nd;
        nd.first = next_index;
        nd.np = (int)verts.size();
        for (int v : verts) {
            P.perm[next_index] = v;
            P.inv[v] = next_index;
            ++next_index;
            owner[v] = -1;
        }
        int lvl = 0;
        for (int c : children) lvl = std::max(lvl, P.nodes[c].level + 1);
        nd.level = lvl;
        const int id = (int)P.nodes.size();
        P.nodes.push_back(nd);
        int slot = 0;
        for (int c : children) {
            P.nodes[c].parent = id;
            P.nodes[c].slot = slot++;
        }
        P.max_children = std::max(P.max_children, (int)children.size());
        P.max_np = std::max(P.max_np, nd.np);
        return id;
    }

    struct Piece {
        int sid;
        std::vector<int> verts;
    };

    // Separator of one connected set (owner == sid): a thinned level of the breadth-first structure rooted at a
    // pseudo-peripheral vertex.  0 = no usable separator (the set stays one dense block); 1 = `sep` holds it (its
    // vertices get an owner of their own) and `comps` the connected pieces of the rest, each under a fresh set id.
    int split(const std::vector<int> &verts, int sid, std::vector<int> &sep, std::vector<Piece> &comps)
    {
        const int m = (int)verts.size();
        std::vector<int> order, lp;
        int root = verts[0];
        {
            int best = degree_in(root, sid);
            for (int v : verts) {
                const int d = degree_in(v, sid);
                if (d < best) {
                    best = d;
                    root = v;
                }
            }
        }
        bfs(root, sid, order, lp);
        for (int it = 0; it < 4; ++it) {  // repeat from a vertex of smallest degree in the last level while the structure gets deeper
            const int h = (int)lp.size() - 1;
            int cand = order[lp[h - 1]], cd = degree_in(cand, sid);
            for (int q = lp[h - 1]; q < lp[h]; ++q) {
                const int d = degree_in(order[q], sid);
                if (d < cd) {
                    cd = d;
                    cand = order[q];
                }
            }
            std::vector<int> o2, l2;
            bfs(cand, sid, o2, l2);
            if ((int)l2.size() <= (int)lp.size()) break;
            order.swap(o2);
            lp.swap(l2);
            root = cand;
        }
        bfs(root, sid, order, lp);  // dist[] of the structure that is used (the last trial may have been rejected)
        const int h = (int)lp.size() - 1;  // number of levels
        if ((int)order.size() != m || h < 3) return 0;  // (a clique-like piece: dense block)
        // separator level: the smallest level among those that leave at least 40 % of the vertices on either side (a tree of
        // balanced halves is what keeps the number of dependent launches of a solve low), else 25 %, else the median level
        int ms = -1;
        for (int pct : {40, 25}) {
            long best = -1;
            for (int l = 1; l < h - 1; ++l) {
                const long below = lp[l], above = m - lp[l + 1];
                if (below * 100 < (long)m * pct || above * 100 < (long)m * pct) continue;
                const long sz = lp[l + 1] - lp[l];
                if (best < 0 || sz < best) {
                    best = sz;
                    ms = l;
                }
            }
            if (ms >= 0) break;
        }
        if (ms < 0) {
            ms = 1;
            for (int l = 1; l < h - 1; ++l)
                if (lp[l] <= m / 2) ms = l;
        }
        // thin it: a vertex of the level without a neighbour one level further belongs to the near side
        sep.clear();
        for (int q = lp[ms]; q < lp[ms + 1]; ++q) {
            const int v = order[q];
            bool far = false;
            for (int j = g.xadj[v]; j < g.xadj[v + 1] && !far; ++j) {
                const int w = g.adj[j];
                far = owner[w] == sid && dist[w] == ms + 1;
            }
            if (far) sep.push_back(v);
        }
        if (sep.empty()) return 0;
        const int sep_tag = next_set++;
        for (int v : sep) owner[v] = sep_tag;
        std::vector<int> comp, clp;
        for (int v : order) {
            if (owner[v] != sid) continue;  // separator, or already moved into a component
            bfs(v, sid, comp, clp);
            Piece pc;
            pc.sid = next_set++;
            for (int w : comp) owner[w] = pc.sid;
            pc.verts = comp;
            comps.push_back(std::move(pc));
        }
        return 1;
    }

    // `verts` = one connected set with owner == sid.  Returns the id of the subtree's root node (-1 on failure).
    // A node's pivot block is the separator of the set plus, while the total stays within merge_rows, the separators of the
    // pieces it leaves (and of theirs ...): several levels of bisection eliminated as one dense block -- a few more bytes in
    // the factors for fewer tree levels, i.e. fewer dependent launches per solve.
    int dissect(std::vector<int> &verts, int sid, int depth)
    {
        if (failed) return -1;
        const int m = (int)verts.size();
        auto as_leaf = [&]() -> int {
            if (m > prm.max_pivot) {
                failed = true;
                err = "nested dissection: a subgraph of " + std::to_string(m) + " rows has no usable separator (limit " + std::to_string(prm.max_pivot) +
                      " rows per dense pivot block)";
                return -1;
            }
            return make_node(verts, {});
        };
        if (m <= prm.leaf || depth > 200) return as_leaf();
        std::vector<int> sep;
        std::vector<Piece> pieces;
        if (!split(verts, sid, sep, pieces)) return as_leaf();
        if ((int)sep.size() > prm.max_pivot) {
            failed = true;
            err = "nested dissection: separator of " + std::to_string(sep.size()) + " rows exceeds the limit of " + std::to_string(prm.max_pivot);
            return -1;
        }
        for (int round = 1; round < 8; ++round) {
            // split every piece that is still above the leaf size once more; keep the round if the merged block stays small
            std::vector<Piece> next;
            std::vector<int> add, s2;
            bool any = false;
            for (Piece &pc : pieces) {
                std::vector<Piece> sub;
                if ((int)pc.verts.size() > prm.leaf && split(pc.verts, pc.sid, s2, sub)) {
                    any = true;
                    add.insert(add.end(), s2.begin(), s2.end());
                    for (Piece &q : sub) next.push_back(std::move(q));
                } else {
                    Piece keep;
                    keep.sid = pc.sid;
                    keep.verts = pc.verts;
                    next.push_back(std::move(keep));
                }
            }
            if (!any) break;
            // near the root the tree is narrow (1, 2, 4 ... nodes per level) and a level costs two dependent launches whatever it holds:
            // there the cap is top_merge_rows halved per depth, further down merge_rows
            const int cap = std::max(prm.merge_rows, depth < 30 ? prm.top_merge_rows >> depth : 0);
            if ((int)(sep.size() + add.size()) > cap) {
                // undo the round: every vertex back under its piece's set id
                for (const Piece &pc : pieces)
                    for (int v : pc.verts) owner[v] = pc.sid;
                break;
            }
            // the merged block: inner separators first, the outer one last (any order inside one dense block is equivalent)
            add.insert(add.end(), sep.begin(), sep.end());
            sep.swap(add);
            pieces.swap(next);
        }
        std::vector<int> children;
        for (Piece &pc : pieces) {
            const int child = dissect(pc.verts, pc.sid, depth + 1);
            if (child < 0) return -1;
            children.push_back(child);
        }
        return make_node(sep, children);
    }
};

}  // namespace

// Rough size of the nested-dissection factors of A without dissecting it: one breadth-first level structure from a low-degree
// vertex gives the graph's "diameter" h, n ~ h^d its effective dimension d, and the factors of a d-dimensional mesh grow like
// n log n (d ~ 2) or n^(4/3) (d ~ 3).  Calibrated on this repo's measured plans (2D 31 250 / 280 900 rows: 56 / 493 MB; 3D 31 250 /
// 78 608 / 314 928 rows: 143 / 403 / 2391 MB): within a factor 1.5, which is all the level policy needs (build_hierarchy).
size_t nd_estimate_factor_bytes(const HostCsr &A)
{
    const int n = A.nrow;
    if (n <= 1) return 8;
    const Graph g = build_graph(A);
    int root = 0, best = g.xadj[1] - g.xadj[0];
    for (int v = 0; v < n; ++v)
        if (g.xadj[v + 1] - g.xadj[v] < best) {
            best = g.xadj[v + 1] - g.xadj[v];
            root = v;
        }
    // levels of the component of `root` (a second pass from the farthest vertex: about the diameter)
    std::vector<int> dist((size_t)n);
    int h = 1, reached = 0;
    for (int pass = 0; pass < 2; ++pass) {
        std::fill(dist.begin(), dist.end(), -1);
        std::vector<int> q(1, root);
        dist[root] = 0;
        for (size_t head = 0; head < q.size(); ++head) {
            const int v = q[head];
            for (int j = g.xadj[v]; j < g.xadj[v + 1]; ++j) {
                const int w = g.adj[j];
                if (dist[w] < 0) {
                    dist[w] = dist[v] + 1;
                    q.push_back(w);
                }
            }
        }
        root = q.back();
        h = dist[root] + 1;
        reached = (int)q.size();
    }
    const double m = std::max(2, reached);  // (a disconnected operator: the component reached stands for all)
    const double d = h > 1 ? std::log(m) / std::log((double)h) : 3.0;
    const double nn = (double)n;
    const double two_d = 64.0 * nn * std::log2(nn), three_d = 96.0 * std::pow(nn, 4.0 / 3.0);
    const double t = std::min(1.0, std::max(0.0, (d - 2.0) / 0.6));  // d <= 2: planar-like, d >= 2.6: volume-like, blend between
    return (size_t)(two_d + t * (three_d - two_d));
}

bool nd_make_plan(const HostCsr &A, const NdParams &prm, NdPlan &P, std::string &err)
{
    const int n = A.nrow;
    P = NdPlan();
    P.n = n;
    P.leaf = prm.leaf;
    P.perm.assign((size_t)n, -1);
    P.inv.assign((size_t)n, -1);
    if (n <= 0) {
        err = "nested dissection: empty operator";
        return false;
    }
    const Graph g = build_graph(A);
    {
        Dissector D(g, prm, P, err);
        std::vector<int> comp, clp;
        for (int v = 0; v < n; ++v) {
            if (D.owner[v] != 0) continue;
            D.bfs(v, 0, comp, clp);
            const int cid = D.next_set++;
            for (int w : comp) D.owner[w] = cid;
            std::vector<int> cv(comp);
            if (D.dissect(cv, cid, 0) < 0) return false;
        }
        if (D.next_index != n) {
            err = "nested dissection: internal error (rows left unnumbered)";
            return false;
        }
    }
    const int nn = (int)P.nodes.size();
    P.node_of_row.assign((size_t)n, -1);
    for (int k = 0; k < nn; ++k)
        for (int r = 0; r < P.nodes[k].np; ++r) P.node_of_row[P.nodes[k].first + r] = k;
    // ---- symbolic factorisation: U_k = (new neighbours of P_k beyond it) + (children's update sets minus P_k)
    std::vector<std::vector<int>> kids((size_t)nn);
    for (int k = 0; k < nn; ++k)
        if (P.nodes[k].parent >= 0) kids[P.nodes[k].parent].push_back(k);
    for (auto &kv : kids) std::sort(kv.begin(), kv.end(), [&](int a, int b) { return P.nodes[a].slot < P.nodes[b].slot; });
    std::vector<int> mark((size_t)n, -1);
    std::vector<int> u;
    for (int k = 0; k < nn; ++k) {  // children come before parents
        NdNode &nd = P.nodes[k];
        const int last = nd.first + nd.np;
        u.clear();
        for (int r = nd.first; r < last; ++r) {
            const int v = P.perm[r];
            for (int j = g.xadj[v]; j < g.xadj[v + 1]; ++j) {
                const int w = P.inv[g.adj[j]];
                if (w >= last && mark[w] != k) {
                    mark[w] = k;
                    u.push_back(w);
                }
            }
        }
        for (int c : kids[k]) {
            const NdNode &ch = P.nodes[c];
            for (int i = 0; i < ch.nu; ++i) {
                const int w = P.upd_idx[ch.upd + i];
                if (w >= last && mark[w] != k) {
                    mark[w] = k;
                    u.push_back(w);
                }
            }
        }
        std::sort(u.begin(), u.end());
        nd.nu = (int)u.size();
        nd.upd = P.upd_idx.size();
        P.upd_idx.insert(P.upd_idx.end(), u.begin(), u.end());
    }
    // ---- offsets, levels, position of every update row in the parent's front
    P.nlevels = 0;
    for (int k = 0; k < nn; ++k) P.nlevels = std::max(P.nlevels, P.nodes[k].level + 1);
    P.level_nodes.assign((size_t)P.nlevels, {});
    P.rel_idx.assign(P.upd_idx.size(), -1);
    for (int k = 0; k < nn; ++k) {
        NdNode &nd = P.nodes[k];
        const size_t f = (size_t)nd.np + nd.nu;
        nd.foff = P.front_doubles;
        P.front_doubles += f * f;
        nd.boff = P.b_doubles;
        P.b_doubles += (size_t)nd.np * f;
        nd.loff = P.l_doubles;
        P.l_doubles += (size_t)nd.nu * nd.np;
        nd.ioff = P.idx_ints;
        P.idx_ints += f;
        nd.rel = nd.upd;
        P.level_nodes[nd.level].push_back(k);
        if (nd.parent >= 0) {
            const NdNode &pa = P.nodes[nd.parent];
            const int *pu = P.upd_idx.data() + pa.upd;
            for (int i = 0; i < nd.nu; ++i) {
                const int w = P.upd_idx[nd.upd + i];
                int pos;
                if (w < pa.first + pa.np) {
                    pos = w - pa.first;
                } else {
                    const int *it = std::lower_bound(pu, pu + pa.nu, w);
                    if (it == pu + pa.nu || *it != w) {
                        err = "nested dissection: internal error (update row missing in the parent's front)";
                        return false;
                    }
                    pos = pa.np + (int)(it - pu);
                }
                P.rel_idx[nd.rel + i] = pos;
            }
        } else if (nd.nu != 0) {
            err = "nested dissection: internal error (root with a non-empty update set)";
            return false;
        }
    }
    if (P.factor_bytes() > ((size_t)256 << 20) && (double)P.factor_bytes() > prm.max_dense_fraction * 8.0 * (double)n * (double)n) {
        err = "nested dissection: the operator's graph has no usable separators (factors of " + std::to_string(P.factor_bytes() >> 20) + " MB against " +
              std::to_string(((size_t)n * n * 8) >> 20) + " MB for the dense inverse of its " + std::to_string(n) + " rows)";
        return false;
    }
    if (P.factor_bytes() > prm.max_factor_bytes || P.front_bytes() > prm.max_front_bytes) {
        err = "nested dissection: factors of " + std::to_string(P.factor_bytes() >> 20) + " MB (fronts " + std::to_string(P.front_bytes() >> 20) +
              " MB) exceed the budget; lower coarse_limit so the hierarchy is extended instead";
        return false;
    }
    // ---- where every entry of the operator lands in the fronts
    {
        std::vector<std::pair<long long, double>> ent;
        ent.reserve((size_t)A.nnz());
        auto local = [&](const NdNode &nd, int w) -> int {
            if (w < nd.first + nd.np) return w - nd.first;
            const int *pu = P.upd_idx.data() + nd.upd;
            const int *it = std::lower_bound(pu, pu + nd.nu, w);
            return (it != pu + nd.nu && *it == w) ? nd.np + (int)(it - pu) : -1;
        };
        for (int i = 0; i < n; ++i) {
            const int r = P.inv[i];
            for (int j = A.rowptr[i]; j < A.rowptr[i + 1]; ++j) {
                const int c = P.inv[A.col[j]];
                const NdNode &nd = P.nodes[P.node_of_row[std::min(r, c)]];
                const int lr = local(nd, r), lc = local(nd, c);
                if (lr < 0 || lc < 0) {
                    err = "nested dissection: internal error (entry outside its front)";
                    return false;
                }
                ent.emplace_back((long long)(nd.foff + (size_t)lr * (nd.np + nd.nu) + lc), A.val[j]);
            }
        }
        std::stable_sort(ent.begin(), ent.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        for (size_t q = 0; q < ent.size(); ++q) {
            if (!P.a_dst.empty() && P.a_dst.back() == ent[q].first) {
                P.a_val.back() += ent[q].second;  // duplicate entry of the caller's CSR
            } else {
                P.a_dst.push_back(ent[q].first);
                P.a_val.push_back(ent[q].second);
            }
        }
    }
    // ---- forward schedule: a target row pulls one segment from every descendant whose update set holds it
    {
        P.seg_ptr.assign((size_t)n + 1, 0);
        for (int k = 0; k < nn; ++k)
            for (int i = 0; i < P.nodes[k].nu; ++i) ++P.seg_ptr[(size_t)P.upd_idx[P.nodes[k].upd + i] + 1];
        for (int r = 0; r < n; ++r) P.seg_ptr[r + 1] += P.seg_ptr[r];
        P.segs.resize((size_t)P.seg_ptr[n]);
        std::vector<int> pos(P.seg_ptr.begin(), P.seg_ptr.end() - 1);
        for (int k = 0; k < nn; ++k) {  // ascending node index: a fixed order of additions per row
            const NdNode &nd = P.nodes[k];
            for (int i = 0; i < nd.nu; ++i) {
                const int r = P.upd_idx[nd.upd + i];
                P.segs[pos[r]++] = NdSegment{(long long)(nd.loff + (size_t)i * nd.np), 0, nd.first, nd.np};
            }
        }
        // per-target-row layout and the vector index of each of its elements: a source row of a leaf reads the caller's
        // right-hand side directly (index space [c | x | b], see nd_plan.hpp)
        P.fwd_ptr.assign((size_t)n + 1, 0);
        for (int r = 0; r < n; ++r) {
            long long t = 0;
            for (int s2 = P.seg_ptr[r]; s2 < P.seg_ptr[r + 1]; ++s2) {
                P.segs[s2].dst = P.fwd_ptr[r] + t;
                t += P.segs[s2].p;
            }
            P.fwd_ptr[r + 1] = P.fwd_ptr[r] + t;
        }
        if ((size_t)P.fwd_ptr[n] != P.l_doubles) {
            err = "nested dissection: internal error (forward layout does not match the factor size)";
            return false;
        }
        P.fidx.resize(P.l_doubles);
        for (const NdSegment &g2 : P.segs) {
            const bool leaf_src = P.nodes[P.node_of_row[g2.first]].level == 0;
            for (int t = 0; t < g2.p; ++t) P.fidx[(size_t)g2.dst + t] = leaf_src ? 2 * n + P.perm[g2.first + t] : g2.first + t;
        }
    }
    // ---- backward gather lists and the row records of both passes, level by level (long rows first)
    {
        P.bidx.resize(P.idx_ints);
        for (int k = 0; k < nn; ++k) {
            const NdNode &nd = P.nodes[k];
            int *ix = P.bidx.data() + nd.ioff;
            for (int t = 0; t < nd.np; ++t) ix[t] = nd.level == 0 ? 2 * n + P.perm[nd.first + t] : nd.first + t;
            for (int i = 0; i < nd.nu; ++i) ix[nd.np + i] = n + P.upd_idx[nd.upd + i];
        }
        P.fwd.assign((size_t)P.nlevels, NdPass());
        P.bwd.assign((size_t)P.nlevels, NdPass());
        for (int l = 0; l < P.nlevels; ++l) {
            for (int pass = 0; pass < 2; ++pass) {
                if (pass == 0 && l == 0) continue;
                NdPass &ps = pass ? P.bwd[l] : P.fwd[l];
                std::vector<NdRow> narrow;
                for (int k : P.level_nodes[l]) {
                    const NdNode &nd = P.nodes[k];
                    for (int q = 0; q < nd.np; ++q) {
                        const int r = nd.first + q;
                        NdRow R;
                        if (pass == 0) {
                            R = NdRow{P.fwd_ptr[r], P.fwd_ptr[r], (int)(P.fwd_ptr[r + 1] - P.fwd_ptr[r]), r, P.perm[r], 0};
                        } else {
                            const int len = nd.np + nd.nu;
                            R = NdRow{(long long)(nd.boff + (size_t)q * len), (long long)nd.ioff, len, r, P.perm[r], 0};
                        }
                        (R.len > P.wide_len ? ps.rows : narrow).push_back(R);
                    }
                }
                ps.nwide = (int)ps.rows.size();
                ps.rows.insert(ps.rows.end(), narrow.begin(), narrow.end());
            }
        }
    }
    return true;
}

}  // namespace sparsh
