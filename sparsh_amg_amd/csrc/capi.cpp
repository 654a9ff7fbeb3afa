// capi.cpp -- extern "C" surface declared in include/sparsh_amg.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/sparsh_amg.h"
#include "engine.hpp"

using namespace sparsh;

struct sparsh_handle_s {
    std::unique_ptr<Engine> eng;
    LocalOp cached_op;  // last sparsh_dist_local_op result
    DeepLocal cached_deep;  // last sparsh_dist_deep_op result
};

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

// end of an operator-level entry point: a fault raised inside the call wins over the copy-back status
int done(const Engine &E, bool copied_back)
{
    if (E.fault() != SPARSH_OK) return fail(E.fault(), E.error);
    return copied_back ? SPARSH_OK : fail(SPARSH_ENODEV, "D2H failed");
}

int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}

double env_dbl(const char *name, double dflt)
{
    const char *v = std::getenv(name);
    return (v && *v) ? std::atof(v) : dflt;
}

// RAII device buffer for the host-vector operator wrappers
struct DBuf {
    Engine &E;
    double *p = nullptr;
    size_t n;
    DBuf(Engine &e, size_t count, const double *src = nullptr) : E(e), n(count)
    {
        p = static_cast<double *>(E.dalloc(count * sizeof(double)));
        if (p && src) (void)hipMemcpy(p, src, count * sizeof(double), hipMemcpyHostToDevice);
    }
    ~DBuf() { E.dfree(p); }
    bool get(double *dst)
    {
        (void)hipStreamSynchronize(E.stream());
        return hipMemcpy(dst, p, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
    }
};

#define REQUIRE_READY(h)                                                         \
    if (!(h) || !(h)->eng) return fail(SPARSH_EINVAL, "null handle");            \
    if (!(h)->eng->ready()) return fail(SPARSH_ESTATE, "sparsh_setup has not been called (or failed)"); \
    if ((h)->eng->fault() != SPARSH_OK) return fail((h)->eng->fault(), (h)->eng->error) /* sticky device / transport fault */

#define REQUIRE_LEVEL(h, l) \
    if ((l) < 0 || (l) >= (int)(h)->eng->host().levels.size()) return fail(SPARSH_EINVAL, "level out of range")

#define REQUIRE_SINGLE(h) \
    if ((h)->eng->distributed()) return fail(SPARSH_ESTATE, "operator-level entry points are single-GPU test hooks (host vectors carry no halo)")

#define REQUIRE_HOST(h)                                               \
    if (!(h) || !(h)->eng) return fail(SPARSH_EINVAL, "null handle"); \
    if (!(h)->eng->host_ready()) return fail(SPARSH_ESTATE, "sparsh_setup / sparsh_setup_host has not been called (or failed)")

}  // namespace

extern "C" {

const char *sparsh_last_error(void) { return g_err.c_str(); }

int sparsh_version(void) { return 100; }

int sparsh_host_cpus(void) { return effective_cpus(); }

int sparsh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void sparsh_default_params(sparsh_params *p)
{
    if (!p) return;
    // reference macros, include/AMG.hpp:15-27
    p->omega = env_dbl("SPARSH_OMEGA", 0.66667);
    p->tol = env_dbl("SPARSH_TOL", 1e-8);
    p->sweeps = env_int("SPARSH_NU", 6 + 1);  // smooth_iter + 1: the CPU path's count (src/AMG_smoothers.cpp:59-60)
    p->max_levels = env_int("SPARSH_LEVELS", 6);
    p->limit_upper = 4000;
    p->limit_lower = 2000;
    p->coarsening = 0;
    if (const char *c = std::getenv("SPARSH_COARSENING")) p->coarsening = (std::strcmp(c, "beck") == 0) ? 1 : 0;
    p->max_iter = env_int("SPARSH_MAXIT", 100000);
    p->coarse_limit = env_int("SPARSH_COARSE_LIMIT", 40000);
    p->dense_limit = env_int("SPARSH_DENSE_LIMIT", 8192);
    p->extend_until = env_int("SPARSH_EXTEND_UNTIL", 0);
    p->coarse_factor_mb = env_int("SPARSH_COARSE_FACTOR_MB", 1024);
    p->host_threads = env_int("SPARSH_THREADS", 0);
    p->device = -1;
    if (const char *lr = std::getenv("LOCAL_RANK")) p->device = std::atoi(lr);
    const int pr = env_int("SPARSH_PRINT", 1);
    p->print_setup = pr;
    p->print_solve = pr;
    p->check_every = 1;
    p->use_graph = env_int("SPARSH_GRAPH", 0);
    p->replicate_rows = env_int("SPARSH_REPLICATE_ROWS", 0);
    p->precond_fp32 = env_int("SPARSH_PRECOND_FP32", 0);
}

int sparsh_create_csr(int nrow, int ncol, const int *rowptr, const int *colindex, const double *val, sparsh_handle *out)
{
    if (!out || !rowptr || nrow <= 0 || ncol <= 0) return fail(SPARSH_EINVAL, "bad CSR arguments");
    if (rowptr[0] != 0 || rowptr[nrow] < 0) return fail(SPARSH_EINVAL, "rowptr must start at 0");
    if (rowptr[nrow] > 0 && (!colindex || !val)) return fail(SPARSH_EINVAL, "null colindex/val");
    // one O(nnz) pass: a non-monotone rowptr or an out-of-range column would become an out-of-bounds
    // access on the device
    {
        long bad_rp = 0, bad_col = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad_rp, bad_col)
        for (int i = 0; i < nrow; ++i) {
            const int j0 = rowptr[i], j1 = rowptr[i + 1];
            if (j1 < j0 || j0 < 0 || j1 > rowptr[nrow]) {
                ++bad_rp;
                continue;
            }
            for (int j = j0; j < j1; ++j) bad_col += (colindex[j] < 0 || colindex[j] >= ncol);
        }
        if (bad_rp) return fail(SPARSH_EINVAL, "rowptr is not monotonically non-decreasing (" + std::to_string(bad_rp) + " rows)");
        if (bad_col) return fail(SPARSH_EINVAL, "colindex out of range [0, ncol) (" + std::to_string(bad_col) + " entries)");
    }
    auto *h = new sparsh_handle_s;
    h->eng.reset(new Engine(nrow, ncol, rowptr, colindex, val));
    *out = h;
    return SPARSH_OK;
}

void sparsh_destroy(sparsh_handle h) { delete h; }

int sparsh_setup(sparsh_handle h, const sparsh_params *p)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    sparsh_params prm;
    if (p)
        prm = *p;
    else
        sparsh_default_params(&prm);
    if (prm.sweeps < 1 || prm.max_levels < 1 || prm.limit_upper < 1) return fail(SPARSH_EINVAL, "bad params");
    int rc = h->eng->setup(prm);
    if (rc != SPARSH_OK) return fail(rc, h->eng->error);
    return rc;
}

int sparsh_setup_host(sparsh_handle h, const sparsh_params *p)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    sparsh_params prm;
    if (p)
        prm = *p;
    else
        sparsh_default_params(&prm);
    if (prm.sweeps < 1 || prm.max_levels < 1 || prm.limit_upper < 1) return fail(SPARSH_EINVAL, "bad params");
    int rc = h->eng->setup_host(prm);
    if (rc != SPARSH_OK) return fail(rc, h->eng->error);
    return rc;
}

int sparsh_set_stopping(sparsh_handle h, double tol, int max_iter, int check_every)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->set_stopping(tol, max_iter, check_every);
    return SPARSH_OK;
}

int sparsh_set_kernel_config(sparsh_handle h, int kind, int vec, int nt, int remap)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (kind < 0 || kind > 3) return fail(SPARSH_EINVAL, "kind must be 0 (workgroup CSR-stream), 1 (wave CSR-stream), 2 (sliced ELL) or 3 (sliced diagonals)");
    KernelConfig &c = h->eng->kernel_cfg();
    c.kind = kind;
    c.vec = vec < 0 ? 0 : (vec > 4 ? 3 : vec);
    c.nt = nt > 0;
    c.remap = remap < 0 ? 0 : remap;
    c.auto_policy = (nt < 0 || remap < 0);
    h->eng->config_changed();
    return SPARSH_OK;
}

// which kernel family the SpMV-type operators of a level run with under the current config:
// 3 sliced diagonals, 2 sliced ELL, 1 wave CSR-stream, 0 workgroup CSR-stream; bytes_per_entry =
// what that layout streams per stored entry (values + indices), slots/entries incl. padding.
int sparsh_level_format(sparsh_handle h, int level, int *kind, long *stored_entries)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    const DevCsr &A = h->eng->level(level).A;
    const CsrFamily fam = csr_family(A, h->eng->kernel_cfg());  // the launcher's own decision
    int k = 0;
    long e = A.nnz;
    if (fam == FAM_SDIA || fam == FAM_SDIA_TAB) {
        k = 3;
        e = A.sd_vblocks * 64;  // values actually stored: constant slots own no block
    } else if (fam == FAM_SELL) {
        k = 2;
        e = A.sell_entries;
    } else if (fam == FAM_CSR_WAVE) {
        k = 1;
    }
    if (kind) *kind = k;
    if (stored_entries) *stored_entries = e;
    return SPARSH_OK;
}

int sparsh_level_layout(sparsh_handle h, int level, long *slots, long *value_blocks, long *meta_bytes)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    const DevCsr &A = h->eng->level(level).A;
    if (slots) *slots = A.has_sdia() ? A.sd_slots : 0;
    if (value_blocks) *value_blocks = A.has_sdia() ? A.sd_vblocks : 0;
    if (meta_bytes) {
        // what one sweep reads besides values and vectors: per-slice records (192 B) where they exist
        // (slices off the record path additionally read their 24 B/slot headers: counted for all slots
        // only when there are no records), else sd_ptr + 24 B per slot
        if (!A.has_sdia())
            *meta_bytes = 0;
        else if (A.sd_tmask)  // table path: 8 lane masks + the conformity word per slice
            *meta_bytes = (long)A.nslice * 68 + A.sd_vblocks * 24;
        else if (A.sd_rec)
            *meta_bytes = (long)A.nslice * kSdRecInts * 4 + A.sd_vblocks * 24;
        else
            *meta_bytes = (long)A.nslice * 4 + A.sd_slots * 24;
    }
    return SPARSH_OK;
}

const char *sparsh_level_kernel(sparsh_handle h, int level)
{
    if (!h || !h->eng || !h->eng->ready() || level < 0 || level >= (int)h->eng->host().levels.size()) return "";
    return csr_family_name(csr_family(h->eng->level(level).A, h->eng->kernel_cfg()));
}

int sparsh_level_placement(sparsh_handle h, int level, int *nt, int *remap)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    bool b = false;
    int r = 0;
    csr_placement(h->eng->level(level).A, h->eng->kernel_cfg(), &b, &r);
    if (nt) *nt = b ? 1 : 0;
    if (remap) *remap = r;
    return SPARSH_OK;
}

int sparsh_bench_comm(sparsh_handle h, int what, int level, int reps, double *avg_seconds)
{
    REQUIRE_READY(h);
    if (!avg_seconds) return fail(SPARSH_EINVAL, "null output");
    *avg_seconds = h->eng->bench_comm(what, level, reps);
    return SPARSH_OK;
}

int sparsh_set_tile(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->kernel_cfg().tile = enable != 0;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_level_tile_rows(sparsh_handle h, int level, int *rows)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    if (rows) *rows = csr_family(h->eng->level(level).A, h->eng->kernel_cfg()) == FAM_SDIA_TAB ? sdia_tile_rows(h->eng->level(level).A, h->eng->kernel_cfg()) : 0;
    return SPARSH_OK;
}

int sparsh_set_const_slots(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->kernel_cfg().const_slots = enable != 0;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_set_setup_broadcast(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->set_share_setup(enable != 0);
    return SPARSH_OK;
}

int sparsh_setup_share_info(sparsh_handle h, int *built_locally, long *image_bytes)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (built_locally) *built_locally = h->eng->built_locally() ? 1 : 0;
    if (image_bytes) *image_bytes = (long)h->eng->shared_image_bytes();
    return SPARSH_OK;
}

// test hook (host only): byte image of the hierarchy -> fresh hierarchy -> compare every array; returns the image size
long sparsh_debug_hierarchy_roundtrip(sparsh_handle h, long truncate_to)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (!h->eng->host_ready()) return fail(SPARSH_ESTATE, "sparsh_setup_host has not been called");
    const HostHierarchy &H = h->eng->host();
    std::vector<char> img;
    serialize_hierarchy(H, img);
    HostHierarchy G;
    const size_t use = truncate_to >= 0 ? std::min<size_t>((size_t)truncate_to, img.size()) : img.size();
    if (!deserialize_hierarchy(img.data(), use, H.levels[0].A, G)) return fail(SPARSH_EINVAL, G.error);
    const auto same_csr = [](const HostCsr &a, const HostCsr &b) {
        if (a.nrow != b.nrow || a.ncol != b.ncol || a.nnz() != b.nnz()) return false;
        if (!a.rowptr) return !b.rowptr;
        const size_t nnz = (size_t)a.nnz();
        return std::memcmp(a.rowptr, b.rowptr, ((size_t)a.nrow + 1) * sizeof(int)) == 0 && std::memcmp(a.col, b.col, nnz * sizeof(int)) == 0 &&
               std::memcmp(a.val, b.val, nnz * sizeof(double)) == 0;
    };
    bool same = G.levels.size() == H.levels.size() && G.nL == H.nL && G.coarse_dense == H.coarse_dense && G.extended == H.extended &&
                G.coarse_inverse == H.coarse_inverse;
    for (size_t l = 0; same && l < H.levels.size(); ++l)
        same = same_csr(G.levels[l].A, H.levels[l].A) && same_csr(G.levels[l].P, H.levels[l].P) && same_csr(G.levels[l].R, H.levels[l].R) &&
               G.levels[l].diag == H.levels[l].diag && G.levels[l].P_is_aggregation == H.levels[l].P_is_aggregation;
    if (!same) return fail(SPARSH_ENUMERIC, "hierarchy image round trip changed the hierarchy");
    return (long)img.size();
}

// test hook (host only): row-block schedule -> 16-bit delta form -> decoded columns, compared with the input
int sparsh_debug_index16_roundtrip(int nrow, const int *rowptr, const int *col, long *blocks16, long *blocks)
{
    if (nrow < 0 || !rowptr || (!col && rowptr[nrow] > 0)) return fail(SPARSH_EINVAL, "bad arguments");
    int nblk = 0;
    const std::vector<int> rec = rowblock_records(nrow, rowptr, &nblk);
    std::vector<unsigned short> c16((size_t)rowptr[nrow] + kCsrPad, 0);
    std::vector<int> cb((size_t)std::max(nblk, 1), -1);
    const int n16 = build_col16(rowptr, col, rec.data(), nblk, c16.data(), cb.data());
    int counted = 0;
    for (int k = 0; k < nblk; ++k) {
        if (cb[k] < 0) continue;
        ++counted;
        for (int r = rec[(size_t)4 * k]; r < rec[(size_t)4 * k + 1]; ++r) {
            int c = cb[k];
            for (int j = rowptr[r]; j < rowptr[r + 1]; ++j) {
                c += c16[j];
                if (c != col[j]) return fail(SPARSH_ENUMERIC, "16-bit delta form decodes to a different column");
            }
        }
    }
    if (counted != n16) return fail(SPARSH_ENUMERIC, "block count mismatch");
    if (blocks16) *blocks16 = n16;
    if (blocks) *blocks = nblk;
    return SPARSH_OK;
}

int sparsh_set_placement_search(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->kernel_cfg().place_search = enable != 0;
    return SPARSH_OK;
}

int sparsh_placement_info(sparsh_handle h, double *chosen_us, double *worst_us, double *initial_us, int *triples, double *seconds)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (chosen_us) *chosen_us = h->eng->place_best_us;
    if (worst_us) *worst_us = h->eng->place_worst_us;
    if (initial_us) *initial_us = h->eng->place_first_us;
    if (triples) *triples = h->eng->place_tried;
    if (seconds) *seconds = h->eng->place_seconds;
    return SPARSH_OK;
}

int sparsh_set_fused_zero_sweep(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->kernel_cfg().fuse_cg_zero = enable != 0;
    h->eng->kernel_cfg().cg_nt = enable == 2;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_set_alternate_sweeps(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (enable < 0 || enable > 2) return fail(SPARSH_EINVAL, "mode must be 0 (never), 1 (where a sweep streams more than the Infinity Cache holds) or 2 (always)");
    h->eng->kernel_cfg().alt_dir = enable;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_set_double_sweep(sparsh_handle h, int mode)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (mode < 0 || mode > 2) return fail(SPARSH_EINVAL, "mode must be 0 (never), 1 (where the setup measures it faster) or 2 (wherever the level is a box grid with a plan)");
    h->eng->kernel_cfg().box2 = mode;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_level_double_sweep(sparsh_handle h, int level, int *on, int *dims, int *plan, double *single_us, double *double_us)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    const DevLevel &L = h->eng->level(level);
    if (on) *on = (!h->eng->distributed() || L.replicated) && box2_applies(L.A, h->eng->kernel_cfg()) ? 1 : 0;
    if (dims) {
        dims[0] = L.A.box_nx;
        dims[1] = L.A.box_ny;
        dims[2] = L.A.box_nz;
    }
    if (plan) {
        plan[0] = L.A.box_q;
        plan[1] = L.A.box_ty;
        plan[2] = L.A.box_cz;
    }
    if (single_us) *single_us = L.box_single_us;
    if (double_us) *double_us = L.box_double_us;
    return SPARSH_OK;
}

int sparsh_set_deferred_x(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->kernel_cfg().defer_x = enable != 0;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_set_zero_start(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->kernel_cfg().zero_start = enable != 0;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_set_marching_ops(sparsh_handle h, int mode)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (mode < 0 || mode > 2) return fail(SPARSH_EINVAL, "mode must be 0 (never), 1 (where the setup measures it faster) or 2 (wherever the level is a box grid with a plan)");
    h->eng->kernel_cfg().box1 = mode;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_level_marching_ops(sparsh_handle h, int level, int *on, int *plan, double *table_us, double *marching_us)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    const DevLevel &L = h->eng->level(level);
    if (on) *on = (!h->eng->distributed() || L.replicated) && box1_applies(L.A, h->eng->kernel_cfg()) ? 1 : 0;
    if (plan) {
        plan[0] = L.A.box1_q;
        plan[1] = L.A.box1_ty;
        plan[2] = L.A.box1_cz;
    }
    if (table_us) *table_us = L.box1_table_us;
    if (marching_us) *marching_us = L.box1_us;
    return SPARSH_OK;
}

int sparsh_set_constant_diagonal(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->kernel_cfg().const_diag = enable != 0;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_level_constant_diagonal(sparsh_handle h, int level, int *is_const, double *value)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    const DevLevel &L = h->eng->level(level);
    if (is_const) *is_const = h->eng->diag_stream(L) == nullptr ? 1 : 0;
    if (value) *value = L.diag_const;
    return SPARSH_OK;
}

int sparsh_set_fused_prolongation(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->kernel_cfg().fuse_prolong = enable != 0;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_level_prolong_fused(sparsh_handle h, int level, int *fused)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    if (!fused) return fail(SPARSH_EINVAL, "null output");
    *fused = !h->eng->level_prolong_fused(level) ? 0 : (h->eng->level(level - 1).pair_aggregates ? 1 : 2);
    return SPARSH_OK;
}

int sparsh_set_paired_restriction(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->kernel_cfg().pair_restrict = enable != 0;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_level_paired(sparsh_handle h, int level, int *paired)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    if (!paired) return fail(SPARSH_EINVAL, "null output");
    *paired = h->eng->level_paired(level);
    return SPARSH_OK;
}

int sparsh_set_index_compression(sparsh_handle h, int mode)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (mode < 0 || mode > 2) return fail(SPARSH_EINVAL, "mode must be 0 (off), 1 (operators that stream from HBM through the CSR-stream kernel) or 2 (every operator)");
    h->eng->kernel_cfg().idx16 = mode;
    h->eng->config_changed();
    return SPARSH_OK;
}

int sparsh_level_index16(sparsh_handle h, int level, long *blocks16, long *blocks)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    const DevCsr &A = h->eng->level(level).A;
    if (blocks16) *blocks16 = A.col16 ? A.nblk16 : 0;
    if (blocks) *blocks = A.nblk;
    return SPARSH_OK;
}

int sparsh_num_levels(sparsh_handle h)
{
    if (!h || !h->eng || !h->eng->host_ready()) return 0;
    return (int)h->eng->host().levels.size();
}

int sparsh_level_info(sparsh_handle h, int level, int *nrow, int *nnz, int *p_ncol, int *p_nnz)
{
    REQUIRE_HOST(h);
    REQUIRE_LEVEL(h, level);
    const HostLevel &L = h->eng->host().levels[level];
    if (nrow) *nrow = L.A.nrow;
    if (nnz) *nnz = L.A.nnz();
    if (p_ncol) *p_ncol = L.P.rowptr ? L.P.ncol : 0;
    if (p_nnz) *p_nnz = L.P.rowptr ? L.P.nnz() : 0;
    return SPARSH_OK;
}

int sparsh_level_csr(sparsh_handle h, int level, int which, int *rowptr, int *colindex, double *val)
{
    REQUIRE_HOST(h);
    REQUIRE_LEVEL(h, level);
    const HostLevel &L = h->eng->host().levels[level];
    const HostCsr &M = (which == 0) ? L.A : L.P;
    if (!M.rowptr) return fail(SPARSH_EINVAL, "no such matrix on this level");
    std::memcpy(rowptr, M.rowptr, sizeof(int) * ((size_t)M.nrow + 1));
    std::memcpy(colindex, M.col, sizeof(int) * (size_t)M.nnz());
    std::memcpy(val, M.val, sizeof(double) * (size_t)M.nnz());
    return SPARSH_OK;
}

int sparsh_coarse_inverse(sparsh_handle h, double *inv)
{
    REQUIRE_HOST(h);
    const HostHierarchy &H = h->eng->host();
    if (!H.coarse_dense) return fail(SPARSH_ESTATE, "the coarsest level is above dense_limit: it is factored on the device (nested-dissection or block-tridiagonal form), no dense inverse exists");
    if (H.coarse_inverse.empty()) return fail(SPARSH_ESTATE, "host copy of the inverse was released by sparsh_setup; use sparsh_setup_host");
    std::memcpy(inv, H.coarse_inverse.data(), sizeof(double) * (size_t)H.nL * H.nL);
    return SPARSH_OK;
}

int sparsh_coarse_info(sparsh_handle h, int *info6, long *bytes)
{
    REQUIRE_HOST(h);
    const HostHierarchy &H = h->eng->host();
    const CoarseSolver &c = h->eng->coarse();
    const bool bt = c.ready() && !c.dense() && !c.nested();
    if (info6) {
        info6[0] = H.nL;
        info6[1] = H.coarse_dense ? 1 : 0;
        info6[2] = bt ? c.block() : 0;
        info6[3] = bt ? c.nblocks() : 0;
        info6[4] = bt ? c.bandwidth() : 0;
        info6[5] = H.extended ? 1 : 0;
    }
    if (bytes) *bytes = c.ready() ? (long)c.bytes() : 0;
    return SPARSH_OK;
}

int sparsh_set_coarse_interface(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->coarse_mut().set_allow_windowed(enable != 0);
    h->eng->coarse_mut().set_unrolled_chain(enable != 2);
    return SPARSH_OK;
}

int sparsh_set_coarse_form(sparsh_handle h, int form, int leaf, int merge_rows)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (form != 0 && form != 1) return fail(SPARSH_EINVAL, "form must be 0 (nested dissection) or 1 (block tridiagonal)");
    h->eng->coarse_mut().set_form(form);
    h->eng->coarse_mut().set_nd_params(leaf, merge_rows);
    return SPARSH_OK;
}

int sparsh_set_coarse_top_merge(sparsh_handle h, int top_merge_rows)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (top_merge_rows < 0) return fail(SPARSH_EINVAL, "top_merge_rows must be >= 0");
    h->eng->coarse_mut().set_nd_params(0, -1, top_merge_rows);
    return SPARSH_OK;
}

int sparsh_coarse_nd_info(sparsh_handle h, int *info6)
{
    REQUIRE_HOST(h);
    const CoarseSolver &c = h->eng->coarse();
    const bool nd = c.ready() && c.nested();
    if (info6) {
        info6[0] = nd ? 1 : 0;
        info6[1] = nd ? c.nd().nnodes() : 0;
        info6[2] = nd ? c.nd().nlevels() : 0;
        info6[3] = nd ? c.nd().max_pivot_rows() : 0;
        info6[4] = nd ? c.nd().launches_per_solve() : 0;
        info6[5] = nd ? c.nd().leaf() : 0;
    }
    return SPARSH_OK;
}

int sparsh_set_coarse_block(sparsh_handle h, int rows)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (rows < 0) return fail(SPARSH_EINVAL, "rows must be >= 0 (0 = built-in rule)");
    h->eng->coarse_mut().set_block_hint(rows);
    return SPARSH_OK;
}

int sparsh_coarse_window(sparsh_handle h, int *window)
{
    REQUIRE_HOST(h);
    const CoarseSolver &c = h->eng->coarse();
    if (window) *window = (c.ready() && !c.dense() && c.windowed()) ? c.window() : 0;
    return SPARSH_OK;
}

double sparsh_setup_seconds(sparsh_handle h) { return (h && h->eng) ? h->eng->setup_seconds : 0.0; }

int sparsh_vcycle(sparsh_handle h, const double *b, double *x, int iterations, double *hist, int hist_cap, int *ncycles)
{
    REQUIRE_READY(h);
    Engine &E = *h->eng;
    const size_t n = (size_t)E.local_n0();  // multi-GPU: this rank's block of level 0
    DBuf db(E, n, b), dx(E, n, x);
    if (!db.p || !dx.p) return fail(SPARSH_ENODEV, E.error);
    int rc = E.amg_solve_dev(db.p, dx.p, iterations, hist, hist_cap, ncycles);
    dx.get(x);
    if (rc != SPARSH_OK) return fail(rc, E.error);
    return rc;
}

int sparsh_vcycle_dev(sparsh_handle h, const double *b_dev, double *x_dev, int iterations, double *hist, int hist_cap, int *ncycles)
{
    REQUIRE_READY(h);
    int rc = h->eng->amg_solve_dev(b_dev, x_dev, iterations, hist, hist_cap, ncycles);
    if (rc != SPARSH_OK) return fail(rc, h->eng->error);
    return rc;
}

int sparsh_solve(sparsh_handle h, int method, const double *b, double *x, double *hist, int hist_cap, int *iters)
{
    REQUIRE_READY(h);
    Engine &E = *h->eng;
    const size_t n = (size_t)E.local_n0();  // multi-GPU: this rank's block of level 0
    DBuf db(E, n, b), dx(E, n, x);
    if (!db.p || !dx.p) return fail(SPARSH_ENODEV, E.error);
    int rc = E.solve_dev(method, db.p, dx.p, 0, hist, hist_cap, iters, nullptr);
    dx.get(x);
    if (rc != SPARSH_OK) return fail(rc, E.error);
    return rc;
}

int sparsh_solve_dev(sparsh_handle h, int method, const double *b_dev, double *x_dev, int max_iters, double *hist, int hist_cap,
                     int *iters, double *seconds)
{
    REQUIRE_READY(h);
    int rc = h->eng->solve_dev(method, b_dev, x_dev, max_iters, hist, hist_cap, iters, seconds);
    if (rc != SPARSH_OK) return fail(rc, h->eng->error);
    return rc;
}

// ---- multi-GPU ----

int sparsh_set_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(SPARSH_ENODEV, "no HIP device visible");
    if (hipSetDevice(device % n) != hipSuccess) return fail(SPARSH_ENODEV, "hipSetDevice failed");
    return SPARSH_OK;
}

int sparsh_comm_unique_id(char id128[128])
{
    std::string err;
    if (!rccl_unique_id(id128, err)) return fail(SPARSH_ECOMM, err);
    return SPARSH_OK;
}

int sparsh_comm_init_rccl(sparsh_handle h, const char id128[128], int rank, int nranks)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (rank < 0 || rank >= nranks) return fail(SPARSH_EINVAL, "rank out of range");
    std::string err;
    auto c = make_rccl_comm(id128, rank, nranks, err);
    if (!c) return fail(SPARSH_ECOMM, err);
    h->eng->set_comm(std::move(c));
    return SPARSH_OK;
}

int sparsh_comm_group_create(int nranks, void **group)
{
    if (nranks < 1 || !group) return fail(SPARSH_EINVAL, "bad arguments");
    *group = thread_group_create(nranks);
    return SPARSH_OK;
}

void sparsh_comm_group_destroy(void *group) { thread_group_destroy(static_cast<ThreadGroup *>(group)); }

int sparsh_comm_group_set_delay(void *group, double microseconds)
{
    if (!group || microseconds < 0.0) return fail(SPARSH_EINVAL, "bad group / delay");
    thread_group_set_delay(static_cast<ThreadGroup *>(group), microseconds);
    return SPARSH_OK;
}

int sparsh_set_comm_tuning(sparsh_handle h, int mode)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (mode != 0 && mode != 1) return fail(SPARSH_EINVAL, "mode must be 0 (replicate_rows / set_deep_halo decide) or 1 (measured)");
    h->eng->set_comm_tuning(mode);
    return SPARSH_OK;
}

int sparsh_plan_comm_schedule(sparsh_handle h, int nranks, const double *m7)
{
    REQUIRE_HOST(h);
    if (nranks < 2 || !m7) return fail(SPARSH_EINVAL, "nranks >= 2 and seven measured numbers are needed");
    Engine::CommMeasured m;
    m.exchange_us = m7[0];
    m.exchange_us_per_mb = m7[1];
    m.allreduce_us = m7[2];
    m.allgather_us = m7[3];
    m.allgather_us_per_mb = m7[4];
    m.sweep_floor_us = m7[5];
    m.sweep_us_per_mb = m7[6];
    h->eng->plan_comm_schedule(h->eng->params(), nranks, m);
    return SPARSH_OK;
}

int sparsh_comm_schedule(sparsh_handle h, int level, int *info4, double *cost_us3)
{
    REQUIRE_HOST(h);
    if (level < 0 || level >= (int)h->eng->host().levels.size()) return fail(SPARSH_EINVAL, "level out of range");
    const auto &sc = h->eng->comm_schedule();
    if (sc.empty()) return fail(SPARSH_ESTATE, "no measured schedule: one rank, tuning off, or replicate_rows given by the caller");
    const auto &c = sc[level];
    if (info4) {
        info4[0] = c.rows;
        info4[1] = c.halo_rows;
        info4[2] = c.partitioned ? 1 : 0;
        info4[3] = c.deep ? 1 : 0;
    }
    if (cost_us3) {
        cost_us3[0] = c.cost_deep_us;
        cost_us3[1] = c.cost_per_sweep_us;
        cost_us3[2] = c.cost_replicated_us;
    }
    return SPARSH_OK;
}

int sparsh_comm_measured(sparsh_handle h, double *m7)
{
    REQUIRE_HOST(h);
    const auto &m = h->eng->comm_measured();
    if (!m.valid) return fail(SPARSH_ESTATE, "the transport was not measured in this setup");
    if (m7) {
        m7[0] = m.exchange_us;
        m7[1] = m.exchange_us_per_mb;
        m7[2] = m.allreduce_us;
        m7[3] = m.allgather_us;
        m7[4] = m.allgather_us_per_mb;
        m7[5] = m.sweep_floor_us;
        m7[6] = m.sweep_us_per_mb;
    }
    return SPARSH_OK;
}

int sparsh_comm_group_fail_after(void *group, int ncalls)
{
    if (!group) return fail(SPARSH_EINVAL, "null group");
    thread_group_fail_after(static_cast<ThreadGroup *>(group), ncalls);
    return SPARSH_OK;
}

int sparsh_comm_init_group(sparsh_handle h, void *group, int rank)
{
    if (!h || !h->eng || !group) return fail(SPARSH_EINVAL, "bad arguments");
    h->eng->set_comm(make_thread_comm(static_cast<ThreadGroup *>(group), rank));
    return SPARSH_OK;
}

int sparsh_deep_info(sparsh_handle h, int level, int *info4)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    const DevLevel &L = h->eng->level(level);
    info4[0] = L.deep ? L.K : 0;
    info4[1] = L.deep ? L.A.nrow : 0;
    info4[2] = L.deep ? L.A.ncol : 0;
    info4[3] = L.deep ? L.npad : 0;
    return SPARSH_OK;
}

int sparsh_deep_layer_end(sparsh_handle h, int level, int d, int *end)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    const DevLevel &L = h->eng->level(level);
    if (!L.deep || d < 0 || d > L.K) return fail(SPARSH_EINVAL, "not a deep-halo level / layer out of range");
    *end = L.layer_end[d];
    return SPARSH_OK;
}

int sparsh_deep_prefix_spmv(sparsh_handle h, int level, int rows, const double *x_ext, double *y)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    Engine &E = *h->eng;
    const DevLevel &L = E.level(level);
    if (!L.deep || rows < 0 || rows > L.A.nrow) return fail(SPARSH_EINVAL, "not a deep-halo level / bad prefix");
    DBuf dx(E, (size_t)L.A.ncol, x_ext), dy(E, (size_t)L.A.nrow + 64);
    if (!E.debug_prefix_spmv(level, rows, dx.p, dy.p)) return fail(SPARSH_ENODEV, E.error);
    if (hipMemcpy(y, dy.p, (size_t)rows * 8, hipMemcpyDeviceToHost) != hipSuccess) return fail(SPARSH_ENODEV, "D2H failed");
    return SPARSH_OK;
}

int sparsh_set_deep_halo(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->set_deep_halo(enable != 0);
    return SPARSH_OK;
}

long sparsh_exchanges_issued(sparsh_handle h) { return (h && h->eng) ? h->eng->exchanges_issued() : 0; }

int sparsh_set_overlap(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->set_overlap(enable != 0);
    return SPARSH_OK;
}

int sparsh_local_range(sparsh_handle h, int level, int *lo, int *hi, int *replicated)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    Engine &E = *h->eng;
    const Partition &p = E.partition(level);
    const int r = E.comm() ? E.comm()->rank : 0;
    if (lo) *lo = p.replicated ? 0 : p.lo(r);
    if (hi) *hi = p.replicated ? p.n : p.hi(r);
    if (replicated) *replicated = p.replicated ? 1 : 0;
    return SPARSH_OK;
}

// Host-only planning query (needs sparsh_setup_host): the part of operator `which` (0 A_l, 1 P_l,
// 2 R_l) that `rank` of `nranks` holds when no level is replicated, and its halo plan.
int sparsh_dist_local_op(sparsh_handle h, int level, int which, int rank, int nranks, int *sizes8)
{
    REQUIRE_HOST(h);
    REQUIRE_LEVEL(h, level);
    if (nranks < 1 || rank < 0 || rank >= nranks || which < 0 || which > 2) return fail(SPARSH_EINVAL, "bad arguments");
    const HostHierarchy &H = h->eng->host();
    const int nl = (int)H.levels.size();
    if (which != 0 && level + 1 >= nl) return fail(SPARSH_EINVAL, "no coarser level");
    std::vector<Partition> parts((size_t)nl);
    for (int l = 0; l <= std::min(level + 1, nl - 1); ++l)
        parts[l] = (l == 0) ? make_partition(H.levels[0].A.nrow, nranks) : coarse_partition(H.levels[l - 1].R, parts[l - 1]);
    const HostLevel &L = H.levels[level];
    if (which == 0)
        h->cached_op = extract_local(L.A, parts[level], parts[level], rank);
    else if (which == 1)
        h->cached_op = extract_local(L.P, parts[level], parts[level + 1], rank);
    else
        h->cached_op = extract_local(L.R, parts[level + 1], parts[level], rank);
    const LocalOp &o = h->cached_op;
    const Partition &rowsP = (which == 2) ? parts[level + 1] : parts[level];
    sizes8[0] = o.M.nrow;
    sizes8[1] = o.M.nnz();
    sizes8[2] = o.plan.nloc;
    sizes8[3] = o.plan.nhalo;
    sizes8[4] = (int)o.plan.send.size();
    sizes8[5] = (int)o.plan.recv.size();
    sizes8[6] = (int)o.plan.send_idx.size();
    sizes8[7] = rowsP.lo(rank);
    return SPARSH_OK;
}

int sparsh_dist_local_op_get(sparsh_handle h, int *rowptr, int *col, double *val, int *halo_global, int *send_idx, int *send_segs3,
                             int *recv_segs3)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    const LocalOp &o = h->cached_op;
    if (!o.M.rowptr) return fail(SPARSH_ESTATE, "call sparsh_dist_local_op first");
    std::memcpy(rowptr, o.M.rowptr, sizeof(int) * ((size_t)o.M.nrow + 1));
    std::memcpy(col, o.M.col, sizeof(int) * (size_t)o.M.nnz());
    std::memcpy(val, o.M.val, sizeof(double) * (size_t)o.M.nnz());
    std::copy(o.plan.halo_global.begin(), o.plan.halo_global.end(), halo_global);
    std::copy(o.plan.send_idx.begin(), o.plan.send_idx.end(), send_idx);
    for (size_t k = 0; k < o.plan.send.size(); ++k) {
        send_segs3[3 * k] = o.plan.send[k].peer;
        send_segs3[3 * k + 1] = o.plan.send[k].off;
        send_segs3[3 * k + 2] = o.plan.send[k].cnt;
    }
    for (size_t k = 0; k < o.plan.recv.size(); ++k) {
        recv_segs3[3 * k] = o.plan.recv[k].peer;
        recv_segs3[3 * k + 1] = o.plan.recv[k].off;
        recv_segs3[3 * k + 2] = o.plan.recv[k].cnt;
    }
    return SPARSH_OK;
}

int sparsh_dist_deep_op(sparsh_handle h, int level, int rank, int nranks, int K, int depth, int *sizes8)
{
    REQUIRE_HOST(h);
    REQUIRE_LEVEL(h, level);
    if (nranks < 2 || rank < 0 || rank >= nranks || K < 1 || depth < 1 || depth > K) return fail(SPARSH_EINVAL, "bad arguments");
    const HostHierarchy &H = h->eng->host();
    std::vector<Partition> parts((size_t)level + 1);
    for (int l = 0; l <= level; ++l)
        parts[l] = (l == 0) ? make_partition(H.levels[0].A.nrow, nranks) : coarse_partition(H.levels[l - 1].R, parts[l - 1]);
    h->cached_deep = extract_local_deep(H.levels[level].A, parts[level], rank, K, {depth});
    const DeepLocal &d = h->cached_deep;
    sizes8[0] = d.M.nrow;
    sizes8[1] = d.M.nnz();
    sizes8[2] = d.nloc;
    sizes8[3] = d.npad;
    sizes8[4] = (int)d.global_of.size();
    sizes8[5] = (int)d.plans[0].send.size();
    sizes8[6] = (int)d.plans[0].recv.size();
    sizes8[7] = (int)d.plans[0].send_idx.size();
    return SPARSH_OK;
}

int sparsh_dist_deep_op_get(sparsh_handle h, int *rowptr, int *col, double *val, int *global_of, int *layer_end, int *send_idx,
                            int *send_segs3, int *recv_pos, int *recv_segs3)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    const DeepLocal &d = h->cached_deep;
    if (!d.M.rowptr) return fail(SPARSH_ESTATE, "call sparsh_dist_deep_op first");
    const DeepPlan &pl = d.plans[0];
    std::memcpy(rowptr, d.M.rowptr, sizeof(int) * ((size_t)d.M.nrow + 1));
    std::memcpy(col, d.M.col, sizeof(int) * (size_t)d.M.nnz());
    std::memcpy(val, d.M.val, sizeof(double) * (size_t)d.M.nnz());
    std::copy(d.global_of.begin(), d.global_of.end(), global_of);
    std::copy(d.layer_end.begin(), d.layer_end.end(), layer_end);
    std::copy(pl.send_idx.begin(), pl.send_idx.end(), send_idx);
    std::copy(pl.recv_pos.begin(), pl.recv_pos.end(), recv_pos);
    for (size_t k = 0; k < pl.send.size(); ++k) {
        send_segs3[3 * k] = pl.send[k].peer;
        send_segs3[3 * k + 1] = pl.send[k].off;
        send_segs3[3 * k + 2] = pl.send[k].cnt;
    }
    for (size_t k = 0; k < pl.recv.size(); ++k) {
        recv_segs3[3 * k] = pl.recv[k].peer;
        recv_segs3[3 * k + 1] = pl.recv[k].off;
        recv_segs3[3 * k + 2] = pl.recv[k].cnt;
    }
    return SPARSH_OK;
}

int sparsh_krylov_init_dev(sparsh_handle h, int method, const double *b_dev, double *x_dev)
{
    REQUIRE_READY(h);
    if (method != SPARSH_CG && method != SPARSH_PCG) return fail(SPARSH_EINVAL, "stepwise interface covers SPARSH_CG and SPARSH_PCG");
    int rc = h->eng->pcg_init(b_dev, x_dev, method == SPARSH_PCG);
    if (rc != SPARSH_OK) return fail(rc, h->eng->error);
    return rc;
}

int sparsh_krylov_step_dev(sparsh_handle h, int nsteps, int *done, double *residual)
{
    REQUIRE_READY(h);
    int rc = h->eng->pcg_steps(nsteps, done);
    if (residual) *residual = h->eng->krylov_residual();
    if (rc != SPARSH_OK) return fail(rc, h->eng->error);
    return rc;
}

int sparsh_krylov_history(sparsh_handle h, double *hist, int hist_cap, int *iters)
{
    REQUIRE_READY(h);
    int c = h->eng->krylov_hist(hist, hist_cap);
    if (iters) *iters = c;
    return SPARSH_OK;
}

int sparsh_op_spmv(sparsh_handle h, int level, const double *x, double *y)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    REQUIRE_LEVEL(h, level);
    Engine &E = *h->eng;
    const DevLevel &L = E.level(level);
    DBuf dx(E, (size_t)L.A.ncol, x), dy(E, (size_t)L.n);
    E.op_spmv(level, dx.p, dy.p);
    return done(E, dy.get(y));
}

int sparsh_op_jacobi(sparsh_handle h, int level, const double *b, double *x, int sweeps, int x_is_zero)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    REQUIRE_LEVEL(h, level);
    if (sweeps < 0) return fail(SPARSH_EINVAL, "sweeps < 0");
    Engine &E = *h->eng;
    const size_t n = (size_t)E.level(level).n;
    DBuf db(E, n, b), dx(E, n, x), dt(E, n);
    E.op_jacobi(level, db.p, dx.p, dt.p, sweeps, x_is_zero != 0);
    return done(E, dx.get(x));
}

int sparsh_op_residual(sparsh_handle h, int level, const double *b, const double *x, double *r)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    REQUIRE_LEVEL(h, level);
    Engine &E = *h->eng;
    const size_t n = (size_t)E.level(level).n;
    DBuf db(E, n, b), dx(E, n, x), dr(E, n);
    E.op_residual(level, db.p, dx.p, dr.p);
    return done(E, dr.get(r));
}

int sparsh_op_resnorm(sparsh_handle h, int level, const double *b, const double *x, double *nrm)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    REQUIRE_LEVEL(h, level);
    Engine &E = *h->eng;
    const size_t n = (size_t)E.level(level).n;
    DBuf db(E, n, b), dx(E, n, x);
    *nrm = E.op_resnorm(level, db.p, dx.p);
    return SPARSH_OK;
}

int sparsh_op_restrict(sparsh_handle h, int level, const double *r, double *bc)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    REQUIRE_LEVEL(h, level);
    if (level + 1 >= h->eng->nlevels()) return fail(SPARSH_EINVAL, "no coarser level");
    Engine &E = *h->eng;
    DBuf dr(E, (size_t)E.level(level).n, r), dc(E, (size_t)E.level(level + 1).n);
    E.op_restrict(level, dr.p, dc.p);
    return done(E, dc.get(bc));
}

int sparsh_op_residual_restrict(sparsh_handle h, int level, const double *b, const double *x, double *bc, double *xc)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    REQUIRE_LEVEL(h, level);
    Engine &E = *h->eng;
    if (!E.level_paired(level)) return fail(SPARSH_ESTATE, "the level does not take the fused residual + restriction launch (sparsh_level_paired)");
    DBuf db(E, (size_t)E.level(level).n, b), dx(E, (size_t)E.level(level).A.ncol, x);
    DBuf dc(E, (size_t)E.level(level + 1).n), dz(E, (size_t)E.level(level + 1).n);
    E.op_residual_restrict(level, db.p, dx.p, dc.p, dz.p);
    const int rc = done(E, dc.get(bc));
    return rc != SPARSH_OK ? rc : done(E, dz.get(xc));
}

int sparsh_op_jacobi_prolong(sparsh_handle h, int level, const double *b, const double *x, double *xf)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    REQUIRE_LEVEL(h, level);
    Engine &E = *h->eng;
    if (!E.level_prolong_fused(level)) return fail(SPARSH_ESTATE, "the level's last post-sweep does not prolongate itself (sparsh_level_prolong_fused)");
    DBuf db(E, (size_t)E.level(level).n, b), dx(E, (size_t)E.level(level).A.ncol, x), df(E, (size_t)E.level(level - 1).n, xf);
    E.op_jacobi_prolong(level, db.p, dx.p, df.p);
    return done(E, df.get(xf));
}

int sparsh_op_prolong(sparsh_handle h, int level, const double *xc, double *xf)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    REQUIRE_LEVEL(h, level);
    if (level + 1 >= h->eng->nlevels()) return fail(SPARSH_EINVAL, "no coarser level");
    Engine &E = *h->eng;
    DBuf dc(E, (size_t)E.level(level + 1).n, xc), df(E, (size_t)E.level(level).n, xf);
    E.op_prolong(level, dc.p, df.p);
    return done(E, df.get(xf));
}

int sparsh_op_coarse(sparsh_handle h, const double *b, double *x)
{
    REQUIRE_READY(h);
    Engine &E = *h->eng;
    const size_t n = (size_t)E.level(E.nlevels() - 1).n;
    DBuf db(E, n, b), dx(E, n);
    E.op_coarse(db.p, dx.p);
    return done(E, dx.get(x));
}

int sparsh_op_precond_f32(sparsh_handle h, const double *r, double *z)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    Engine &E = *h->eng;
    const size_t n = (size_t)E.level(0).n;
    DBuf dr(E, n, r), dz(E, n);
    if (!E.op_precond_f32(dr.p, dz.p)) return fail(SPARSH_ESTATE, E.error);
    return done(E, dz.get(z));
}

int sparsh_op_dot(sparsh_handle h, int n, const double *x, const double *y, double *out)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    if (n <= 0) return fail(SPARSH_EINVAL, "n <= 0");
    Engine &E = *h->eng;
    DBuf dx(E, (size_t)n, x), dy(E, (size_t)n, y);
    *out = E.op_dot(n, dx.p, dy.p);
    return SPARSH_OK;
}

int sparsh_op_nrm2(sparsh_handle h, int n, const double *x, double *out)
{
    double d = 0.0;
    int rc = sparsh_op_dot(h, n, x, x, &d);
    if (rc == SPARSH_OK) *out = std::sqrt(d);
    return rc;
}

int sparsh_op_axpby(sparsh_handle h, int n, double a, const double *x, double bcoef, double *y)
{
    REQUIRE_READY(h);
    REQUIRE_SINGLE(h);
    if (n <= 0) return fail(SPARSH_EINVAL, "n <= 0");
    Engine &E = *h->eng;
    DBuf dx(E, (size_t)n, x), dy(E, (size_t)n, y);
    launch_axpby(n, a, dx.p, bcoef, dy.p, E.stream());
    return done(E, dy.get(y));
}

int sparsh_bench_op(sparsh_handle h, int op, int level, int reps, double *avg_seconds)
{
    REQUIRE_READY(h);
    REQUIRE_LEVEL(h, level);
    if (reps <= 0 || !avg_seconds) return fail(SPARSH_EINVAL, "bad reps");
    Engine &E = *h->eng;
    const DevLevel &L = E.level(level);
    const size_t n = (size_t)L.n;
    const bool has_coarse = level + 1 < E.nlevels();
    if ((op == 3 || op == 4) && !has_coarse) return fail(SPARSH_EINVAL, "no coarser level");
    const size_t nc = has_coarse ? (size_t)E.level(level + 1).n : 1;
    // multi-GPU: operator inputs carry the halo behind the own entries
    const size_t xh = (size_t)std::max(L.planA.nhalo, L.planR.nhalo), ch = (size_t)L.planP.nhalo;
    DBuf x(E, n + xh), y(E, n), b(E, n), c(E, nc + ch);
    std::vector<double> ones(std::max(n, nc), 1.0);
    (void)hipMemcpy(x.p, ones.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(b.p, ones.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(y.p, ones.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(c.p, ones.data(), nc * 8, hipMemcpyHostToDevice);
    hipStream_t st = E.stream();
    auto run = [&]() {
        switch (op) {
        case 0: E.op_spmv(level, x.p, y.p); break;
        case 1: {
            CsrArgs a;
            a.x = x.p;
            a.b = b.p;
            a.d = L.diag;
            a.y = y.p;
            a.omega = E.params().omega;
            launch_csr(L.A, OP_JACOBI, a, L.fine, st, E.kernel_cfg());
        } break;
        case 2: E.op_residual(level, b.p, x.p, y.p); break;
        case 3: E.op_restrict(level, x.p, c.p); break;
        case 4: E.op_prolong(level, c.p, y.p); break;
        case 5: E.op_coarse(E.level(E.nlevels() - 1).b ? E.level(E.nlevels() - 1).b : b.p, E.level(E.nlevels() - 1).r); break;
        case 6: {
            int nb = 0;
            launch_dot((int)n, x.p, b.p, E.level(0).r, &nb, st);
        } break;
        case 7: launch_axpby((int)n, 0.5, x.p, 0.5, y.p, st); break;
        case 8: launch_copy_int((int)n, L.A.rowptr, reinterpret_cast<int *>(y.p), st); break;  // n int32: reads 4n, writes 4n
        case 9:    // fused Jacobi sweeps the way a smoothing leg issues them: ping-pong between two vectors
        case 10: { // ... on the level's own resident buffers (x, x2, r as the right-hand side)
            static thread_local int flip = 0;
            double *xa = op == 9 ? x.p : L.x, *xb = op == 9 ? y.p : L.x2;
            CsrArgs a;
            a.x = (flip & 1) ? xb : xa;
            a.y = (flip & 1) ? xa : xb;
            a.b = op == 9 ? b.p : L.r;
            a.d = L.diag;
            a.omega = E.params().omega;
            a.reverse = csr_alternates(L.A, E.kernel_cfg()) && (flip & 1);
            ++flip;
            launch_csr(L.A, OP_JACOBI, a, L.fine, st, E.kernel_cfg());
        } break;
        case 11: {  // double sweeps (sdia_box2_kernel) ping-ponging on the level's resident buffers, as a smoothing leg issues them
            static thread_local int flip2 = 0;
            double *xa = (flip2 & 1) ? L.x2 : L.x, *xb = (flip2 & 1) ? L.x : L.x2;
            ++flip2;
            launch_box2(L.A, xa, L.r, xb, E.params().omega, L.fine, st);
        } break;
        default: break;
        }
    };
    if (op < 0 || op > 11) return fail(SPARSH_EINVAL, "unknown op");
    if (op == 11 && (E.distributed() || !box2_applies(L.A, E.kernel_cfg()))) return fail(SPARSH_ESTATE, "the level does not run double sweeps (sparsh_level_double_sweep)");
    for (int i = 0; i < 3; ++i) run();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) run();
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_seconds = ms * 1e-3 / reps;
    return SPARSH_OK;
}

int sparsh_dev_alloc(sparsh_handle h, long nbytes, void **out)
{
    if (!h || !h->eng || !out || nbytes < 0) return fail(SPARSH_EINVAL, "bad arguments");
    *out = h->eng->dalloc((size_t)nbytes);
    return *out ? SPARSH_OK : fail(SPARSH_ENODEV, h->eng->error);
}

int sparsh_dev_free(sparsh_handle h, void *p)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    h->eng->dfree(p);
    return SPARSH_OK;
}

int sparsh_dev_fill(sparsh_handle h, double *dst_dev, long n, double value)
{
    if (!h || !h->eng || !h->eng->stream()) return fail(SPARSH_EINVAL, "null handle or no stream (call sparsh_setup first)");
    if (n < 0 || n > 0x7fffffffL) return fail(SPARSH_EINVAL, "bad length");
    launch_fill((int)n, value, dst_dev, h->eng->stream());
    return SPARSH_OK;
}

int sparsh_h2d(sparsh_handle h, void *dst_dev, const void *src, long nbytes)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (hipMemcpy(dst_dev, src, (size_t)nbytes, hipMemcpyHostToDevice) != hipSuccess) return fail(SPARSH_ENODEV, "H2D failed");
    return SPARSH_OK;
}

int sparsh_d2h(sparsh_handle h, void *dst, const void *src_dev, long nbytes)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (h->eng->stream()) (void)hipStreamSynchronize(h->eng->stream());
    if (hipMemcpy(dst, src_dev, (size_t)nbytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(SPARSH_ENODEV, "D2H failed");
    return SPARSH_OK;
}

int sparsh_sync(sparsh_handle h)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (h->eng->stream() && hipStreamSynchronize(h->eng->stream()) != hipSuccess) return fail(SPARSH_ENODEV, "sync failed");
    return SPARSH_OK;
}

int sparsh_profile(sparsh_handle h, int enable)
{
    if (!h || !h->eng) return fail(SPARSH_EINVAL, "null handle");
    if (enable) {
        h->eng->profile_begin();
        h->eng->prof.enabled = true;
    } else if (h->eng->prof.enabled) {
        h->eng->profile_collect();
        h->eng->prof.enabled = false;
    }
    return SPARSH_OK;
}

int sparsh_profile_read(sparsh_handle h, double *out4)
{
    REQUIRE_READY(h);
    out4[0] = h->eng->prof.launches;
    out4[1] = h->eng->prof.seconds;
    out4[2] = h->eng->level(0).n;
    out4[3] = h->eng->level(0).A.nnz;
    return SPARSH_OK;
}

}  // extern "C"
