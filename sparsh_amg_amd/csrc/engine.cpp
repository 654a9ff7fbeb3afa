// engine.cpp -- device-resident AMG hierarchy, V-cycle driver and Krylov loops.
//
// Reference behaviour followed (file:line relative to the reference tree):
//   V-cycle order of operations      src/AMG_phases.cpp:196-225 (CPU), src/AMG_gpu_phases_2.cu:173-226 ("MI")
//   AMG as stand-alone solver        src/AMG_phases.cpp:151-230
//   CG / AMG-PCG                     src/AMG_main_solvers.cpp:47-103, 107-167
//   BiCGStab / AMG-PBiCGStab         src/AMG_main_solvers.cpp:271-355, 358-458
// Design differences (MI355X-first): the whole hierarchy stays in HBM; every smoothing sweep is
// one fused kernel; the restriction uses an explicit R = P^T (gather, deterministic) instead of
// a transposed csrmv; the coarsest solve is a device GEMV with the explicit inverse instead of a
// host PARDISO round trip; CG/BiCGStab coefficients live in device memory and only the
// convergence norm is read back.
#include "engine.hpp"

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>

namespace sparsh {

namespace {
constexpr int kProfEvents = 8192;
}  // namespace

bool Engine::check(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

bool Engine::note_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    if (fault_ == SPARSH_OK) {
        fault_ = SPARSH_ENODEV;
        error = std::string("HIP error on the solve path: ") + what + ": " + hipGetErrorString(e);
    }
    return false;
}

bool Engine::note_comm(bool ok, const char *what)
{
    if (ok) return true;
    if (fault_ == SPARSH_OK) {
        fault_ = SPARSH_ECOMM;
        error = std::string(what) + " failed: " + (comm_ ? comm_->error : std::string("no transport"));
    }
    return false;
}

#define HIPCHK(call) note_hip((call), #call)

void *Engine::dalloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0) bytes = 8;
    if (!check(hipMalloc(&p, bytes), "hipMalloc")) return nullptr;
    allocs_.push_back(p);
    return p;
}

void Engine::dfree(void *p)
{
    if (!p) return;
    auto it = std::find(allocs_.begin(), allocs_.end(), p);
    if (it != allocs_.end()) allocs_.erase(it);
    (void)hipFree(p);
}

Engine::Engine(int nrow, int ncol, const int *rowptr, const int *col, const double *val)
{
    A0_ = HostCsr::alias(nrow, ncol, rowptr, col, val);
}

Engine::~Engine()
{
    if (st_) (void)hipStreamSynchronize(st_);
    drop_graph();
    for (auto e : prof.ev) (void)hipEventDestroy(e);
    for (void *p : allocs_) (void)hipFree(p);
    if (pinned_) (void)hipHostFree(pinned_);
    if (ev_ready_) (void)hipEventDestroy(ev_ready_);
    if (ev_halo_) (void)hipEventDestroy(ev_halo_);
    if (st2_) (void)hipStreamDestroy(st2_);
    if (st_) (void)hipStreamDestroy(st_);
}

namespace {

template <class T>
T *upload(Engine &E, const T *src, size_t count)
{
    T *d = static_cast<T *>(E.dalloc(count * sizeof(T)));
    if (!d) return nullptr;
    if (count && !E.check(hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D")) return nullptr;
    return d;
}

// col/val get kCsrPad zeroed tail entries: the paired loads of the stream kernels may read one
// entry past the last row
template <class T>
T *upload_padded(Engine &E, const T *src, size_t count)
{
    T *d = static_cast<T *>(E.dalloc((count + kCsrPad) * sizeof(T)));
    if (!d) return nullptr;
    if (!E.check(hipMemset(d + count, 0, kCsrPad * sizeof(T)), "hipMemset")) return nullptr;
    if (count && !E.check(hipMemcpy(d, src, count * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D")) return nullptr;
    return d;
}

// sliced-ELL mirror (slices of 64 rows); built only when padding stays below 1/8 of the entries
bool upload_sell(Engine &E, const HostCsr &A, DevCsr &D)
{
    const int n = A.nrow;
    if (n < 64) return true;
    const int nslice = (n + 63) / 64;
    std::vector<int> sp((size_t)nslice + 1, 0);
    long total = 0;
    for (int s = 0; s < nslice; ++s) {
        int mx = 0;
        const int r1 = std::min(n, (s + 1) * 64);
        for (int r = s * 64; r < r1; ++r) mx = std::max(mx, A.rowptr[r + 1] - A.rowptr[r]);
        total += (long)mx * 64;
        if (total >= (1l << 31) - 64) return true;  // would overflow int offsets: keep CSR only
        sp[(size_t)s + 1] = (int)total;
    }
    const long nnz = A.nnz();
    if (total > nnz + nnz / 8) return true;  // too ragged for ELL slices
    std::vector<int> sc((size_t)total, 0);
    std::vector<double> sv((size_t)total, 0.0);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < n; ++r) {
        const int s = r >> 6, lane = r & 63;
        const long base = sp[s];
        for (int j = A.rowptr[r], k = 0; j < A.rowptr[r + 1]; ++j, ++k) {
            sc[(size_t)(base + (long)k * 64 + lane)] = A.col[j];
            sv[(size_t)(base + (long)k * 64 + lane)] = A.val[j];
        }
    }
    D.nslice = nslice;
    D.sell_entries = total;
    D.slice_ptr = upload(E, sp.data(), sp.size());
    D.sell_col = upload(E, sc.data(), sc.size());
    D.sell_val = upload(E, sv.data(), sv.size());
    return D.slice_ptr && D.sell_col && D.sell_val;
}

// sliced-diagonal mirror; built only for operators whose slices hold few distinct diagonals
// (slots*64 <= 1.25 nnz) -- finite-difference / finite-volume stencils and their pairwise-
// aggregated coarse operators.  A slot is one (offset) diagonal inside a slice; slots are ordered
// by the entries' GLOBAL column offset, which is each row's entry order (sorted global columns),
// so the kernel adds a row's products in CSR order.  For a rank-local block the address offset
// (local col - local row) differs from the global one on halo columns; both are kept.
bool upload_sdia(Engine &E, const HostCsr &A, DevCsr &D)
{
    const int n = A.nrow;
    if (n < 64) return true;
    const bool local = !A.gcol_store.empty();
    if (!local && A.ncol != A.nrow) return true;
    const int *gcol = local ? A.gcol_store.data() : A.col;
    const int nslice = (n + 63) / 64;
    const long nnz = A.nnz();
    using Slot = std::pair<int, int>;  // (global offset = ordering key, local offset = address offset)
    auto slice_slots = [&](int s, std::vector<Slot> &slots, bool &sorted) {
        slots.clear();
        const int r1 = std::min(n, (s + 1) * 64);
        for (int r = s * 64; r < r1; ++r) {
            for (int j = A.rowptr[r]; j < A.rowptr[r + 1]; ++j) {
                slots.emplace_back(gcol[j] - A.grow(r), A.col[j] - r);
                if (j > A.rowptr[r] && gcol[j] <= gcol[j - 1]) sorted = false;  // unsorted / duplicate columns
            }
        }
        std::sort(slots.begin(), slots.end());
        slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
    };
    std::vector<int> sp((size_t)nslice + 1, 0);
    bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok)
    for (int s = 0; s < nslice; ++s) {
        std::vector<Slot> slots;
        bool sorted = true;
        slice_slots(s, slots, sorted);
        ok = ok && sorted;
        sp[(size_t)s + 1] = (int)slots.size();
    }
    if (!ok) return true;
    long total = 0;
    for (int s = 0; s < nslice; ++s) {
        total += sp[(size_t)s + 1];
        sp[(size_t)s + 1] = (int)total;
        if (total * 64 > (1l << 31) - 64) return true;
    }
    if (total * 64 > nnz + nnz / 4) return true;  // too many sparse diagonals: keep ELL / CSR
    std::vector<int> off((size_t)total);
    std::vector<unsigned long long> mask((size_t)total, 0ull);
    std::vector<double> val((size_t)total * 64, 0.0);
#pragma omp parallel for schedule(static)
    for (int s = 0; s < nslice; ++s) {
        std::vector<Slot> slots;
        bool sorted = true;
        slice_slots(s, slots, sorted);
        const int base = sp[s];
        for (size_t d = 0; d < slots.size(); ++d) off[(size_t)base + d] = slots[d].second;
        const int r1 = std::min(n, (s + 1) * 64);
        for (int r = s * 64; r < r1; ++r) {
            const int lane = r & 63;
            for (int j = A.rowptr[r]; j < A.rowptr[r + 1]; ++j) {
                const Slot key(gcol[j] - A.grow(r), A.col[j] - r);
                const int d = (int)(std::lower_bound(slots.begin(), slots.end(), key) - slots.begin());
                mask[(size_t)base + d] |= 1ull << lane;
                val[((size_t)base + d) * 64 + lane] = A.val[j];
            }
        }
    }
    // constant slots: every present entry of the slot holds the same bit pattern -> keep one scalar,
    // drop the 64-value block (the product v*x is the same multiplication, so results do not change)
    std::vector<int> vidx((size_t)total, 0);
    std::vector<double> cval((size_t)total, 0.0);
    const bool fold = E.kernel_cfg().const_slots && !E.params().precond_fp32;  // the float mirror converts whole blocks
#pragma omp parallel for schedule(static)
    for (long q = 0; q < total; ++q) {
        const unsigned long long m = mask[(size_t)q];
        const double *blk = &val[(size_t)q * 64];
        bool same = fold && m != 0;
        double first = 0.0;
        if (same) {
            first = blk[__builtin_ctzll(m)];
            for (int lane = 0; lane < 64 && same; ++lane)
                if ((m >> lane) & 1ull) same = std::memcmp(&blk[lane], &first, sizeof(double)) == 0;
        }
        vidx[(size_t)q] = same ? -1 : 0;
        cval[(size_t)q] = same ? first : 0.0;
    }
    long nblocks = 0;
    for (long q = 0; q < total; ++q)
        if (vidx[(size_t)q] == 0) vidx[(size_t)q] = (int)nblocks++;
    if (nblocks < total) {  // compact the value stream in place (blocks only move towards the front)
        for (long q = 0; q < total; ++q) {
            const int k = vidx[(size_t)q];
            if (k >= 0 && k != q) std::memmove(&val[(size_t)k * 64], &val[(size_t)q * 64], 64 * sizeof(double));
        }
        val.resize((size_t)std::max(nblocks, 1l) * 64);
    }
    // slices whose slots are all constant take the kernel's value-free path, slices without any
    // constant slot the indirection-free one: flag both kinds in sd_ptr (readers mask the bits off)
    if (total > kSdPtrMask) return true;  // flag bits need the room: keep ELL / CSR for such a level
    std::vector<char> allconst((size_t)nslice, 0), noconst((size_t)nslice, 0);
#pragma omp parallel for schedule(static)
    for (int sl = 0; sl < nslice; ++sl) {
        bool all = sp[(size_t)sl + 1] > sp[sl], none = true;
        for (int q = sp[sl]; q < sp[(size_t)sl + 1]; ++q) {
            all = all && vidx[(size_t)q] < 0;
            none = none && vidx[(size_t)q] >= 0;
        }
        allconst[(size_t)sl] = all;
        noconst[(size_t)sl] = none && sp[(size_t)sl + 1] > sp[sl];
    }
    for (int sl = 0; sl < nslice; ++sl) {
        if (allconst[(size_t)sl]) sp[sl] |= kSdConstBit;
        if (noconst[(size_t)sl]) sp[sl] |= kSdPlainBit;
    }
    // fixed-stride records of the value-free slices with at most 8 slots
    std::vector<int> rec;
    const bool with_rec = fold;
    if (with_rec) {
        rec.assign((size_t)nslice * kSdRecInts, 0);
#pragma omp parallel for schedule(static)
        for (int sl = 0; sl < nslice; ++sl) {
            int *r = &rec[(size_t)sl * kSdRecInts];
            const int q0 = sp[sl] & kSdPtrMask, q1 = sp[(size_t)sl + 1] & kSdPtrMask;
            if (!allconst[(size_t)sl] || q1 - q0 > 8) {
                r[40] = -1;
                continue;
            }
            for (int q = q0; q < q1; ++q) {
                r[q - q0] = off[(size_t)q];
                std::memcpy(&r[8 + 2 * (q - q0)], &mask[(size_t)q], 8);
                std::memcpy(&r[24 + 2 * (q - q0)], &cval[(size_t)q], 8);
            }
            r[40] = q1 - q0;
        }
    }
    D.nslice = nslice;
    D.sd_slots = total;
    D.sd_vblocks = nblocks;
    if (with_rec) {
        D.sd_rec = upload(E, rec.data(), rec.size());
        if (!D.sd_rec) return false;
    }
    // level-wide stencil table (see DevCsr::sd_tab): the most frequent full signature among the
    // value-free slices; a slice conforms when each of its slots is a (offset, constant) of the table
    if (with_rec) {
        auto sig_less = [&](int sa, int sb) {  // order slices by (count, offsets, constants)
            const int *ra = &rec[(size_t)sa * kSdRecInts], *rb = &rec[(size_t)sb * kSdRecInts];
            if (ra[40] != rb[40]) return ra[40] < rb[40];
            const int c = std::memcmp(ra, rb, 8 * sizeof(int));
            if (c) return c < 0;
            return std::memcmp(ra + 24, rb + 24, 8 * sizeof(double)) < 0;
        };
        auto sig_eq = [&](int sa, int sb) { return !sig_less(sa, sb) && !sig_less(sb, sa); };
        std::vector<int> cand;
        for (int sl = 0; sl < nslice; ++sl)
            if (rec[(size_t)sl * kSdRecInts + 40] > 0) cand.push_back(sl);
        if (!cand.empty()) {
            std::sort(cand.begin(), cand.end(), sig_less);
            int best_sl = -1;
            long best = 0;
            for (size_t i = 0; i < cand.size();) {
                size_t j = i;
                while (j < cand.size() && sig_eq(cand[i], cand[j])) ++j;
                if ((long)(j - i) > best) {
                    best = (long)(j - i);
                    best_sl = cand[i];
                }
                i = j;
            }
            SdTable tab;
            const int *rb = &rec[(size_t)best_sl * kSdRecInts];
            tab.nd = rb[40];
            for (int u = 0; u < tab.nd; ++u) {
                tab.off[u] = rb[u];
                std::memcpy(&tab.cval[u], &rb[24 + 2 * u], 8);
            }
            {
                const int c0 = tab.nd / 2;
                if ((tab.nd == 7 || tab.nd == 5 || tab.nd == 3) && tab.off[c0] == 0 && tab.off[c0 - 1] == -1 && tab.off[c0 + 1] == 1)
                    tab.near = tab.nd * 10 + c0;
            }
            std::vector<unsigned long long> tmask((size_t)nslice * 8, 0ull);
            std::vector<int> tconf((size_t)nslice, 0);
            long nconf = 0;
#pragma omp parallel for schedule(static) reduction(+ : nconf)
            for (int sl = 0; sl < nslice; ++sl) {
                const int *r = &rec[(size_t)sl * kSdRecInts];
                if (r[40] <= 0) continue;
                bool ok = true;
                unsigned long long mm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                int last_hit = -1;
                for (int q = 0; q < r[40] && ok; ++q) {
                    int hit = -1;
                    for (int u = 0; u < tab.nd; ++u)
                        if (tab.off[u] == r[q] && std::memcmp(&tab.cval[u], &r[24 + 2 * q], 8) == 0) hit = u;
                    // the slice's slots are stored in each row's entry order (global column order); the table kernel adds
                    // in table order, so the two orders must agree.  They can differ on ghost rows of a deep-halo block,
                    // whose neighbours in other layers sit at local offsets of the opposite sign (+plane <-> -plane).
                    if (hit <= last_hit) ok = false;
                    else std::memcpy(&mm[hit], &r[8 + 2 * q], 8);
                    last_hit = hit;
                }
                if (!ok) continue;
                for (int u = 0; u < 8; ++u) tmask[(size_t)sl * 8 + u] = mm[u];
                tconf[(size_t)sl] = 1;
                ++nconf;
            }
            if (nconf * 10 >= (long)nslice * 9) {  // worth a launch-wide assumption only if it nearly always holds
                D.sd_tab = tab;
                D.sd_tmask = upload(E, tmask.data(), tmask.size());
                D.sd_tconf = upload(E, tconf.data(), tconf.size());
                if (!D.sd_tmask || !D.sd_tconf) return false;
            }
            // box grid (DevCsr::box_nx): 7-point table (-P, -L, -1, 0, +1, +L, +P), every slice on it, and every row lacking
            // exactly the neighbours outside an L x P/L x n/P box
            if (nconf == (long)nslice && !local && tab.near == 73 && A.ncol == n && tab.off[5] >= 2 && tab.off[1] == -tab.off[5] && tab.off[6] > tab.off[5] &&
                tab.off[0] == -tab.off[6] && tab.off[6] % tab.off[5] == 0 && n % tab.off[6] == 0) {
                const int nx = tab.off[5], P = tab.off[6], ny = P / nx, nz = n / P;
                long bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
                for (int sl = 0; sl < nslice; ++sl) {
                    const unsigned long long *mm = &tmask[(size_t)sl * 8];
                    const int r1 = std::min(n, (sl + 1) * 64);
                    for (int r = sl * 64; r < r1; ++r) {
                        const int k = r / P, rem = r - k * P, j = rem / nx, i = rem - j * nx;
                        const bool want[7] = {k > 0, j > 0, i > 0, true, i < nx - 1, j < ny - 1, k < nz - 1};
                        for (int u = 0; u < 7; ++u)
                            if ((((mm[u] >> (r & 63)) & 1ull) != 0) != want[u]) ++bad;
                    }
                }
                if (bad == 0) {
                    D.box_nx = nx;
                    D.box_ny = ny;
                    D.box_nz = nz;
                    box2_plan(D);
                    box1_plan(D);
                }
            }
        }
    }
    D.sd_ptr = upload(E, sp.data(), sp.size());
    D.sd_off = upload(E, off.data(), off.size());
    D.sd_mask = upload(E, mask.data(), mask.size());
    D.sd_vidx = upload(E, vidx.data(), vidx.size());
    D.sd_cval = upload(E, cval.data(), cval.size());
    D.sd_val = upload(E, val.data(), val.size());
    return D.sd_ptr && D.sd_off && D.sd_mask && D.sd_vidx && D.sd_cval && D.sd_val;
}

// cuts: row indices at which a row block / wave block must end (deep-halo operators: launches over a prefix of
// the rows -- own rows, own rows + some ghost layers -- must not spill into the next segment); blk_first /
// wblk_first receive the first row of every block for the prefix look-up
bool upload_csr(Engine &E, const HostCsr &A, DevCsr &D, bool with_sell, const std::vector<int> *cuts = nullptr,
                std::vector<int> *blk_first = nullptr, std::vector<int> *wblk_first = nullptr)
{
    D.nrow = A.nrow;
    D.ncol = A.ncol;
    D.nnz = A.nnz();
    D.rowptr = upload(E, A.rowptr, (size_t)A.nrow + 1);
    D.col = upload_padded(E, A.col, (size_t)D.nnz);
    D.val = upload_padded(E, A.val, (size_t)D.nnz);
    std::vector<int> rb((size_t)A.nrow + 2);
    std::vector<int> recs;  // the row-block records as uploaded
    if (!cuts) {
        const std::vector<int> rec = rowblock_records(A.nrow, A.rowptr, &D.nblk);
        recs = rec;
        D.rowblk = upload(E, rec.data(), rec.size());
        D.nwblk = build_waveblocks(A.nrow, A.rowptr, rb.data());
        D.waveblk = upload(E, rb.data(), (size_t)D.nwblk + 1);
    } else {
        // schedules built segment by segment: no block straddles a cut
        std::vector<int> rec, wb(1, 0);
        int seg0 = 0;
        std::vector<int> ends(*cuts);
        ends.push_back(A.nrow);
        for (int seg1 : ends) {
            if (seg1 <= seg0) continue;
            int nb = 0;
            const std::vector<int> r = rowblock_records(seg1 - seg0, A.rowptr + seg0, &nb);
            for (int k = 0; k < nb; ++k) {
                rec.push_back(r[(size_t)4 * k] + seg0);
                rec.push_back(r[(size_t)4 * k + 1] + seg0);
                rec.push_back(r[(size_t)4 * k + 2]);
                rec.push_back(r[(size_t)4 * k + 3]);
                if (blk_first) blk_first->push_back(r[(size_t)4 * k] + seg0);
            }
            const int nw = build_waveblocks(seg1 - seg0, A.rowptr + seg0, rb.data());
            for (int k = 0; k < nw; ++k) {
                if (wblk_first) wblk_first->push_back(rb[k] + seg0);
                wb.push_back(rb[k + 1] + seg0);
            }
            seg0 = seg1;
        }
        D.nblk = (int)(rec.size() / 4);
        if (rec.empty()) rec.assign(4, 0);
        recs = rec;
        D.rowblk = upload(E, rec.data(), rec.size());
        D.nwblk = (int)wb.size() - 1;
        D.waveblk = upload(E, wb.data(), wb.size());
    }
    // 16-bit delta form of the column indices (csr_rowlane16_kernel): for every operator when asked for, by default for
    // those the default policy streams from HBM through the CSR-stream kernel (decided below, once the mirrors exist)
    const auto build16 = [&]() -> bool {
        std::vector<unsigned short> c16((size_t)D.nnz + kCsrPad, 0);
        std::vector<int> cb((size_t)std::max(D.nblk, 1), -1);
        D.nblk16 = build_col16(A.rowptr, A.col, recs.data(), D.nblk, c16.data(), cb.data());
        if (D.nblk16 == 0) return true;  // nothing fits: the operator keeps its 32-bit indices only
        D.col16 = upload(E, c16.data(), c16.size());
        D.cbase = upload(E, cb.data(), cb.size());
        return D.col16 && D.cbase;
    };
    if (with_sell && !upload_sell(E, A, D)) {
        if (E.error.empty()) E.error = "building the sliced-ELL mirror failed";
        return false;
    }
    if (with_sell && !upload_sdia(E, A, D)) {
        if (E.error.empty()) E.error = "building the sliced-diagonal mirror failed";
        return false;
    }
    if (D.nblk > 0 && D.rowblk) {
        const int mode = E.kernel_cfg().idx16;
        KernelConfig dflt;
        const bool streams = csr_family(D, dflt) == FAM_CSR_ROWLANE;  // default policy: CSR-stream kernel, operator > 240 MB (col16 not built yet)
        if ((mode == 2 || (mode == 1 && streams)) && !build16()) {
            if (E.error.empty()) E.error = "building the 16-bit column index form failed";
            return false;
        }
    }
    if (!(D.rowptr && D.col && D.val && D.rowblk && D.waveblk)) {
        if (E.error.empty()) E.error = "uploading a CSR operator failed";
        return false;
    }
    return true;
}

int partial_count(const DevCsr &D) { return std::max(D.nblk, std::max((D.nwblk + 3) / 4, (D.nslice + 3) / 4)); }

}  // namespace

int Engine::setup_host(const sparsh_params &p)
{
    prm_ = p;
    ready_ = false;
    host_ready_ = false;
    SetupParams sp;
    sp.max_levels = p.max_levels;
    sp.limit_upper = p.limit_upper;
    sp.limit_lower = p.limit_lower;
    sp.coarsening = p.coarsening;
    sp.coarse_limit = p.coarse_limit;
    if (p.dense_limit > 0) sp.dense_limit = p.dense_limit;
    sp.extend_until = p.extend_until;
    sp.coarse_factor_bytes = p.coarse_factor_mb > 0 ? (size_t)p.coarse_factor_mb << 20 : 0;
    if (coarse_.form() == 1) sp.coarse_factor_bytes = 0;  // the estimate is the nested-dissection solver's
    sp.host_threads = p.host_threads;
    sp.print = p.print_setup != 0;
    if (!build_hierarchy(A0_, sp, H_)) {
        error = H_.error;
        return SPARSH_ENUMERIC;
    }
    setup_seconds = H_.seconds;
    if (!H_.coarse_dense) {
        // The device direct solver may refuse what max_levels left over (no usable separators / band, factors beyond the memory
        // budget -- e.g. a coarsening that stalled far above limit_upper).  Then the hierarchy is extended by the reference's own
        // coarsening rule instead of failing: coarse_limit = dense_limit, the behaviour before the reference policy became the default.
        const double tp = omp_get_wtime();
        std::string perr;
        if (!coarse_.probe(H_.levels.back().A, perr)) {
            if (sp.print) std::printf("note: %s -- extending the hierarchy instead\n", perr.c_str());
            SetupParams sp2 = sp;
            sp2.coarse_limit = sp.dense_limit;
            sp2.extend_until = sp.limit_upper;
            sp2.coarse_factor_bytes = 0;
            const double before = H_.seconds;
            if (!build_hierarchy(A0_, sp2, H_)) {
                error = H_.error;
                return SPARSH_ENUMERIC;
            }
            H_.seconds += before;
            if (!H_.coarse_dense && !coarse_.probe(H_.levels.back().A, perr)) {
                error = perr;
                return SPARSH_EINVAL;
            }
        }
        H_.seconds += omp_get_wtime() - tp;
        setup_seconds = H_.seconds;
    }
    host_ready_ = true;
    return SPARSH_OK;
}

int Engine::setup_host_shared(const sparsh_params &p)
{
    const int me = comm_->rank;
    const double t0 = omp_get_wtime();
    // every rank, not only the one that builds: the OpenMP regions of the local extraction and the uploads that follow must not
    // oversubscribe the host with (ranks x all cores) threads
    omp_set_num_threads(p.host_threads > 0 ? p.host_threads : effective_cpus());
    std::vector<char> image;
    int rc0 = SPARSH_OK;
    std::string err0;
    if (me == 0) {
        rc0 = setup_host(p);
        err0 = error;
        if (rc0 == SPARSH_OK) serialize_hierarchy(H_, image);
    }
    built_locally_ = (me == 0);
    // header: image size and rank 0's return code (every rank takes part in both broadcasts, also after a failed setup)
    long long *hdr = nullptr;
    if (!check(hipMalloc(reinterpret_cast<void **>(&hdr), 16), "hipMalloc")) return SPARSH_ENODEV;
    long long h2[2] = {(long long)image.size(), (long long)rc0};
    bool good = check(hipMemcpyAsync(hdr, h2, 16, hipMemcpyHostToDevice, st_), "hipMemcpy");
    good = good && comm_->bcast(hdr, 16, 0, st_);
    good = good && check(hipMemcpyAsync(h2, hdr, 16, hipMemcpyDeviceToHost, st_), "hipMemcpy") && check(hipStreamSynchronize(st_), "hipStreamSynchronize");
    (void)hipFree(hdr);
    if (!good) {
        if (error.empty()) error = "broadcast of the hierarchy header failed: " + comm_->error;
        return SPARSH_ECOMM;
    }
    if (h2[1] != SPARSH_OK) {
        error = me == 0 ? err0 : "the setup on rank 0 failed";
        host_ready_ = false;
        return (int)h2[1];
    }
    const size_t total = (size_t)h2[0];
    image_bytes_ = total;
    if (me != 0) image.resize(total);
    // the image travels through one device staging buffer, chunk by chunk
    const size_t chunk = std::min<size_t>(total, (size_t)256 << 20);
    char *stage = nullptr;
    // A rank whose staging buffer cannot be allocated must not leave the others waiting in the broadcasts below: the ranks
    // agree on the outcome first (sum of failure flags), then either all proceed or all return the error.
    {
        const bool mine = chunk == 0 || hipMalloc(reinterpret_cast<void **>(&stage), chunk) == hipSuccess;
        double *flag = nullptr;
        if (!check(hipMalloc(reinterpret_cast<void **>(&flag), sizeof(double)), "hipMalloc")) {
            if (stage) (void)hipFree(stage);
            return SPARSH_ENODEV;
        }
        double f = mine ? 0.0 : 1.0;
        bool okf = check(hipMemcpyAsync(flag, &f, sizeof(double), hipMemcpyHostToDevice, st_), "hipMemcpy") && comm_->allreduce_sum(flag, 1, st_) &&
                   check(hipMemcpyAsync(&f, flag, sizeof(double), hipMemcpyDeviceToHost, st_), "hipMemcpy") && check(hipStreamSynchronize(st_), "hipStreamSynchronize");
        (void)hipFree(flag);
        if (!okf || f != 0.0) {
            if (stage) (void)hipFree(stage);
            if (!okf) {
                if (error.empty()) error = "agreement on the hierarchy staging buffers failed: " + comm_->error;
                return SPARSH_ECOMM;
            }
            error = mine ? "another rank could not allocate its staging buffer for the hierarchy broadcast"
                         : "hipMalloc of the " + std::to_string(chunk >> 20) + " MB staging buffer for the hierarchy broadcast failed";
            return SPARSH_ENODEV;
        }
    }
    for (size_t off = 0; good && off < total; off += chunk) {
        const size_t nb = std::min(chunk, total - off);
        if (me == 0) good = check(hipMemcpyAsync(stage, image.data() + off, nb, hipMemcpyHostToDevice, st_), "hipMemcpy");
        good = good && comm_->bcast(stage, nb, 0, st_);
        if (good && me != 0) good = check(hipMemcpyAsync(image.data() + off, stage, nb, hipMemcpyDeviceToHost, st_), "hipMemcpy");
        good = good && check(hipStreamSynchronize(st_), "hipStreamSynchronize");
    }
    if (stage) (void)hipFree(stage);
    if (!good) {
        if (error.empty()) error = "broadcast of the hierarchy failed: " + comm_->error;
        return SPARSH_ECOMM;
    }
    if (me != 0) {
        prm_ = p;
        ready_ = false;
        if (!deserialize_hierarchy(image.data(), image.size(), A0_, H_)) {
            error = H_.error;
            host_ready_ = false;
            return SPARSH_ENUMERIC;
        }
        host_ready_ = true;
        setup_seconds = omp_get_wtime() - t0;
        H_.seconds = setup_seconds;
    }
    return SPARSH_OK;
}

// The dominant kernel of a large stencil level streams three vectors per sweep: iterate, ping-pong twin, right-hand side (inside
// PCG: the Krylov residual).  At 10 M rows that is 0.24 GB, the size of the Infinity Cache, and how much of it survives from
// one sweep to the next depends on the physical pages behind the three allocations: 44 to 63 us for the same sweep from one
// process to the next (profiles/r02_finest_sweep_placement_luck.txt).  The engine owns eleven buffers of that size anyway
// (x, x2, r of level 0 and the Krylov vectors), so it times the sweep on the candidate triples once and lets the best one
// play the three roles -- an assignment of pointers, no extra memory, results unchanged.
// Box-grid levels: does the double sweep (sdia_box2_kernel) beat two single sweeps here?  Timed on the level's own buffers
// (KernelConfig::box2 = 1, levels of >= 400 000 rows: below that two launches of a cache-resident level win, tools/micro/box2_proto)
// or switched on wherever a plan exists (box2 = 2: tests, A/B).
void Engine::tune_box_kernels()
{
    for (size_t l = 0; l + 1 < lev_.size(); ++l) {
        DevLevel &L = lev_[l];
        L.A.box_on = L.A.box1_on = false;
        L.box_single_us = L.box_double_us = L.box1_table_us = L.box1_us = 0.0;
        // (several GPUs: the replicated levels are whole levels on every rank and take the same path; ranks may decide differently,
        // the results are the same bits either way)
        if (L.A.box_q <= 0 || (dist_ && !L.replicated) || L.deep || csr_family(L.A, cfg_) != FAM_SDIA_TAB) continue;
        if (cfg_.box2 >= 2) L.A.box_on = true;
        if (cfg_.box1 >= 2) L.A.box1_on = L.A.box1_q > 0;
        const bool time2 = cfg_.box2 == 1, time1 = cfg_.box1 == 1 && L.A.box1_q > 0;
        if (!(time1 || time2)) continue;
        if (L.n < 400000) continue;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) continue;
        CsrArgs a;
        a.b = L.r;  // (scratch: any vector of the level's length; ones, so that the sweeps work on ordinary numbers)
        launch_fill(L.n, 1.0, L.r, st_);
        a.d = L.diag;
        a.omega = prm_.omega;
        auto timed = [&](bool fused) -> double {
            double *p = L.x, *q = L.x2;
            const int launches = fused ? 5 : 10;
            for (int it = 0; it < launches; ++it) {
                if (it == (fused ? 1 : 2)) (void)hipEventRecord(e0, st_);
                if (fused) {
                    launch_box2(L.A, p, a.b, q, prm_.omega, L.fine, st_);
                } else {
                    a.x = p;
                    a.y = q;
                    launch_csr(L.A, OP_JACOBI, a, L.fine, st_, cfg_);
                }
                std::swap(p, q);
            }
            (void)hipEventRecord(e1, st_);
            if (hipEventSynchronize(e1) != hipSuccess) return 1e30;
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            return (double)ms * 1e3 / 4.0;  // per pair of sweeps
        };
        if (time2) {
            L.box_single_us = timed(false);
            L.box_double_us = timed(true);
            L.A.box_on = L.box_double_us < 0.97 * L.box_single_us;
        }
        if (time1) {
            // the launches with an epilogue: the last post-sweep with its dot through the table kernel against the plane-marching kernel
            auto timed1 = [&](bool marching) -> double {
                double *p = L.x, *q = L.x2;
                for (int it = 0; it < 8; ++it) {
                    if (it == 2) (void)hipEventRecord(e0, st_);
                    a.x = p;
                    a.y = q;
                    a.partial = part0_;
                    if (marching) launch_box1(L.A, 1, a, L.fine, st_);
                    else launch_csr(L.A, OP_JACOBI_DOT, a, L.fine, st_, cfg_);
                    std::swap(p, q);
                }
                (void)hipEventRecord(e1, st_);
                if (hipEventSynchronize(e1) != hipSuccess) return 1e30;
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                return (double)ms * 1e3 / 6.0;
            };
            L.box1_table_us = timed1(false);
            L.box1_us = timed1(true);
            {  // the plan that lets two workgroups share a CU, if it is another one: keep the faster
                const int q0 = L.A.box1_q, ty0 = L.A.box1_ty, cz0 = L.A.box1_cz;
                if (box1_plan(L.A, true) && (L.A.box1_q != q0 || L.A.box1_ty != ty0 || L.A.box1_cz != cz0)) {
                    const double t_alt = timed1(true);
                    if (t_alt < L.box1_us) {
                        L.box1_us = t_alt;
                    } else {
                        L.A.box1_q = q0;
                        L.A.box1_ty = ty0;
                        L.A.box1_cz = cz0;
                    }
                } else {
                    L.A.box1_q = q0;
                    L.A.box1_ty = ty0;
                    L.A.box1_cz = cz0;
                }
            }
            L.A.box1_on = L.box1_us < 0.97 * L.box1_table_us;
        }
        (void)hipMemsetAsync(L.x, 0, (size_t)L.n * 8, st_);
        (void)hipMemsetAsync(L.x2, 0, (size_t)L.n * 8, st_);
        (void)hipMemsetAsync(L.r, 0, (size_t)L.n * 8, st_);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
}

void Engine::tune_placement()
{
    if (lev_.size() < 2) return;
    DevLevel &L = lev_[0];
    const size_t bytes = (size_t)L.n * 24;
    if (bytes < ((size_t)96 << 20) || bytes > ((size_t)512 << 20)) return;  // far below / far above the cache: nothing to choose
    const CsrFamily fam = csr_family(L.A, cfg_);
    if (fam != FAM_SDIA_TAB) return;  // layouts that stream a matrix are not sensitive (their working set is several caches)
    std::vector<double **> slots = {&L.x, &L.x2, &work_[0], &L.r};
    for (size_t k = 1; k < work_.size(); ++k) slots.push_back(&work_[k]);
    std::vector<double *> buf;
    for (double **s : slots) buf.push_back(*s);
    const size_t owned = buf.size();
    // a few spare buffers widen the choice (the sweep times of the triples fall into three groups -- no, one or two pairs of
    // vectors in each other's way -- and eleven buffers do not always contain a clean triple); the ones not chosen are freed
    const size_t vbytes = (size_t)L.n * 8;
    for (int q = 0; q < 5; ++q) {
        void *sp = nullptr;
        if (hipMalloc(&sp, vbytes) != hipSuccess) break;
        buf.push_back(static_cast<double *>(sp));
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        for (size_t q = owned; q < buf.size(); ++q) (void)hipFree(buf[q]);
        return;
    }
    auto probe = [&](double *x, double *x2, double *b) -> double {
        CsrArgs a;
        a.b = b;
        a.d = L.diag;
        a.omega = prm_.omega;
        double *p = x, *q = x2;
        const bool pairs = box2_applies(L.A, cfg_);  // the kernel the smoothing legs will run: 5 double sweeps or 10 single ones
        for (int it = 0; it < (pairs ? 5 : 10); ++it) {
            if (it == (pairs ? 1 : 2)) (void)hipEventRecord(e0, st_);
            if (pairs) {
                launch_box2(L.A, p, b, q, prm_.omega, L.fine, st_);
            } else {
                a.x = p;
                a.y = q;
                launch_csr(L.A, OP_JACOBI, a, L.fine, st_, cfg_);
            }
            std::swap(p, q);
        }
        (void)hipEventRecord(e1, st_);
        if (hipEventSynchronize(e1) != hipSuccess) return 1e30;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return (double)ms * 1e3 / 8.0;
    };
    const double t_begin = omp_get_wtime();
    // the probes sweep ordinary numbers (ones), not the zeros a fresh setup leaves: a zero residual takes the plain-division branch of
    // the double sweep (div_const), which the solve never sees
    for (double *q : buf) launch_fill(L.n, 1.0, q, st_);
    // stop at the first triple within 4 % of what the sweep's own 24 bytes per row take at 5.8 TB/s (a cache-resident run);
    // otherwise the best of at most 260 triples, the owned buffers first
    // (double sweeps: the probe reports us per sweep = half a launch; 0.6 of the single sweep's figure is a cache-friendly run,
    // tools/micro/box2_proto 57 us per launch at 216^3)
    const double good_us = (double)L.n * 24.0 / 5.8e12 * 1e6 * 1.04 * (box2_applies(L.A, cfg_) ? 0.6 : 1.0);
    double best = 1e30, worst = 0.0;
    int bi = 0, bj = 1, bk = 2;
    const int nb = (int)buf.size();
    bool done = false;
    for (int hi = 2; hi < nb && !done; ++hi)          // triples ordered by their largest member: spares come last
        for (int j = 1; j < hi && !done; ++j)
            for (int i = 0; i < j && !done; ++i) {
                const double t = probe(buf[i], buf[j], buf[hi]);
                if (place_tried == 0) place_first_us = t;  // (0, 1, 2) = the assignment the allocation order gives
                ++place_tried;
                if (t < best) {
                    best = t;
                    bi = i;
                    bj = j;
                    bk = hi;
                }
                worst = std::max(worst, t);
                done = best <= good_us || place_tried >= 260;
            }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    place_best_us = best;
    place_worst_us = worst;
    // hand out the buffers: chosen triple to the three roles, the rest in their old order to the remaining slots; what is left
    // over (as many as there were spares) is freed -- a chosen spare takes the place of an owned buffer in the engine's books
    std::vector<double *> rest;
    for (int q = 0; q < nb; ++q)
        if (q != bi && q != bj && q != bk) rest.push_back(buf[q]);
    *slots[0] = buf[bi];
    *slots[1] = buf[bj];
    *slots[2] = buf[bk];
    size_t r = 0;
    for (size_t q = 3; q < slots.size(); ++q) *slots[q] = rest[r++];
    std::vector<double *> keep = {buf[bi], buf[bj], buf[bk]};
    for (size_t q = 3; q < slots.size(); ++q) keep.push_back(*slots[q]);
    for (; r < rest.size(); ++r) {  // leftovers
        double *dead = rest[r];
        auto it = std::find(allocs_.begin(), allocs_.end(), static_cast<void *>(dead));
        if (it != allocs_.end()) allocs_.erase(it);
        (void)hipFree(dead);
    }
    for (double *kq : keep)
        if (std::find(allocs_.begin(), allocs_.end(), static_cast<void *>(kq)) == allocs_.end()) allocs_.push_back(kq);
    // the probes wrote into the buffers: back to the zeroed state a fresh setup leaves
    for (double *kq : keep) (void)hipMemsetAsync(kq, 0, vbytes, st_);
    place_seconds = omp_get_wtime() - t_begin;
}

bool Engine::upload_plan(const HaloPlan &h, DevPlan &d)
{
    d.nloc = h.nloc;
    d.nhalo = h.nhalo;
    d.nsend = (int)h.send_idx.size();
    d.recv = h.recv;
    d.send = h.send;
    d.need_pack = false;
    for (const HaloSeg &sg : d.send) d.need_pack = d.need_pack || sg.start < 0;
    if (d.nsend > 0) {
        d.send_idx = upload(*this, h.send_idx.data(), h.send_idx.size());
        d.sendbuf = static_cast<double *>(dalloc((size_t)d.nsend * 8));
        if (!d.send_idx || !d.sendbuf) return false;
    }
    return true;
}

bool Engine::upload_deep_plan(const DeepPlan &h, DevDeepPlan &d)
{
    d.depth = h.depth;
    d.nrecv = h.nrecv;
    d.nsend = (int)h.send_idx.size();
    d.recv = h.recv;
    d.send = h.send;
    if (d.nsend > 0) {
        d.send_idx = upload(*this, h.send_idx.data(), h.send_idx.size());
        d.sendbuf = static_cast<double *>(dalloc((size_t)d.nsend * 8));
        if (!d.send_idx || !d.sendbuf) return false;
    }
    if (d.nrecv > 0) {
        d.recv_pos = upload(*this, h.recv_pos.data(), h.recv_pos.size());
        d.recvbuf = static_cast<double *>(dalloc((size_t)d.nrecv * 8));
        if (!d.recv_pos || !d.recvbuf) return false;
    }
    return true;
}

bool Engine::deep_exchange(DevLevel &L, int which, double *vec)
{
    const DevDeepPlan &p = L.dplan[which];
    if (p.depth <= 0) return true;
    ++n_exchanges_;
    if (!note_comm(comm_->exchange_staged(p, vec, st_), "deep-halo exchange")) return false;
    launch_unpack(p.nrecv, p.recv_pos, p.recvbuf, vec, st_);
    return true;
}

// A deep level's operator restricted to its first `rows` local rows (own rows, or own rows + the ghost layers a
// sweep still has to update).  Whole slices / row blocks are launched: rows past the prefix inside the last one
// produce values nobody reads.
int Engine::launch_prefix(DevLevel &L, int rows, CsrOp op, const CsrArgs &a)
{
    DevCsr V = L.A;
    rows = std::min(rows, L.A.nrow);
    V.nrow = rows;  // exact: the sliced kernels mask the rows past it, the block schedules are cut at every prefix end
    if (V.nslice > 0) V.nslice = (rows + 63) / 64;
    V.nblk = (int)(std::lower_bound(L.blk_first.begin(), L.blk_first.end(), rows) - L.blk_first.begin());
    V.nwblk = (int)(std::lower_bound(L.wblk_first.begin(), L.wblk_first.end(), rows) - L.wblk_first.begin());
    return launch_csr(V, op, a, L.fine, st_, cfg_);
}

bool Engine::debug_prefix_spmv(int l, int rows, const double *x_ext, double *y)
{
    CsrArgs a;
    a.x = x_ext;
    a.y = y;
    launch_prefix(lev_[l], rows, OP_SPMV, a);
    return check(hipStreamSynchronize(st_), "sync");
}

// ---- transport-measured schedule (multi-rank setups) -------------------------------------------------------------------
// Nobody can tune replicate_rows / deep-halo by hand for a node this code has never run on, so the setup measures: one
// neighbour exchange (small and large: latency + rate), the 16-byte all-reduce, one all-gather (small and large), and the
// device's streaming rate and launch floor (an axpby, small and large).  Every rank measures, the slowest rank's numbers
// count.  A level's share of one V(nu,nu) cycle is then modelled three ways -- partitioned with deep halos (4 exchanges,
// redundant ghost-row sweeps), partitioned with one exchange per sweep (2 nu + 2 exchanges), replicated (no exchange, G times
// the rows) -- and the cheapest consistent choice wins: a prefix of levels is partitioned, with one smoothing schedule.
namespace {

double stream_bytes_per_row(const HostCsr &A)
{
    // what the layout builder will stream per row and sweep (mirrors its own rule on a sample of rows): constant-coefficient
    // stencils fold into a table (24 B of vectors + 1 B of masks), few diagonals stream 8 B per entry, the rest CSR's 12 B
    const int n = A.nrow;
    if (n <= 0) return 36.0;
    std::vector<std::pair<int, double>> seen;
    const int step = std::max(1, n / 2048);
    bool many = false;
    for (int i = 0; i < n && !many; i += step)
        for (int j = A.rowptr[i]; j < A.rowptr[i + 1] && !many; ++j) {
            const std::pair<int, double> e(A.col[j] - i, A.val[j]);
            if (std::find(seen.begin(), seen.end(), e) == seen.end()) {
                seen.push_back(e);
                many = seen.size() > 64;
            }
        }
    const double per_row = (double)A.nnz() / n;
    if (seen.size() <= 8) return 25.0;
    if (!many) return 8.0 * per_row + 27.0;
    return 12.0 * per_row + 36.0;
}

int boundary_rows(const HostCsr &A, int lo, int hi)
{
    // entries of the input vector a rank owning rows [lo, hi) needs from the others
    std::vector<char> mark((size_t)A.ncol, 0);
    int cnt = 0;
    for (int i = lo; i < hi; ++i)
        for (int j = A.rowptr[i]; j < A.rowptr[i + 1]; ++j) {
            const int c = A.col[j];
            if ((c < lo || c >= hi) && !mark[c]) {
                mark[c] = 1;
                ++cnt;
            }
        }
    return cnt;
}

}  // namespace

bool Engine::measure_transport()
{
    meas_ = CommMeasured();
    const int G = comm_->size, me = comm_->rank;
    const int small = 128, large = 1 << 16;  // doubles per neighbour
    const int agn_small = 1 << 14, agn_large = 1 << 21;
    const int axn_small = 1 << 14, axn_large = 1 << 22;
    double *vec = nullptr, *ag = nullptr, *ax = nullptr, *slots = nullptr;
    auto cleanup = [&]() {
        for (double *q : {vec, ag, ax, slots})
            if (q) (void)hipFree(q);
    };
    // the small table first: with it every rank can tell the others whether its larger buffers could be allocated, so that either
    // all ranks go on to the timed collectives or none does (a rank that left alone would leave the others waiting in an exchange)
    if (hipMalloc(reinterpret_cast<void **>(&slots), (size_t)G * 8 * 8) != hipSuccess) return false;
    {
        const bool mine = hipMalloc(reinterpret_cast<void **>(&vec), (size_t)4 * large * 8) == hipSuccess &&
                          hipMalloc(reinterpret_cast<void **>(&ag), (size_t)agn_large * 8) == hipSuccess &&
                          hipMalloc(reinterpret_cast<void **>(&ax), (size_t)2 * axn_large * 8) == hipSuccess;
        double f = mine ? 0.0 : 1.0;
        const bool okf = check(hipMemcpyAsync(slots, &f, sizeof(double), hipMemcpyHostToDevice, st_), "hipMemcpy") && comm_->allreduce_sum(slots, 1, st_) &&
                         check(hipMemcpyAsync(&f, slots, sizeof(double), hipMemcpyDeviceToHost, st_), "hipMemcpy") && check(hipStreamSynchronize(st_), "hipStreamSynchronize");
        if (!okf || f != 0.0) {
            cleanup();
            return false;
        }
    }
    (void)hipMemsetAsync(vec, 0, (size_t)4 * large * 8, st_);
    (void)hipMemsetAsync(ag, 0, (size_t)agn_large * 8, st_);
    (void)hipMemsetAsync(ax, 0, (size_t)2 * axn_large * 8, st_);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    bool good = true;
    auto timed = [&](int reps, const std::function<bool()> &fn) {
        for (int i = 0; i < 2 && good; ++i) good = fn();
        HIPCHK(hipEventRecord(e0, st_));
        for (int i = 0; i < reps && good; ++i) good = fn();
        HIPCHK(hipEventRecord(e1, st_));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms * 1e3 / reps;  // microseconds per call
    };
    auto ring_plan = [&](int cnt) {
        DevPlan pl;
        pl.nloc = 2 * cnt;
        int k = 0;
        for (int peer : {me - 1, me + 1}) {
            if (peer < 0 || peer >= G) continue;
            HaloSeg r, sd;
            r.peer = sd.peer = peer;
            r.off = k * cnt;
            r.cnt = sd.cnt = cnt;
            sd.off = 0;
            sd.start = k * cnt;  // a contiguous run of the own entries: sent in place
            pl.recv.push_back(r);
            pl.send.push_back(sd);
            ++k;
        }
        pl.nhalo = k * cnt;
        return pl;
    };
    double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    {
        const DevPlan ps = ring_plan(small), pl = ring_plan(large);
        m[0] = timed(20, [&] { return comm_->exchange(ps, vec, st_); });
        m[1] = timed(10, [&] { return comm_->exchange(pl, vec, st_); });
    }
    m[2] = timed(20, [&] { return comm_->allreduce_sum(slots, 2, st_); });
    {
        const Partition ps = make_partition(agn_small, G), pl = make_partition(agn_large, G);
        m[3] = timed(10, [&] { return comm_->allgather(ag, ps, st_); });
        m[4] = timed(5, [&] { return comm_->allgather(ag, pl, st_); });
    }
    m[5] = timed(50, [&] {
        launch_axpby(axn_small, 0.5, ax, 0.5, ax + axn_large, st_);
        return true;
    });
    m[6] = timed(20, [&] {
        launch_axpby(axn_large, 0.5, ax, 0.5, ax + axn_large, st_);
        return true;
    });
    // the slowest rank's numbers count, and every rank must decide from the same ones: each rank writes its slot, all-reduce, max
    std::vector<double> all((size_t)G * 8, 0.0);
    for (int q = 0; q < 8; ++q) all[(size_t)me * 8 + q] = m[q];
    good = good && check(hipMemcpyAsync(slots, all.data(), all.size() * 8, hipMemcpyHostToDevice, st_), "hipMemcpy") && comm_->allreduce_sum(slots, G * 8, st_) &&
           check(hipMemcpyAsync(all.data(), slots, all.size() * 8, hipMemcpyDeviceToHost, st_), "hipMemcpy") && check(hipStreamSynchronize(st_), "hipStreamSynchronize");
    HIPCHK(hipEventDestroy(e0));
    HIPCHK(hipEventDestroy(e1));
    cleanup();
    if (!good) return false;
    for (int q = 0; q < 8; ++q) {
        m[q] = 0.0;
        for (int r = 0; r < G; ++r) m[q] = std::max(m[q], all[(size_t)r * 8 + q]);
    }
    const double mb = 1.0 / 1048576.0;
    meas_.exchange_us = m[0];
    meas_.exchange_us_per_mb = std::max(0.0, (m[1] - m[0]) / ((large - small) * 8.0 * mb));
    meas_.allreduce_us = m[2];
    meas_.allgather_us = m[3];
    meas_.allgather_us_per_mb = std::max(0.0, (m[4] - m[3]) / ((agn_large - agn_small) * 8.0 * mb));
    meas_.sweep_floor_us = m[5];
    meas_.sweep_us_per_mb = std::max(1e-6, (m[6] - m[5]) / ((axn_large - axn_small) * 24.0 * mb));
    meas_.valid = true;
    return true;
}

void Engine::tune_comm_schedule(const sparsh_params &p)
{
    if (comm_->size <= 1 || H_.levels.size() < 2 || !measure_transport()) return;
    decide_comm_schedule(p, comm_->size);
    if (p.print_setup && comm_->rank == 0) {
        std::printf("transport measured: exchange %.1f us + %.2f us/MB, all-reduce %.1f us, all-gather %.1f us + %.2f us/MB; device sweep floor %.1f us, %.2f us/MB\n",
                    meas_.exchange_us, meas_.exchange_us_per_mb, meas_.allreduce_us, meas_.allgather_us, meas_.allgather_us_per_mb, meas_.sweep_floor_us, meas_.sweep_us_per_mb);
        std::printf("schedule: %d of %d levels partitioned over %d ranks, %s\n", tuned_repl_level_, (int)H_.levels.size(), comm_->size,
                    tuned_repl_level_ == 0 ? "everything replicated" : (tuned_deep_ ? "deep-halo smoothing" : "one exchange per sweep"));
    }
}

// the decision alone: pure host arithmetic on the hierarchy and on meas_ (also reachable without a device: sparsh_plan_comm_schedule)
void Engine::decide_comm_schedule(const sparsh_params &p, int G)
{
    const int nl = (int)H_.levels.size();
    if (G <= 1 || nl < 2 || !meas_.valid) return;
    const double mb = 1.0 / 1048576.0;
    const int nu = std::max(1, p.sweeps);
    auto sweep_us = [&](double rows, double bpr) { return meas_.sweep_floor_us + rows * bpr * mb * meas_.sweep_us_per_mb; };
    auto exch_us = [&](double bytes) { return meas_.exchange_us + bytes * mb * meas_.exchange_us_per_mb; };
    sched_.assign((size_t)nl, CommLevelChoice());
    std::vector<double> part_deep((size_t)nl, 0.0), part_sweep((size_t)nl, 0.0), repl((size_t)nl, 0.0);
    for (int l = 0; l < nl; ++l) {
        const HostCsr &A = H_.levels[l].A;
        const int n = A.nrow;
        CommLevelChoice &c = sched_[l];
        c.rows = n;
        if (l == nl - 1) continue;  // the coarsest level is solved directly, replicated
        const double bpr = stream_bytes_per_row(A);
        const Partition part = make_partition(n, G);
        const int g = G / 2;
        const int lo = part.lo(g), hi = part.hi(g);
        const int h = boundary_rows(A, lo, hi);
        c.halo_rows = h;
        const double own = (double)(hi - lo);
        // deep halos: nu sweeps of a leg also update the ghost layers <= nu + 1 - s: h nu (nu + 1) / 2 extra row-sweeps per leg
        const double ghost = own > 0 ? (double)h * (nu + 1) / 2.0 : 0.0;
        c.cost_deep_us = 2.0 * nu * sweep_us(own + ghost, bpr) + sweep_us(own, bpr) + 4.0 * exch_us((double)h * (nu + 1) * 8.0);
        c.cost_per_sweep_us = (2.0 * nu + 1.0) * sweep_us(own, bpr) + (2.0 * nu + 2.0) * exch_us((double)h * 8.0);
        c.cost_replicated_us = (2.0 * nu + 1.0) * sweep_us((double)n, bpr);
        part_deep[l] = c.cost_deep_us;
        part_sweep[l] = c.cost_per_sweep_us;
        repl[l] = c.cost_replicated_us;
    }
    const bool deep_allowed = deep_halo_ && p.sweeps >= 1 && !p.precond_fp32;
    double best = -1.0;
    int best_lr = 0;
    bool best_deep = deep_allowed;
    for (int lr = 0; lr < nl; ++lr) {       // levels [0, lr) partitioned
        if (lr > 0 && H_.levels[lr - 1].A.nrow <= 64 * G) break;
        for (int deep = 0; deep < 2; ++deep) {
            if (deep && !deep_allowed) continue;
            if (lr == 0 && deep) continue;
            double t = 0.0;
            for (int l = 0; l < nl - 1; ++l) t += l < lr ? (deep ? part_deep[l] : part_sweep[l]) : repl[l];
            if (lr > 0) t += meas_.allgather_us + (double)H_.levels[lr].A.nrow * 8.0 * mb * meas_.allgather_us_per_mb;
            if (best < 0.0 || t < best) {
                best = t;
                best_lr = lr;
                best_deep = deep != 0;
            }
        }
    }
    tuned_repl_level_ = best_lr;
    tuned_deep_ = best_deep;
    for (int l = 0; l < nl; ++l) {
        sched_[l].partitioned = l < best_lr;
        sched_[l].deep = l < best_lr && best_deep;
    }
}

int Engine::setup(const sparsh_params &p)
{
    prm_ = p;
    ready_ = false;
    fault_ = SPARSH_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        error = "no HIP device visible: the MI355X HIP path is the only compute path (no CPU fallback)";
        return SPARSH_ENODEV;
    }
    if (p.device >= 0) {
        if (!check(hipSetDevice(p.device % ndev), "hipSetDevice")) return SPARSH_ENODEV;
        device_ = p.device % ndev;
    } else {
        (void)hipGetDevice(&device_);
    }
    if (!st_ && !check(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking), "hipStreamCreate")) return SPARSH_ENODEV;
    for (void *q : allocs_) (void)hipFree(q);  // a second setup replaces the resident hierarchy
    allocs_.clear();
    coarse_.release();
    if (!comm_) comm_ = make_self_comm();
    const int G = comm_->size, me = comm_->rank;

    sparsh_params hp = p;
    if (G > 1 && me != 0) hp.print_setup = 0;  // one copy of the "Level k:" lines
    built_locally_ = true;
    image_bytes_ = 0;
    if (int rc = (G > 1 && share_setup_) ? setup_host_shared(hp) : setup_host(hp); rc != SPARSH_OK) return rc;
    prm_ = p;

    // ---- row partition of every level (multi-GPU).  Levels at or below replicate_rows, and always
    // the coarsest one (dense direct solve), are held and computed by every rank -- or, with more than one rank and the
    // tuner on, the levels the measured transport says are cheaper replicated.
    const int nl = (int)H_.levels.size();
    parts_.assign((size_t)nl, Partition());
    repl_level_ = 0;
    tuned_repl_level_ = -1;
    sched_.clear();
    // replicate_rows <= 0 (default): the tuner decides; a positive value is the caller's own threshold, as in round 2
    const int replicate_rows = p.replicate_rows > 0 ? p.replicate_rows : 1500000;
    if (G > 1 && comm_tune_ != 0 && p.replicate_rows <= 0) tune_comm_schedule(p);
    {
        bool repl = (G == 1);
        for (int l = 0; l < nl; ++l) {
            const int n = H_.levels[l].A.nrow;
            if (tuned_repl_level_ >= 0) {
                if (!repl && (l >= tuned_repl_level_ || n <= 64 * G || l == nl - 1)) repl = true;
            } else if (!repl && (n <= std::max(replicate_rows, 64 * G) || l == nl - 1)) {
                repl = true;
            }
            if (repl) {
                parts_[l] = Partition::whole(n, G);
            } else {
                parts_[l] = (l == 0) ? make_partition(n, G) : coarse_partition(H_.levels[l - 1].R, parts_[l - 1]);
                repl_level_ = l + 1;
            }
        }
    }
    dist_ = (G > 1 && repl_level_ > 0);
    if (dist_) gather_part_ = make_partition(H_.levels[repl_level_].A.nrow, G);

    // SPARSH_SETUP_TIMING=1: where the device half of the setup spends its time (each phase synchronised)
    const bool timing = std::getenv("SPARSH_SETUP_TIMING") != nullptr;
    double t_phase = omp_get_wtime();
    auto phase = [&](const char *what) {
        if (!timing) return;
        (void)hipDeviceSynchronize();
        const double t = omp_get_wtime();
        std::printf("setup timing: %-34s %.3f s\n", what, t - t_phase);
        t_phase = t;
    };
    if (timing) std::printf("setup timing: %-34s %.3f s\n", "host hierarchy (+ direct-solver plan)", setup_seconds);
    lev_.assign((size_t)nl, DevLevel());
    int max_blk = 4096;
    for (int l = 0; l < nl; ++l) {
        if (timing && l > 0) phase(("level " + std::to_string(l - 1) + ": layouts + uploads").c_str());
        const HostLevel &h = H_.levels[l];
        DevLevel &d = lev_[l];
        d.nglob = h.A.nrow;
        d.replicated = parts_[l].replicated;
        d.fine = (l == 0);  // finest level: separately named kernel instances (profiling)
        size_t xcap, rcap;
        if (d.replicated) {
            d.n = h.A.nrow;
            if (!upload_csr(*this, h.A, d.A, true)) return SPARSH_ENODEV;
            d.diag = upload(*this, h.diag.data(), (size_t)d.n);
            d.diag_is_const = d.n > 0;
            for (int i = 1; i < d.n && d.diag_is_const; ++i) d.diag_is_const = h.diag[i] == h.diag[0];
            if (d.diag_is_const) d.diag_const = h.diag[0];
            if (l + 1 < nl) {
                if (!upload_csr(*this, h.P, d.P, false) || !upload_csr(*this, h.R, d.R, false)) return SPARSH_ENODEV;
                d.P_is_aggregation = h.P_is_aggregation;
                if (h.P_is_aggregation) {  // (a replicated level: the next one is replicated too)
                    // HEM pairs neighbours along the first grid line on most levels of a lexicographically ordered grid:
                    // aggregate J = rows (2J, 2J+1) (a last single row when n is odd), listed in that order by R
                    const int n = h.A.nrow, nc = h.R.nrow;
                    bool pairs = nc == (n + 1) / 2 && h.R.rowptr[nc] == n && h.P.rowptr[n] == n;
                    for (int J = 0; pairs && J < nc; ++J) pairs = h.R.rowptr[J] == 2 * J;
                    for (int j = 0; pairs && j < n; ++j) pairs = h.R.col[j] == j && h.P.col[j] == j / 2;
                    d.pair_aggregates = pairs;
                    if (!pairs && d.A.box_nx > 0 && n == 2 * nc && h.R.rowptr[nc] == n) {
                        // box-grid level: aggregate J, lexicographic in the coarse box, = point (i, 2j', k) + (i, 2j'+1, k) (axis 1) or
                        // (i, j, 2k') + (i, j, 2k'+1) (axis 2)?
                        const int bx = d.A.box_nx, by = d.A.box_ny, bz = d.A.box_nz;
                        for (int axis = 1; axis <= 2 && d.pair_axis == 0; ++axis) {
                            if ((axis == 1 ? by : bz) % 2 != 0) continue;
                            const int cny = axis == 1 ? by / 2 : by, stride = axis == 1 ? bx : bx * by;
                            for (int rev = 0; rev <= 1 && d.pair_axis == 0; ++rev) {  // (numbered from the near or from the far end of the box)
                                bool ok = true;
                                for (int J = 0; ok && J < nc; ++J) {
                                    const int i = J % bx, t = J / bx, cj = t % cny, ck = t / cny;
                                    const int f1 = i + bx * ((axis == 1 ? 2 * cj : cj) + by * (axis == 2 ? 2 * ck : ck));
                                    const int j0 = h.R.rowptr[rev ? nc - 1 - J : J];
                                    ok = h.R.rowptr[(rev ? nc - 1 - J : J) + 1] - j0 == 2 && h.R.col[j0] == f1 && h.R.col[j0 + 1] == f1 + stride;
                                }
                                if (ok) d.pair_axis = rev ? -axis : axis;
                            }
                        }
                    }
                    if (!pairs && nc > 0) {  // aggregates of one or two rows (pairwise matching): their rows, for the fused prolongation
                        std::vector<int> mem((size_t)2 * nc, -1);
                        bool two = h.R.rowptr[nc] == n && h.P.rowptr[n] == n;
                        for (int J = 0; two && J < nc; ++J) {
                            const int j0 = h.R.rowptr[J], len = h.R.rowptr[J + 1] - j0;
                            two = len == 1 || len == 2;
                            if (!two) break;
                            mem[(size_t)2 * J] = h.R.col[j0];
                            if (len == 2) mem[(size_t)2 * J + 1] = h.R.col[j0 + 1];
                            for (int q = 0; q < len; ++q) two = two && h.P.col[h.R.col[j0 + q]] == J;
                        }
                        if (two) d.members = upload(*this, mem.data(), mem.size());
                    }
                }
            }
            xcap = rcap = (size_t)d.n;
        } else {
            const Partition &pl = parts_[l];
            d.n = pl.hi(me) - pl.lo(me);
            d.deep = deep_halo_ && prm_.sweeps >= 1 && !prm_.precond_fp32 && (tuned_repl_level_ < 0 || tuned_deep_);
            if (d.deep) {
                // deep-halo layout: own rows + K = sweeps + 1 ghost layers; exchange plans of depth 1 (SpMV-type
                // calls outside a smoothing leg), K-1 (right-hand side of a leg) and K (iterate of a leg)
                d.K = prm_.sweeps + 1;
                DeepLocal dl = extract_local_deep(h.A, pl, me, d.K, {1, d.K - 1, d.K});
                d.npad = dl.npad;
                d.layer_end = dl.layer_end;
                {
                    std::vector<int> cuts;  // own rows | padding | layer 1 | ... : prefix launches end exactly there
                    cuts.push_back(d.n);
                    for (int q = 0; q + 1 < d.K; ++q) cuts.push_back(dl.layer_end[q]);
                    if (!upload_csr(*this, dl.M, d.A, true, &cuts, &d.blk_first, &d.wblk_first)) return SPARSH_ENODEV;
                }
                for (int q = 0; q < 3; ++q)
                    if (!upload_deep_plan(dl.plans[q], d.dplan[q])) return SPARSH_ENODEV;
                std::vector<double> dg((size_t)dl.M.nrow, 1.0);  // padding rows: 1 (never divides anything but 0)
                for (int r = 0; r < dl.M.nrow; ++r)
                    if (dl.global_of[r] >= 0) dg[r] = h.diag[dl.global_of[r]];
                d.diag = upload(*this, dg.data(), dg.size());
            } else {
            LocalOp la = extract_local(h.A, pl, pl, me);
            if (!upload_csr(*this, la.M, d.A, true) || !upload_plan(la.plan, d.planA)) return SPARSH_ENODEV;
            {  // slices whose rows reference no halo column can run while the exchange is in flight
                const int nloc = la.plan.nloc, nsl = (la.M.nrow + 63) / 64;
                std::vector<int> il, bl;
                for (int sl = 0; sl < nsl; ++sl) {
                    bool touches = false;
                    const int r1 = std::min(la.M.nrow, (sl + 1) * 64);
                    for (int j = la.M.rowptr[sl * 64]; j < la.M.rowptr[r1] && !touches; ++j) touches = la.M.col[j] >= nloc;
                    (touches ? bl : il).push_back(sl);
                }
                d.A.nint = (int)il.size();
                d.A.nbnd = (int)bl.size();
                if (!il.empty()) d.A.int_list = upload(*this, il.data(), il.size());
                if (!bl.empty()) d.A.bnd_list = upload(*this, bl.data(), bl.size());
            }
            d.diag = upload(*this, h.diag.data() + pl.lo(me), (size_t)d.n);
            }
            // P_l: my fine rows, columns in the coarse space (replicated coarse space: global columns)
            LocalOp lp = extract_local(h.P, pl, parts_[l + 1], me);
            // R_l: my share of the coarse rows; on the boundary to the replicated levels the share is
            // the balanced gather partition and the result is all-gathered
            const Partition &rrows = parts_[l + 1].replicated ? gather_part_ : parts_[l + 1];
            LocalOp lr = extract_local(h.R, rrows, pl, me);
            if (!upload_csr(*this, lp.M, d.P, false) || !upload_plan(lp.plan, d.planP)) return SPARSH_ENODEV;
            if (!upload_csr(*this, lr.M, d.R, false) || !upload_plan(lr.plan, d.planR)) return SPARSH_ENODEV;
            d.P_is_aggregation = h.P_is_aggregation;
            int xh = d.planA.nhalo;
            if (l > 0 && !lev_[l - 1].replicated) xh = std::max(xh, lev_[l - 1].planP.nhalo);  // x_l is also P_{l-1}'s input
            xcap = (size_t)d.n + xh;
            rcap = (size_t)d.n + d.planR.nhalo;
            if (d.deep) {
                xcap = (size_t)d.layer_end[d.K];                                   // own + padding + all ghost layers
                rcap = std::max((size_t)d.npad, (size_t)d.n + d.planR.nhalo) + 64;  // residual over the own slices, then R's halo
                // the prolongation reads the coarse iterate through a staging vector of its own: [own | P's halo]
                d.xc_stage = static_cast<double *>(dalloc(((size_t)d.planP.nloc + d.planP.nhalo + 64) * 8));
                if (!d.xc_stage) return SPARSH_ENODEV;
                if (l == 0) {
                    d.b_ext = static_cast<double *>(dalloc(xcap * 8));
                    if (!d.b_ext || !check(hipMemsetAsync(d.b_ext, 0, xcap * 8, st_), "hipMemsetAsync")) return SPARSH_ENODEV;
                }
            }
        }
        d.x = static_cast<double *>(dalloc(xcap * 8));
        d.x2 = static_cast<double *>(dalloc(xcap * 8));
        d.r = static_cast<double *>(dalloc(rcap * 8));
        if (l > 0) {
            const size_t bcap = d.deep ? xcap : (size_t)d.n;
            d.b = static_cast<double *>(dalloc(bcap * 8));
            if (d.b && !check(hipMemsetAsync(d.b, 0, bcap * 8, st_), "hipMemsetAsync")) return SPARSH_ENODEV;
        }
        if (!d.diag || !d.x || !d.x2 || !d.r || (l > 0 && !d.b)) return SPARSH_ENODEV;
        if (l + 1 < nl) max_blk = std::max(max_blk, std::max(partial_count(d.P), partial_count(d.R)));
        max_blk = std::max(max_blk, partial_count(d.A));
        if (!check(hipMemsetAsync(d.x, 0, xcap * 8, st_), "hipMemsetAsync")) return SPARSH_ENODEV;
        if (!check(hipMemsetAsync(d.x2, 0, xcap * 8, st_), "hipMemsetAsync")) return SPARSH_ENODEV;
        if (!check(hipMemsetAsync(d.r, 0, rcap * 8, st_), "hipMemsetAsync")) return SPARSH_ENODEV;
    }
    phase("last level: layouts + uploads");
    nL_ = H_.nL;
    if (H_.coarse_dense) {
        if (!coarse_.setup_dense(nL_, H_.coarse_inverse.data(), error)) return SPARSH_ENODEV;
        std::vector<double>().swap(H_.coarse_inverse);  // host copy no longer needed
    } else {
        // large coarsest level (the reference's level1 = 6 policy at >= ~260 k rows): block-tridiagonal
        // factorisation on the device
        const double t_f = omp_get_wtime();
        int why = 0;
        const bool okc = coarse_.form() == 1 ? coarse_.setup_bt(H_.levels.back().A, st_, error, &why) : coarse_.setup_nd(H_.levels.back().A, st_, error, &why);
        if (!okc) return why == 1 ? SPARSH_EINVAL : (why == 2 ? SPARSH_ENUMERIC : SPARSH_ENODEV);
        setup_seconds += omp_get_wtime() - t_f;
        if (p.print_setup && (G == 1 || me == 0)) {
            if (coarse_.nested())
                std::printf("coarsest level: %d rows, nested dissection: %d nodes on %d levels (largest pivot block %d), factors %.1f MB in HBM, %d launches per solve, %.2f s\n",
                            nL_, coarse_.nd().nnodes(), coarse_.nd().nlevels(), coarse_.nd().max_pivot_rows(), coarse_.bytes() / 1e6,
                            coarse_.nd().launches_per_solve(), coarse_.factor_seconds);
            else
                std::printf("coarsest level: %d rows, RCM bandwidth %d -> %d blocks of %d, factors %.1f MB in HBM, %.2f s\n", nL_, coarse_.bandwidth(),
                            coarse_.nblocks(), coarse_.block(), coarse_.bytes() / 1e6, coarse_.factor_seconds);
        }
    }

    part_cap_ = max_blk + 8;
    part0_ = static_cast<double *>(dalloc((size_t)part_cap_ * 8));
    part1_ = static_cast<double *>(dalloc((size_t)part_cap_ * 8));
    scal_ = static_cast<double *>(dalloc(S_COUNT * 8));
    hist_cap_dev_ = std::max(1024, p.max_iter + 2);
    hist_cap_dev_ = std::min(hist_cap_dev_, 1 << 22);
    hist_dev_ = static_cast<double *>(dalloc((size_t)hist_cap_dev_ * 8));
    iter_ctr_ = static_cast<int *>(dalloc(sizeof(int)));
    if (!part0_ || !part1_ || !scal_ || !hist_dev_ || !iter_ctr_) return SPARSH_ENODEV;
    if (!check(hipMemsetAsync(scal_, 0, S_COUNT * 8, st_), "hipMemsetAsync")) return SPARSH_ENODEV;
    if (!check(hipMemsetAsync(iter_ctr_, 0, sizeof(int), st_), "hipMemsetAsync")) return SPARSH_ENODEV;
    drop_graph();  // a captured iteration refers to the buffers of the previous setup
    if (!pinned_ && !check(hipHostMalloc(reinterpret_cast<void **>(&pinned_), 64 * sizeof(double), hipHostMallocDefault), "hipHostMalloc"))
        return SPARSH_ENODEV;
    work_.clear();
    // Krylov vectors may be SpMV inputs: room for the halo (deep layout: padding + all ghost layers)
    const size_t wcap = lev_[0].deep ? (size_t)lev_[0].layer_end[lev_[0].K] : (size_t)lev_[0].n + lev_[0].planA.nhalo;
    for (int k = 0; k < 8; ++k) {
        double *w = static_cast<double *>(dalloc(wcap * 8));
        if (!w) return SPARSH_ENODEV;
        if (!check(hipMemsetAsync(w, 0, wcap * 8, st_), "hipMemsetAsync")) return SPARSH_ENODEV;
        work_.push_back(w);
    }
    place_tried = 0;
    place_best_us = place_worst_us = place_first_us = 0.0;
    phase("coarsest-level factorisation + workspace");
    tune_box_kernels();
    if (G == 1 && cfg_.place_search) tune_placement();
    phase("placement search");
    f32_ready_ = false;
    if (p.precond_fp32 && !setup_f32()) return SPARSH_EINVAL;
    if (dist_ && !st2_) {
        if (!check(hipStreamCreateWithFlags(&st2_, hipStreamNonBlocking), "hipStreamCreate") ||
            !check(hipEventCreateWithFlags(&ev_ready_, hipEventDisableTiming), "hipEventCreate") ||
            !check(hipEventCreateWithFlags(&ev_halo_, hipEventDisableTiming), "hipEventCreate"))
            return SPARSH_ENODEV;
    }
    if (!check(hipStreamSynchronize(st_), "setup sync")) return SPARSH_ENODEV;
    if (!comm_->barrier(st_)) {
        error = "comm barrier after setup failed: " + comm_->error;
        return SPARSH_ECOMM;
    }
    ready_ = true;
    return SPARSH_OK;
}

// Float copy of the hierarchy for the opt-in fp32 preconditioner (values converted on the device).
bool Engine::setup_f32()
{
    const int nl = (int)lev_.size();
    if (dist_) {
        error = "precond_fp32 is implemented for one GPU";
        return false;
    }
    if (nl < 2) return true;  // single level: the "V-cycle" is the direct solve, nothing to gain
    // levels with the sliced-diagonal mirror get a float copy of its value blocks (4 B per stored entry);
    // every other level (unstructured operators, small coarse levels) a float copy of its CSR values
    // (8 B per entry with the int32 column index) run by a row-per-thread kernel
    f32_.assign((size_t)nl, F32Level());
    for (int l = 0; l < nl; ++l) {
        DevLevel &d = lev_[l];
        F32Level &f = f32_[l];
        const size_t n = (size_t)d.n;
        f.x = static_cast<float *>(dalloc(n * 4));
        f.b = static_cast<float *>(dalloc(n * 4));
        if (!f.x || !f.b) return false;
        if (l + 1 < nl) {
            f.x2 = static_cast<float *>(dalloc(n * 4));
            f.r = static_cast<float *>(dalloc(n * 4));
            f.diag = static_cast<float *>(dalloc(n * 4));
            if (!f.x2 || !f.r || !f.diag) return false;
            launch_cvt_d2f((long)n, d.diag, f.diag, st_);
            if (d.A.has_sdia()) {
                f.A.nrow = d.n;
                f.A.nslice = d.A.nslice;
                f.A.sd_ptr = d.A.sd_ptr;
                f.A.sd_off = d.A.sd_off;
                f.A.sd_mask = d.A.sd_mask;
                f.A.slots = d.A.sd_slots;
                f.A.val = static_cast<float *>(dalloc((size_t)d.A.sd_slots * 64 * 4));
                if (!f.A.val) return false;
                launch_cvt_d2f((long)d.A.sd_slots * 64, d.A.sd_val, f.A.val, st_);
            } else {
                f.csr_val = static_cast<float *>(dalloc((size_t)std::max(d.A.nnz, 1) * 4));
                if (!f.csr_val) return false;
                launch_cvt_d2f((long)d.A.nnz, d.A.val, f.csr_val, st_);
            }
        }
    }
    if (coarse_.dense()) {
        coarse_inv_f32_ = static_cast<float *>(dalloc((size_t)nL_ * nL_ * 4));
        if (!coarse_inv_f32_) return false;
        launch_cvt_d2f((long)nL_ * nL_, coarse_.dense_inverse(), coarse_inv_f32_, st_);
    } else {  // block-tridiagonal factors stay fp64: b_L and x_L are converted around the solve
        coarse_inv_f32_ = nullptr;
        coarse_tmp_ = static_cast<double *>(dalloc((size_t)nL_ * 2 * 8));
        if (!coarse_tmp_) return false;
    }
    f32_ready_ = true;
    return true;
}

void Engine::vcycle_f32(const double *r64, double *z64, double *partial, int *nblk)
{
    const int last = (int)lev_.size() - 1;
    const int nu = prm_.sweeps;
    const float w = (float)prm_.omega;
    launch_cvt_d2f((long)lev_[0].n, r64, f32_[0].b, st_);
    auto apply = [&](int l, CsrOp op, const float *x, float *y) {
        F32Level &f = f32_[l];
        if (f.A.val)
            launch_sdia_f32(f.A, op, x, f.b, y, w, st_);
        else
            launch_csr_f32(lev_[l].A, f.csr_val, op, f.diag, x, f.b, y, w, st_);
    };
    auto sweeps = [&](int l, int count) {
        F32Level &f = f32_[l];
        for (int k = 0; k < count; ++k) {
            apply(l, OP_JACOBI, f.x, f.x2);
            std::swap(f.x, f.x2);
        }
    };
    for (int l = 0; l < last; ++l) {
        F32Level &f = f32_[l];
        launch_jacobi_zero_f32(lev_[l].n, f.b, f.diag, w, f.x, st_);
        sweeps(l, nu - 1);
        apply(l, OP_RESID, f.x, f.r);
        launch_restrict_f32(lev_[l + 1].n, lev_[l].R.rowptr, lev_[l].R.col, lev_[l].R.val, f.r, f32_[l + 1].b, st_);
    }
    if (coarse_inv_f32_) {
        launch_gemv_f32(nL_, coarse_inv_f32_, f32_[last].b, f32_[last].x, st_);
    } else {
        launch_cvt_f2d(nL_, f32_[last].b, coarse_tmp_, st_);
        coarse_.solve(coarse_tmp_, coarse_tmp_ + nL_, st_);
        launch_cvt_d2f(nL_, coarse_tmp_ + nL_, f32_[last].x, st_);
    }
    for (int l = last; l > 0; --l) {
        DevLevel &F = lev_[l - 1];
        if (F.P_is_aggregation)
            launch_prolong_agg_f32(F.n, F.P.col, f32_[l].x, f32_[l - 1].x, st_);
        else
            launch_prolong_csr_f32(F.n, F.P.rowptr, F.P.col, F.P.val, f32_[l].x, f32_[l - 1].x, st_);
        sweeps(l - 1, nu);
    }
    launch_cvt_f2d_dot(lev_[0].n, f32_[0].x, r64, z64, partial, nblk, st_);
}

bool Engine::op_precond_f32(const double *r, double *z)
{
    if (!f32_ready_) {
        error = "the fp32 preconditioner hierarchy was not built (sparsh_params.precond_fp32 = 0, or a single level)";
        return false;
    }
    int nb = 0;
    vcycle_f32(r, z, part0_, &nb);
    return true;
}

double Engine::read_scalar(int slot)
{
    HIPCHK(hipMemcpyAsync(pinned_, scal_ + slot, sizeof(double), hipMemcpyDeviceToHost, st_));
    HIPCHK(hipStreamSynchronize(st_));
    return pinned_[0];
}

double Engine::read_hist(int it)
{
    HIPCHK(hipMemcpyAsync(pinned_, hist_dev_ + it, sizeof(double), hipMemcpyDeviceToHost, st_));
    HIPCHK(hipStreamSynchronize(st_));
    return pinned_[0];
}

// Halo exchange of one operator's input vector: pack what the peers need, then the transport
// fills vec[nloc .. nloc+nhalo).  One GPU: nothing to do.
bool Engine::halo(const DevPlan &p, double *vec)
{
    if (!dist_) return true;
    if (!p.empty()) ++n_exchanges_;
    return note_comm(comm_->exchange(p, vec, st_), "halo exchange");
}

double Engine::bench_comm(int what, int level, int reps)
{
    if (!dist_ || reps <= 0 || level < 0 || level >= (int)lev_.size()) return -1.0;
    DevLevel &L = lev_[level];
    if (what == 0 && L.replicated) return -1.0;
    if (what == 2 && (repl_level_ <= 0 || repl_level_ >= (int)lev_.size())) return -1.0;
    auto step = [&]() {
        if (what == 0 && L.deep) return deep_exchange(L, 2, L.x);
        if (what == 0) return comm_->exchange(L.planA, L.x, st_);
        if (what == 1) return comm_->allreduce_sum(scal_ + S_SUM0, 2, st_);
        return comm_->allgather(lev_[repl_level_].r, gather_part_, st_);  // r of that level is scratch here
    };
    for (int i = 0; i < 3; ++i)
        if (!step()) return -1.0;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, st_));
    bool ok = true;
    for (int i = 0; i < reps && ok; ++i) ok = step();
    HIPCHK(hipEventRecord(e1, st_));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    HIPCHK(hipEventDestroy(e0));
    HIPCHK(hipEventDestroy(e1));
    return ok ? ms * 1e-3 / reps : -1.0;
}

int Engine::apply_A(DevLevel &L, CsrOp op, CsrArgs a)
{
    double *xin = const_cast<double *>(a.x);
    if (L.deep) {  // outside a smoothing leg: refresh the first ghost layer of the input, run the own slices
        if (!deep_exchange(L, 0, xin)) return 0;
        return launch_prefix(L, L.n, op, a);
    }
    if ((op == OP_SPMV_DOT || op == OP_JACOBI_DOT || op == OP_JACOBI_PROLONG || op == OP_JACOBI) && (!dist_ || L.replicated) && !a.slice_list &&
        box1_applies(L.A, cfg_)) {
        // box-grid level: the launches that carry an epilogue (and the odd plain sweep of a leg) run the plane-marching kernel: every x read once
        return launch_box1(L.A, op == OP_SPMV_DOT ? 0 : (op == OP_JACOBI_DOT ? 1 : (op == OP_JACOBI ? 4 : 3)), a, L.fine, st_);
    }
    const CsrFamily fam = csr_family(L.A, cfg_);
    const bool sliced = fam == FAM_SDIA || fam == FAM_SDIA_TAB || fam == FAM_SELL;
    if (!(dist_ && overlap_ && !L.replicated && sliced && L.A.nint > 0 && L.A.nbnd > 0 && st2_)) {
        if (!halo(L.planA, xin)) return 0;  // transport failed: sticky fault, nothing launched on stale data
        return launch_csr(L.A, op, a, L.fine, st_, cfg_);
    }
    // overlap: [st2] wait until x is final -> pack + exchange ; [st] interior slices meanwhile ;
    //          [st] wait for the halo -> boundary slices
    HIPCHK(hipEventRecord(ev_ready_, st_));
    HIPCHK(hipStreamWaitEvent(st2_, ev_ready_, 0));
    const DevPlan &p = L.planA;
    note_comm(comm_->exchange(p, xin, st2_), "halo exchange");
    HIPCHK(hipEventRecord(ev_halo_, st2_));
    a.slice_list = L.A.int_list;
    a.nlist = L.A.nint;
    a.partial_off = 0;
    const int n1 = launch_csr(L.A, op, a, L.fine, st_, cfg_);
    HIPCHK(hipStreamWaitEvent(st_, ev_halo_, 0));
    a.slice_list = L.A.bnd_list;
    a.nlist = L.A.nbnd;
    a.partial_off = n1;
    const int n2 = launch_csr(L.A, op, a, L.fine, st_, cfg_);
    return n1 + n2;
}

// Reduce per-workgroup partials and update the device scalars; across ranks the local sums are
// all-reduced in between (one 16-byte ncclAllReduce).
void Engine::finalize(Fin code, const double *p0, const double *p1, int nblk, int slot, double *hist, int it, int nblk1)
{
    if (!dist_) {
        launch_finalize(code, p0, p1, nblk, scal_, slot, hist, it, st_, 0, iter_ctr_, hist_cap_dev_, nblk1);
        return;
    }
    launch_finalize(code, p0, p1, nblk, scal_, slot, hist, it, st_, 1, iter_ctr_, hist_cap_dev_, nblk1);
    note_comm(comm_->allreduce_sum(scal_ + S_SUM0, 2, st_), "allreduce");
    launch_finalize(code, p0, p1, nblk, scal_, slot, hist, it, st_, 2, iter_ctr_, hist_cap_dev_, nblk1);
}

// ---------------------------------------------------------------------------- operators
// Inputs of SpMV-type operators must have room for the operator's halo behind their own entries
// (all engine-owned vectors do); on one GPU the plans are empty.

void Engine::op_spmv(int l, const double *x, double *y)
{
    CsrArgs a;
    a.x = x;
    a.y = y;
    apply_A(lev_[l], OP_SPMV, a);
}

void Engine::op_residual(int l, const double *b, const double *x, double *r)
{
    CsrArgs a;
    a.x = x;
    a.b = b;
    a.y = r;
    apply_A(lev_[l], OP_RESID, a);
}

void Engine::op_residual_restrict(int l, const double *b, const double *x, double *bc, double *xc)
{
    if (!lev_[l].pair_aggregates && lev_[l].pair_axis != 0) {
        launch_box_resid_pair(lev_[l].A, lev_[l].pair_axis, x, b, diag_stream(lev_[l + 1]), lev_[l + 1].diag_const, prm_.omega, bc, xc, st_);
        return;
    }
    CsrArgs a;
    a.x = x;
    a.b = b;
    a.y = bc;
    a.y2 = xc;
    a.d = diag_stream(lev_[l + 1]);
    a.dconst = lev_[l + 1].diag_const;
    a.omega = prm_.omega;
    if (box1_applies(lev_[l].A, cfg_) && lev_[l].A.box_nx % 2 == 0) {  // (an even line length: every row pair lies within one line)
        launch_box1(lev_[l].A, 2, a, lev_[l].fine, st_);
        return;
    }
    launch_resid_pair(lev_[l].A, a, lev_[l].fine, st_, cfg_);
}

void Engine::op_jacobi_prolong(int l, const double *b, const double *x, double *xf)
{
    DevLevel &F = lev_[l - 1];
    CsrArgs a;
    a.x = x;
    a.b = b;
    a.d = lev_[l].diag;
    a.omega = prm_.omega;
    a.y2 = xf;
    a.members = F.pair_aggregates ? nullptr : F.members;
    a.nfine = F.n;
    apply_A(lev_[l], OP_JACOBI_PROLONG, a);
}

double Engine::op_resnorm(int l, const double *b, const double *x)
{
    CsrArgs a;
    a.x = x;
    a.b = b;
    a.partial = part0_;
    const int np = apply_A(lev_[l], OP_RESNORM, a);
    finalize(FIN_SQRT, part0_, nullptr, np, S_RES, nullptr, 0);
    return read_scalar(S_RES);
}

bool Engine::op_restrict(int l, const double *r, double *bc, bool fuse_zero)
{
    DevLevel &L = lev_[l];
    halo(L.planR, const_cast<double *>(r));
    CsrArgs a;
    a.x = r;
    const bool gather = dist_ && !L.replicated && lev_[l + 1].replicated;
    a.y = gather ? bc + gather_part_.lo(comm_->rank) : bc;  // my share of the replicated level's rhs
    bool fused = false;
    if (L.P_is_aggregation && fuse_zero && !gather && l + 2 < (int)lev_.size() && lev_[l + 1].n == L.R.nrow) {
        // the coarse level's first pre-sweep from a zero guess rides along (one launch less per level)
        launch_restrict_agg_zero(L.R.nrow, L.R.rowptr, L.R.col, r, a.y, diag_stream(lev_[l + 1]), lev_[l + 1].diag_const, prm_.omega, lev_[l + 1].x, st_);
        fused = true;
    } else if (L.P_is_aggregation) {
        launch_restrict_agg(L.R.nrow, L.R.rowptr, L.R.col, r, a.y, st_);
    } else {
        launch_csr(L.R, OP_SPMV, a, false, st_, cfg_);
    }
    if (gather) note_comm(comm_->allgather(bc, gather_part_, st_), "allgather");
    return fused;
}

void Engine::op_prolong(int l, const double *xc, double *xf)
{
    DevLevel &L = lev_[l];
    if (L.deep && !lev_[l + 1].replicated) {
        // the coarse iterate's own tail holds A_{l+1}'s ghost layers: P reads [own | its halo] from a staging vector
        launch_copy(L.planP.nloc, xc, L.xc_stage, st_);
        xc = L.xc_stage;
    }
    halo(L.planP, const_cast<double *>(xc));
    if (L.P_is_aggregation) {
        launch_prolong_agg(L.n, L.P.col, xc, xf, st_);
    } else {
        CsrArgs a;
        a.x = xc;
        a.y = xf;
        launch_csr(L.P, OP_ADD, a, false, st_, cfg_);
    }
}

void Engine::op_coarse(const double *b, double *x) { coarse_.solve(b, x, st_); }

double Engine::op_dot(int n, const double *x, const double *y)
{
    int nb = 0;
    launch_dot(n, x, y, part0_, &nb, st_);
    finalize(FIN_STORE, part0_, nullptr, nb, S_TMP, nullptr, 0);
    return read_scalar(S_TMP);
}

// `sweeps` fused Jacobi sweeps on level buffers; the current iterate is L.x on entry and exit
// (the ping-pong partner L.x2 is scratch).  parallel::jacobi_smoother, src/AMG_smoothers.cpp:53-76.
void Engine::smooth(DevLevel &L, const double *b, int sweeps, bool x_zero, double *dot_partial, int *dot_nblk, bool zero_done,
                    DevLevel *prolong_to)
{
    if (L.deep) {
        // Deep-halo leg: no exchange inside.  On entry b is valid on the layers <= K-1 and (unless x = 0) x on the
        // layers <= K; sweep s leaves the layers <= K-s valid, so after `sweeps` <= K-1 sweeps the own rows and the
        // first ghost layer hold exactly the global iterate.
        const int K = L.K;
        int s = 0;
        const bool timed = prof.enabled && &L == &lev_[0] && prof.used + 2 <= prof.ev.size();
        int in_run = 0;
        if (x_zero && sweeps > 0) {
            if (!zero_done) launch_jacobi_zero(L.layer_end[K - 1], b, L.diag, 0.0, prm_.omega, L.x, st_);
            s = 1;
        }
        if (timed) HIPCHK(hipEventRecord(prof.ev[prof.used], st_));
        for (; s < sweeps; ++s) {
            CsrArgs a;
            a.x = L.x;
            a.b = b;
            a.d = L.diag;
            a.y = L.x2;
            a.omega = prm_.omega;
            a.reverse = csr_alternates(L.A, cfg_) && (in_run & 1) == 0;  // first sweep of the run against the (ascending) kernel before it, then alternating
            const int depth_left = std::max(K - (s + 1), 0);  // this is sweep s+1: it updates the layers <= K-(s+1)
            launch_prefix(L, L.layer_end[depth_left], OP_JACOBI, a);
            ++in_run;
            std::swap(L.x, L.x2);
        }
        if (timed) {
            HIPCHK(hipEventRecord(prof.ev[prof.used + 1], st_));
            prof.run_launches.push_back(in_run);
            prof.used += 2;
        }
        if (dot_partial) launch_dot(L.n, L.x, b, dot_partial, dot_nblk, st_);  // own rows only
        return;
    }
    int k = 0;
    bool dot_done = false;
    const bool special_last = dot_partial || prolong_to;  // the last sweep carries an epilogue of its own
    if (x_zero && sweeps > 0 && zero_start(L) && sweeps - (special_last ? 1 : 0) >= 3) {
        // zero guess on a double-sweep level: sweeps 1 - 3 in one launch that reads b alone (whatever zero-guess sweep a producer left in x is not used)
        launch_box2(L.A, nullptr, b, L.x, prm_.omega, L.fine, st_, true);
        k = 3;
    } else if (x_zero && sweeps > 0) {
        if (!zero_done) launch_jacobi_zero(L.n, b, diag_stream(L), L.diag_const, prm_.omega, L.x, st_);
        k = 1;
    }
    const bool timed = prof.enabled && &L == &lev_[0] && k < sweeps && prof.used + 2 <= prof.ev.size();
    if (timed) HIPCHK(hipEventRecord(prof.ev[prof.used], st_));
    int in_run = 0;
    const bool pairs = (!dist_ || L.replicated) && box2_applies(L.A, cfg_);
    bool timed_open = timed;
    for (; k < sweeps; ++k) {
        const bool last = (k == sweeps - 1);
        if (pairs && k + 2 <= sweeps - (special_last ? 1 : 0)) {  // two plain sweeps in one pass over the vectors
            launch_box2(L.A, L.x, b, L.x2, prm_.omega, L.fine, st_);
            ++k;
            ++in_run;
            std::swap(L.x, L.x2);
            // profiling a level that runs double sweeps: the timed run is its double-sweep launches alone
            if (timed_open && !(k + 3 <= sweeps - (special_last ? 1 : 0))) {
                HIPCHK(hipEventRecord(prof.ev[prof.used + 1], st_));
                prof.run_launches.push_back(in_run);
                prof.used += 2;
                timed_open = false;
            }
            continue;
        }
        CsrArgs a;
        a.x = L.x;
        a.b = b;
        a.d = L.diag;
        a.y = L.x2;
        a.omega = prm_.omega;
        a.reverse = csr_alternates(L.A, cfg_) && (in_run & 1) == 0;  // first sweep of the run against the (ascending) kernel before it, then alternating
        CsrOp op = OP_JACOBI;
        if (last && dot_partial) {
            op = OP_JACOBI_DOT;
            a.partial = dot_partial;
            dot_done = true;
        } else if (last && prolong_to) {  // transfer_solution rides in the epilogue: this level's iterate is dead after the leg
            op = OP_JACOBI_PROLONG;
            a.y2 = prolong_to->x;
            a.members = prolong_to->pair_aggregates ? nullptr : prolong_to->members;
            a.nfine = prolong_to->n;
        }
        const int np = apply_A(L, op, a);
        if (op == OP_JACOBI_DOT) *dot_nblk = np;
        ++in_run;
        std::swap(L.x, L.x2);
    }
    if (timed_open && !pairs) {
        HIPCHK(hipEventRecord(prof.ev[prof.used + 1], st_));
        prof.run_launches.push_back(in_run);
        prof.used += 2;
    }
    if (dot_partial && !dot_done) launch_dot(L.n, L.x, b, dot_partial, dot_nblk, st_);
}

void Engine::op_jacobi(int l, const double *b, double *x, double *tmp, int sweeps, bool x_is_zero)
{
    // run on caller buffers by temporarily borrowing the level's ping-pong slots
    DevLevel &L = lev_[l];
    double *sx = L.x, *sx2 = L.x2;
    L.x = x;
    L.x2 = tmp;
    smooth(L, b, sweeps, x_is_zero, nullptr, nullptr);
    if (L.x != x) launch_copy(L.n, L.x, x, st_);
    L.x = sx;
    L.x2 = sx2;
}

// One V(nu,nu) cycle (body of the while loops in AMG_solve_jacobi, src/AMG_phases.cpp:198-216).
void Engine::vcycle(const double *b0, bool x0_zero, double *dot_partial, int *dot_nblk, bool zero_done0)
{
    const int last = (int)lev_.size() - 1;
    const int nu = prm_.sweeps;
    lev_[0].b = const_cast<double *>(b0);
    if (last == 0) {  // single level: the "coarsest" direct solve is the whole cycle
        op_coarse(b0, lev_[0].x);
        if (dot_partial) launch_dot(lev_[0].n, lev_[0].x, b0, dot_partial, dot_nblk, st_);
        return;
    }
    bool zero_done = zero_done0 && x0_zero;  // the previous level's restriction (level 0: the caller) already wrote this level's zero-guess sweep
    for (int l = 0; l < last; ++l) {
        DevLevel &L = lev_[l];
        if (L.deep) {
            // deep-halo leg: one exchange of the right-hand side's ghost layers (and of the iterate's, unless it
            // is zero) replaces the exchange in front of every sweep and of the residual
            if (l == 0) {
                launch_copy(L.n, b0, L.b_ext, st_);
                L.b = L.b_ext;
            }
            deep_exchange(L, 1, L.b);
            const bool xz = l > 0 || x0_zero;
            if (!xz) deep_exchange(L, 2, L.x);
            smooth(L, L.b, nu, xz, nullptr, nullptr, false);
            CsrArgs a;  // residual on the own slices: x is valid on the first ghost layer
            a.x = L.x;
            a.b = L.b;
            a.y = L.r;
            launch_prefix(L, L.n, OP_RESID, a);
            zero_done = false;
            op_restrict(l, L.r, lev_[l + 1].b, false);
            continue;
        }
        smooth(L, L.b, nu, l > 0 || x0_zero, nullptr, nullptr, zero_done);  // coarse levels start from x = 0 (fill, :204)
        // (a level that starts its leg with the three-sweep launch reads b alone: nobody needs its zero-guess sweep written)
        const bool next_from_b = l + 1 < last && nu >= 3 && zero_start(lev_[l + 1]);
        if (level_paired(l)) {
            // store_residual + transfer_residual (+ x_{l+1} = omega*b/d) in one launch: r_l never goes to memory
            op_residual_restrict(l, L.b, L.x, lev_[l + 1].b, next_from_b ? nullptr : lev_[l + 1].x);
            zero_done = true;
            continue;
        }
        op_residual(l, L.b, L.x, L.r);                                      // store_residual
        zero_done = op_restrict(l, L.r, lev_[l + 1].b, nu > 0 && !next_from_b);  // transfer_residual (+ x_{l+1} = omega*b/d)
    }
    bool coarse_prolonged = false;
    {
        DevLevel &F = lev_[last - 1];
        if (coarse_.nested() && cfg_.fuse_prolong && (!dist_ || F.replicated) && !F.deep && F.P_is_aggregation && (F.pair_aggregates || F.members) &&
            F.R.nrow == lev_[last].n && !prm_.precond_fp32) {
            // Direct_Solver_Pardiso_solve + transfer_solution: the backward pass of the nested-dissection solve adds x_L to the rows it owns
            NdProlong pr;
            pr.xf = F.x;
            pr.members = F.pair_aggregates ? nullptr : F.members;
            pr.nfine = F.n;
            coarse_.solve_prolong(lev_[last].b, lev_[last].x, st_, pr);
            coarse_prolonged = true;
        } else {
            op_coarse(lev_[last].b, lev_[last].x);  // Direct_Solver_Pardiso_solve
        }
    }
    for (int l = last; l > 0; --l) {
        DevLevel &F = lev_[l - 1];
        if (!(l == last ? coarse_prolonged : level_prolong_fused(l))) op_prolong(l - 1, lev_[l].x, F.x);  // transfer_solution (else: done by the launch before)
        if (F.deep) deep_exchange(F, 2, F.x);  // the leg's only exchange: K ghost layers of the prolongated iterate
        const bool want_dot = (l - 1 == 0) && dot_partial;
        smooth(F, F.b, nu, false, want_dot ? dot_partial : nullptr, dot_nblk, false, level_prolong_fused(l - 1) ? &lev_[l - 2] : nullptr);
    }
}

// AMG_solver::AMG_solve_jacobi (src/AMG_phases.cpp:151-230) on device vectors.
int Engine::amg_solve_dev(const double *b, double *x, int iterations, double *hist, int hist_cap, int *ncycles)
{
    if (!ready_) return SPARSH_ESTATE;
    DevLevel &L0 = lev_[0];
    const int n = L0.n;
    HIPCHK(hipMemcpyAsync(L0.x, x, (size_t)n * 8, hipMemcpyDeviceToDevice, st_));
    if (place_tried > 0 && !dist_ && !L0.deep && !ks_.active) {  // (an open Krylov session owns work_[0])
        // the setup chose the buffers of iterate, twin and right-hand side together (tune_placement): run the cycles on the
        // engine's copy of b instead of on a caller buffer placed wherever it happens to be (one 8n-byte copy per solve)
        HIPCHK(hipMemcpyAsync(work_[0], b, (size_t)n * 8, hipMemcpyDeviceToDevice, st_));
        b = work_[0];
    }
    int cycles = 0;
    double r1 = op_resnorm(0, b, L0.x);
    int rc = SPARSH_OK;
    auto one_cycle = [&]() {
        vcycle(b, false, nullptr, nullptr);
        ++cycles;
        r1 = op_resnorm(0, b, lev_[0].x);
        if (hist && cycles - 1 < hist_cap) hist[cycles - 1] = r1;
    };
    if (iterations > 0) {
        while (cycles < iterations && fault_ == SPARSH_OK) one_cycle();
    } else if (iterations == -1) {
        while (r1 > prm_.tol && fault_ == SPARSH_OK) {
            if (cycles >= prm_.max_iter) {
                rc = SPARSH_ENOCONV;
                break;
            }
            one_cycle();
            if (prm_.print_solve) std::printf("%d %g\n", cycles, r1);
            if (!(r1 == r1)) {
                rc = SPARSH_ENUMERIC;
                break;
            }
        }
    }
    HIPCHK(hipMemcpyAsync(x, lev_[0].x, (size_t)n * 8, hipMemcpyDeviceToDevice, st_));
    HIPCHK(hipStreamSynchronize(st_));
    if (ncycles) *ncycles = cycles;
    return fault_ != SPARSH_OK ? fault_ : rc;
}

// Solver_CG_1 (precond = false) / Solver_PCG_1 (precond = true), split into the part before
// the while loop (pcg_init: r0, ||r0||, z0 = V(r0), p = z0; src/AMG_main_solvers.cpp:124-133)
// and the loop body (pcg_steps: :136-159) so a caller can time exactly k iterations.
int Engine::pcg_init(const double *b, double *x, bool precond)
{
    const int n = lev_[0].n;
    double *r = work_[0], *p = work_[1];
    int nb = 0;
    ks_ = KrylovState();
    ks_.precond = precond;
    ks_.b = b;
    ks_.x = x;
    const double *xin = x;
    if (dist_) {  // the caller's x has no room for the halo: stage it in an engine vector
        launch_copy(n, x, work_[3], st_);
        xin = work_[3];
    }
    op_residual(0, b, xin, r);  // r0 = b - A x
    launch_dot(n, r, r, part0_, &nb, st_);
    if (!precond) finalize(FIN_STORE, part0_, nullptr, nb, S_RR, nullptr, 0);
    finalize(FIN_SQRT, part0_, nullptr, nb, S_RES, nullptr, 0);
    ks_.r1 = read_scalar(S_RES);
    if (precond && f32_ready_) {
        vcycle_f32(r, work_[4], part0_, &nb);  // z0 = V32(r0)
        finalize(FIN_STORE, part0_, nullptr, nb, S_RZ, nullptr, 0);
        launch_copy(n, work_[4], p, st_);
    } else if (precond) {
        vcycle(r, true, part0_, &nb);  // z0 = V(r0), zero initial guess
        finalize(FIN_STORE, part0_, nullptr, nb, S_RZ, nullptr, 0);
        launch_copy(n, lev_[0].x, p, st_);
    } else {
        launch_copy(n, r, p, st_);
    }
    ks_.active = true;
    return fault_;
}

// One pass of the loop body of Solver_PCG_1 / Solver_CG_1 (src/AMG_main_solvers.cpp:138-152,
// :72-79) as stream work only; slot = residual-history index, or -1 to take it from the device
// counter (graph replay).
void Engine::pcg_body(bool precond, int slot)
{
    const int n = lev_[0].n;
    double *r = work_[0], *p = work_[1], *Ap = work_[2];
    double *x = ks_.x;
    int nb = 0;
    CsrArgs a;
    a.x = p;
    a.y = Ap;
    a.partial = part0_;
    const int np = apply_A(lev_[0], OP_SPMV_DOT, a);  // Ap = A p ; p.Ap
    finalize(precond ? FIN_PCG_ALPHA : FIN_CG_ALPHA, part0_, nullptr, np, 0, nullptr, 0);
    // x += alpha p ; r -= alpha Ap ; r.r -- with a preconditioner the r.r partials wait in part1_ and
    // are reduced together with z.r after the V-cycle: one finalize launch (one all-reduce) less
    // with the fp64 V-cycle behind it the update also writes the cycle's zero-guess sweep of level 0 (z0 = omega r / d)
    // (... unless the cycle's first launch on level 0 is the three-sweep one, which reads r alone)
    const bool fuse_zero = cfg_.fuse_cg_zero && precond && !f32_ready_ && lev_.size() > 1 && !lev_[0].deep && prm_.sweeps > 0 &&
                           !(prm_.sweeps >= 3 && zero_start(lev_[0]));
    // with the fp64 V-cycle behind it x += alpha p waits for the direction update at the end of this iteration (one read of p for both)
    const bool defer_x = cfg_.defer_x && precond && !f32_ready_;
    double *x_now = defer_x ? nullptr : x;
    if (fuse_zero)
        launch_cg_update_zero(n, scal_, p, Ap, x_now, r, part1_, &nb, diag_stream(lev_[0]), lev_[0].diag_const, prm_.omega, lev_[0].x, st_, cfg_.cg_nt);
    else
        launch_cg_update(n, scal_, p, Ap, x_now, r, precond ? part1_ : part0_, &nb, st_);
    if (precond && f32_ready_) {
        const int nb_rr = nb;
        vcycle_f32(r, work_[4], part0_, &nb);  // float hierarchy, fp64 in/out, fused z0.r0
        finalize(FIN_PCG_BETA_RES, part0_, part1_, nb, 0, hist_dev_, slot, nb_rr);
        launch_p_update(n, scal_, work_[4], p, st_);
    } else if (precond) {
        const int nb_rr = nb;
        vcycle(r, true, part0_, &nb, fuse_zero);  // z0 = 0 ; z0 = V(r0) ; fused z0.r0
        finalize(FIN_PCG_BETA_RES, part0_, part1_, nb, 0, hist_dev_, slot, nb_rr);
        if (defer_x) launch_xp_update(n, scal_, lev_[0].x, p, x, st_);  // x += alpha p ; p = z0 + beta p
        else launch_p_update(n, scal_, lev_[0].x, p, st_);                // p = z0 + beta p
    } else {
        finalize(FIN_CG_BETA, part0_, nullptr, nb, 0, hist_dev_, slot);
        launch_p_update(n, scal_, r, p, st_);
    }
}

void Engine::drop_graph()
{
    if (graph_.exec) HIPCHK(hipGraphExecDestroy(graph_.exec));
    if (graph_.graph) HIPCHK(hipGraphDestroy(graph_.graph));
    graph_ = GraphState();
}

// Capture one PCG iteration (≈230 kernel launches at 13 levels) into a hipGraph.  Every buffer
// the iteration touches is engine-owned and fixed; the Jacobi ping-pong makes a V-cycle from a
// zero guess start and end in fixed buffers, so the captured launches stay valid for every
// later iteration and solve on the same x vector.
bool Engine::capture_graph(bool precond)
{
    drop_graph();
    if (hipStreamBeginCapture(st_, hipStreamCaptureModeThreadLocal) != hipSuccess) return false;
    pcg_body(precond, -1);
    hipGraph_t g = nullptr;
    if (hipStreamEndCapture(st_, &g) != hipSuccess || !g) return false;
    hipGraphExec_t e = nullptr;
    if (hipGraphInstantiate(&e, g, nullptr, nullptr, 0) != hipSuccess) {
        HIPCHK(hipGraphDestroy(g));
        return false;
    }
    graph_.graph = g;
    graph_.exec = e;
    graph_.x = ks_.x;
    graph_.precond = precond;
    return true;
}

int Engine::pcg_steps(int nsteps, int *done)
{
    if (!ks_.active) {
        error = "krylov_init has not been called";
        return SPARSH_ESTATE;
    }
    const bool precond = ks_.precond;
    int rc = SPARSH_OK;
    int did = 0;
    const int check_every = std::max(1, prm_.check_every);
    // hipGraph replay: single GPU, no per-launch profiling events
    bool use_graph = prm_.use_graph && !dist_ && !prof.enabled && (int)lev_.size() > 1;
    if (use_graph && !(graph_.exec && graph_.x == ks_.x && graph_.precond == precond)) use_graph = capture_graph(precond);
    if (use_graph) HIPCHK(hipMemcpyAsync(iter_ctr_, &ks_.count, sizeof(int), hipMemcpyHostToDevice, st_));
    while (ks_.count < lev_[0].nglob && ks_.r1 > prm_.tol && did < nsteps && fault_ == SPARSH_OK) {
        const int count = ++ks_.count;
        ++did;
        const int slot = std::min(count - 1, hist_cap_dev_ - 1);
        if (use_graph)
            HIPCHK(hipGraphLaunch(graph_.exec, st_));
        else
            pcg_body(precond, slot);
        if (count % check_every == 0 || did >= nsteps) {
            ks_.r1 = read_hist(slot);
            if (prm_.print_solve) std::printf("%d\t%g\n", count, ks_.r1);
            if (!(ks_.r1 == ks_.r1)) {
                rc = SPARSH_ENUMERIC;
                break;
            }
        }
    }
    if (done) *done = did;
    return fault_ != SPARSH_OK ? fault_ : rc;
}

int Engine::krylov_hist(double *hist, int hist_cap)
{
    HIPCHK(hipStreamSynchronize(st_));
    const int m = std::min(std::min(ks_.count, hist_cap), hist_cap_dev_);
    if (hist && m > 0) HIPCHK(hipMemcpy(hist, hist_dev_, (size_t)m * 8, hipMemcpyDeviceToHost));
    return ks_.count;
}

int Engine::pcg(const double *b, double *x, int max_iters, double *hist, int hist_cap, int *iters, bool precond)
{
    int rc = pcg_init(b, x, precond);
    if (rc != SPARSH_OK) return rc;
    int did = 0;
    rc = pcg_steps(max_iters, &did);
    if (rc == SPARSH_OK && ks_.r1 > prm_.tol && ks_.count < lev_[0].nglob) rc = SPARSH_ENOCONV;
    if (fault_ != SPARSH_OK) rc = fault_;
    const int count = krylov_hist(hist, hist_cap);
    if (iters) *iters = count;
    ks_.active = false;
    return rc;
}

// Solver_BiCG_1 (precond = false) / Solver_PBiCG_1 (precond = true).
int Engine::bicg(const double *b, double *x, int max_iters, double *hist, int hist_cap, int *iters, bool precond)
{
    const int n = lev_[0].n;
    double *r0 = work_[0], *r = work_[1], *p = work_[2], *Ap = work_[3], *s = work_[4], *As = work_[5], *p1buf = work_[6];
    int nb = 0;
    const double *xin = x;
    if (dist_) {
        launch_copy(n, x, work_[7], st_);
        xin = work_[7];
    }
    op_residual(0, b, xin, r0);
    launch_copy(n, r0, r, st_);
    launch_copy(n, r0, p, st_);
    launch_dot(n, r0, r0, part0_, &nb, st_);
    finalize(FIN_SQRT, part0_, nullptr, nb, S_RES, nullptr, 0);
    double res = read_scalar(S_RES);
    int count = 0;
    int rc = SPARSH_OK;
    const int check_every = std::max(1, prm_.check_every);
    while (res > prm_.tol) {
        if (fault_ != SPARSH_OK) break;
        if (count >= max_iters) {
            rc = SPARSH_ENOCONV;
            break;
        }
        const double *p1 = p;
        if (precond && f32_ready_) {
            vcycle_f32(p, p1buf, part0_, &nb);  // float hierarchy (opt-in), fp64 in/out
            p1 = p1buf;
        } else if (precond) {
            vcycle(p, true, nullptr, nullptr);  // p1 = 0 ; p1 = V(p)
            launch_copy(n, lev_[0].x, p1buf, st_);
            p1 = p1buf;
        }
        op_spmv(0, p1, Ap);
        launch_dot2(n, r, r0, Ap, r0, part0_, part1_, &nb, st_);  // alpha1 = r.r0 ; Ap.r0
        finalize(FIN_BICG_ALPHA, part0_, part1_, nb, 0, nullptr, 0);
        launch_bicg_s(n, scal_, r, Ap, s, st_);
        const double *s1 = s;
        if (precond && f32_ready_) {
            vcycle_f32(s, work_[7], part0_, &nb);
            s1 = work_[7];
        } else if (precond) {
            vcycle(s, true, nullptr, nullptr);  // s1 = 0 ; s1 = V(s)
            s1 = lev_[0].x;
        }
        op_spmv(0, s1, As);
        launch_dot2(n, As, s, As, As, part0_, part1_, &nb, st_);
        finalize(FIN_BICG_OMEGA, part0_, part1_, nb, 0, nullptr, 0);
        launch_bicg_xr(n, scal_, p1, s1, s, As, r0, x, r, part0_, part1_, &nb, st_);
        const int slot = std::min(count, hist_cap_dev_ - 1);
        finalize(FIN_BICG_BETA, part0_, part1_, nb, 0, hist_dev_, slot);
        launch_bicg_p(n, scal_, r, Ap, p, st_);
        ++count;
        if (count % check_every == 0 || count >= max_iters) {
            res = read_hist(slot);
            if (prm_.print_solve) std::printf("%d\t%g\n", count - 1, res);
            if (!(res == res)) {
                rc = SPARSH_ENUMERIC;
                break;
            }
        }
    }
    HIPCHK(hipStreamSynchronize(st_));
    if (hist) {
        const int m = std::min(std::min(count, hist_cap), hist_cap_dev_);
        if (m > 0) HIPCHK(hipMemcpy(hist, hist_dev_, (size_t)m * 8, hipMemcpyDeviceToHost));
    }
    if (iters) *iters = count;
    return fault_ != SPARSH_OK ? fault_ : rc;
}

void Engine::profile_begin()
{
    if (prof.ev.empty()) {
        prof.ev.resize(kProfEvents);
        for (auto &e : prof.ev) HIPCHK(hipEventCreate(&e));
    }
    prof.used = 0;
    prof.run_launches.clear();
    prof.launches = 0;
    prof.seconds = 0;
}

void Engine::profile_collect()
{
    if (st_) HIPCHK(hipStreamSynchronize(st_));
    prof.launches = 0;
    prof.seconds = 0;
    for (size_t k = 0; k + 1 < prof.used; k += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, prof.ev[k], prof.ev[k + 1]) == hipSuccess) {
            prof.seconds += ms * 1e-3;
            prof.launches += k / 2 < prof.run_launches.size() ? prof.run_launches[k / 2] : 1;
        }
    }
}

int Engine::solve_dev(int method, const double *b, double *x, int max_iters, double *hist, int hist_cap, int *iters, double *seconds)
{
    if (!ready_) {
        error = "sparsh_setup has not been called";
        return SPARSH_ESTATE;
    }
    if (fault_ != SPARSH_OK) return fault_;  // sticky: a failed device/transport step leaves undefined state behind
    if (max_iters <= 0) max_iters = prm_.max_iter;
    if (prof.enabled) profile_begin();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (seconds) {
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, st_));
    }
    int rc;
    switch (method) {
    case SPARSH_AMG: {
        int it = (max_iters < prm_.max_iter) ? max_iters : -1;
        rc = amg_solve_dev(b, x, it, hist, hist_cap, iters);
    } break;
    case SPARSH_CG: rc = pcg(b, x, max_iters, hist, hist_cap, iters, false); break;
    case SPARSH_PCG: rc = pcg(b, x, max_iters, hist, hist_cap, iters, true); break;
    case SPARSH_BICG: rc = bicg(b, x, max_iters, hist, hist_cap, iters, false); break;
    case SPARSH_PBICG: rc = bicg(b, x, max_iters, hist, hist_cap, iters, true); break;
    default: error = "unknown method"; return SPARSH_EINVAL;
    }
    if (seconds) {
        HIPCHK(hipEventRecord(e1, st_));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        *seconds = ms * 1e-3;
        HIPCHK(hipEventDestroy(e0));
        HIPCHK(hipEventDestroy(e1));
    }
    if (prof.enabled) profile_collect();
    note_hip(hipGetLastError(), "kernel launch during the solve");
    if (fault_ != SPARSH_OK) return fault_;
    if (rc == SPARSH_ENOCONV) error = "iteration cap reached before ||r|| <= tol";
    if (rc == SPARSH_ENUMERIC) error = "NaN residual";
    return rc;
}

}  // namespace sparsh
