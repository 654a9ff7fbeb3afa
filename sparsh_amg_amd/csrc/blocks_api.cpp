// blocks_api.cpp -- the operator-level building blocks of the reference's CPU path under their own
// names (include/AMG_smoothers.hpp, AMG_cycle_utilities.hpp, AMG_coarsening.hpp,
// AMG_coarse_level_solver.hpp): host vectors in/out, device execution.  Each sp_matrix_mg gets a
// device mirror on first use, cached behind its opaque A1 member (where the reference keeps the
// MKL handle) and released by ~sp_matrix_mg.
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "coarse.hpp"
#include "host_setup.hpp"
#include "kernels.hpp"
#include "../../include/AMG_coarse_level_solver.hpp"
#include "../../include/AMG_coarsening.hpp"
#include "../../include/AMG_cycle_utilities.hpp"
#include "../../include/AMG_smoothers.hpp"

using namespace sparsh;

namespace {

constexpr double kOmegaJacobi = 0.66667;  // include/AMG.hpp:16 of the reference

struct DevBuf {
    double *p = nullptr;
    explicit DevBuf(size_t n, const double *src = nullptr)
    {
        if (hipMalloc(reinterpret_cast<void **>(&p), (n ? n : 1) * sizeof(double)) != hipSuccess) {
            std::cout << "sparsh: device allocation failed" << std::endl;
            std::exit(1);
        }
        if (src) (void)hipMemcpy(p, src, n * sizeof(double), hipMemcpyHostToDevice);
    }
    ~DevBuf() { (void)hipFree(p); }
    void get(double *dst, size_t n) const
    {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(dst, p, n * sizeof(double), hipMemcpyDeviceToHost);
    }
};

// device mirror of one host operator: CSR + explicit transpose (for P^T r) + row-block schedules
struct Mirror {
    const sp_matrix_mg *owner = nullptr;
    int nnz = 0;
    DevCsr A, T;
    double *diag = nullptr, *partial = nullptr, *scal = nullptr;
    std::vector<void *> allocs;
    ~Mirror()
    {
        for (void *q : allocs) (void)hipFree(q);
    }
    template <class V>
    V *up(const V *src, size_t count, size_t pad = 0)
    {
        V *d = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&d), (count + pad + 1) * sizeof(V)) != hipSuccess) {
            std::cout << "sparsh: device allocation failed" << std::endl;
            std::exit(1);
        }
        allocs.push_back(d);
        if (pad) (void)hipMemset(d + count, 0, pad * sizeof(V));
        if (count) (void)hipMemcpy(d, src, count * sizeof(V), hipMemcpyHostToDevice);
        return d;
    }
    void upload(const HostCsr &H, DevCsr &D)
    {
        D.nrow = H.nrow;
        D.ncol = H.ncol;
        D.nnz = H.nnz();
        D.rowptr = up(H.rowptr, (size_t)H.nrow + 1);
        D.col = up(H.col, (size_t)D.nnz, kCsrPad);
        D.val = up(H.val, (size_t)D.nnz, kCsrPad);
        const std::vector<int> rec = rowblock_records(H.nrow, H.rowptr, &D.nblk);
        D.rowblk = up(rec.data(), rec.size());
    }
};

struct Registry {  // A1 holds a Mirror*; tagged so that foreign values of A1 are never dereferenced
    std::vector<Mirror *> live;
    bool has(const void *p) const
    {
        for (const Mirror *m : live)
            if (m == p) return true;
        return false;
    }
};
Registry &registry()
{
    static Registry *r = new Registry();  // never destroyed: ~sp_matrix_mg of static objects may run late
    return *r;
}

Mirror &mirror_of(sp_matrix_mg &A, bool need_transpose)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        std::cout << "sparsh: no HIP device visible: the MI355X HIP path is the only compute path" << std::endl;
        std::exit(1);
    }
    const int nnz = A.rowptr[A.nrow];
    Mirror *m = registry().has(A.A1) ? static_cast<Mirror *>(A.A1) : nullptr;
    if (m && (m->owner != &A || m->nnz != nnz || m->A.nrow != A.nrow)) m = nullptr;  // stale
    if (!m) {
        m = new Mirror();
        m->owner = &A;
        m->nnz = nnz;
        HostCsr H = HostCsr::alias(A.nrow, A.ncol, A.rowptr, A.colindex, A.val);
        m->upload(H, m->A);
        if (A.diagonal && A.nrow == A.ncol) m->diag = m->up(A.diagonal, (size_t)A.nrow);
        std::vector<double> z((size_t)std::max(m->A.nblk, 1) + 64, 0.0);
        m->partial = m->up(z.data(), z.size());
        m->scal = m->up(z.data(), (size_t)S_COUNT);
        registry().live.push_back(m);
        A.A1 = m;
    }
    if (need_transpose && !m->T.rowptr) {
        HostCsr H = HostCsr::alias(A.nrow, A.ncol, A.rowptr, A.colindex, A.val);
        HostCsr T = transpose(H);  // entries in ascending row order = the sequential scatter order
        m->upload(T, m->T);
    }
    return *m;
}

// the stand-alone mirrors carry no sliced layouts: run the CSR-stream kernels
KernelConfig csr_kind0()
{
    KernelConfig c;
    c.kind = 0;
    return c;
}
const KernelConfig kCsr0 = csr_kind0();

void jacobi_device(sp_matrix_mg &A, double *b, double *x, int iteration)
{
    if (!A.diagonal) {
        std::cout << "sparsh: jacobi_smoother needs sp_matrix_fill_diagonal() first" << std::endl;
        return;
    }
    Mirror &m = mirror_of(A, false);
    const size_t n = (size_t)A.nrow;
    DevBuf db(n, b), dx(n, x), dt(n);
    double *cur = dx.p, *nxt = dt.p;
    int count = 0;
    while (count++ <= iteration) {  // iteration + 1 sweeps, src/AMG_smoothers.cpp:59-60
        CsrArgs a;
        a.x = cur;
        a.b = db.p;
        a.d = m.diag;
        a.y = nxt;
        a.omega = kOmegaJacobi;
        launch_csr(m.A, OP_JACOBI, a, false, nullptr, kCsr0);
        std::swap(cur, nxt);
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(x, cur, n * sizeof(double), hipMemcpyDeviceToHost);
}

double residual_device(sp_matrix_mg &A, double *b, double *x)
{
    Mirror &m = mirror_of(A, false);
    const size_t n = (size_t)A.nrow;
    DevBuf db(n, b), dx((size_t)A.ncol, x);
    CsrArgs a;
    a.x = dx.p;
    a.b = db.p;
    a.partial = m.partial;
    const int np = launch_csr(m.A, OP_RESNORM, a, false, nullptr, kCsr0);
    launch_finalize(FIN_SQRT, m.partial, nullptr, np, m.scal, S_RES, nullptr, 0, nullptr);
    double r = 0.0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&r, m.scal + S_RES, sizeof(double), hipMemcpyDeviceToHost);
    return r;
}

void store_residual_device(sp_matrix_mg &A, double *b, double *x, double *r)
{
    Mirror &m = mirror_of(A, false);
    const size_t n = (size_t)A.nrow;
    DevBuf db(n, b), dx((size_t)A.ncol, x), dr(n);
    CsrArgs a;
    a.x = dx.p;
    a.b = db.p;
    a.y = dr.p;
    launch_csr(m.A, OP_RESID, a, false, nullptr, kCsr0);
    dr.get(r, n);
}

void transfer_residual_device(sp_matrix_mg &P, double *r, double *b)
{
    Mirror &m = mirror_of(P, true);
    DevBuf dr((size_t)P.nrow, r), dbc((size_t)P.ncol);
    CsrArgs a;
    a.x = dr.p;
    a.y = dbc.p;
    launch_csr(m.T, OP_SPMV, a, false, nullptr, kCsr0);
    dbc.get(b, (size_t)P.ncol);
}

void transfer_solution_device(sp_matrix_mg &P, double *x, double *x1)
{
    Mirror &m = mirror_of(P, false);
    DevBuf dc((size_t)P.ncol, x), df((size_t)P.nrow, x1);
    CsrArgs a;
    a.x = dc.p;
    a.y = df.p;
    launch_csr(m.A, OP_ADD, a, false, nullptr, kCsr0);
    df.get(x1, (size_t)P.nrow);
}

sp_matrix_mg *to_mg(const HostCsr &H)
{
    sp_matrix_mg *M = new sp_matrix_mg(H.nrow, H.ncol, H.nnz());
    std::copy(H.rowptr, H.rowptr + H.nrow + 1, M->rowptr);
    std::copy(H.col, H.col + H.nnz(), M->colindex);
    std::copy(H.val, H.val + H.nnz(), M->val);
    M->sp_matrix_fill();
    return M;
}

void coarsen_host(sp_matrix_mg &A, sp_matrix_mg *&Ac, sp_matrix_mg &P1)
{
    HostCsr Ha = HostCsr::alias(A.nrow, A.ncol, A.rowptr, A.colindex, A.val);
    HostCsr Hp = HostCsr::alias(P1.nrow, P1.ncol, P1.rowptr, P1.colindex, P1.val);
    HostCsr R = transpose(Hp);
    bool agg = (Hp.nnz() == Hp.nrow);
    for (int i = 0; agg && i < Hp.nrow; ++i) agg = (Hp.rowptr[i + 1] - Hp.rowptr[i] == 1) && Hp.val[Hp.rowptr[i]] == 1.0;
    HostCsr C = galerkin(Ha, Hp, R, agg);
    Ac = to_mg(C);
    Ac->sp_matrix_fill_diagonal();
}

void not_in_build(const char *what)
{
    std::cerr << "sparsh: " << what << " is not part of the MI355X build" << std::endl;
}

}  // namespace

// called from ~sp_matrix_mg (amg_api.cpp)
void sparsh_release_mirror(void *a1)
{
    Registry &r = registry();
    for (size_t k = 0; k < r.live.size(); ++k)
        if (r.live[k] == a1) {
            delete r.live[k];
            r.live.erase(r.live.begin() + (long)k);
            return;
        }
}

namespace sequential
{
void jacobi_smoother(sp_matrix_mg &A, double *&b, double *&x, int iteration) { jacobi_device(A, b, x, iteration); }
void sor_smoother(sp_matrix_mg &, double *&, double *&, int) { not_in_build("sor_smoother"); }
double residual(sp_matrix_mg &A, double *&b, double *&x) { return residual_device(A, b, x); }
void transfer_residual(sp_matrix_mg &P1, double *&r, double *&b) { transfer_residual_device(P1, r, b); }
void transfer_solution(sp_matrix_mg &P1, double *&x, double *&x1) { transfer_solution_device(P1, x, x1); }
void store_residual(sp_matrix_mg &A, double *&b, double *&x, double *&r) { store_residual_device(A, b, x, r); }
void coarsen_matrix(sp_matrix_mg &A, sp_matrix_mg *&Ac, sp_matrix_mg &P1) { coarsen_host(A, Ac, P1); }

void HEM_Prolongator(sp_matrix_mg &A, sp_matrix_mg *&P, int l1)
{
    P = to_mg(hem_prolongator(HostCsr::alias(A.nrow, A.ncol, A.rowptr, A.colindex, A.val), l1));
}
void beck_prolongator(sp_matrix_mg &A, sp_matrix_mg *&P1)
{
    P1 = to_mg(sparsh::beck_prolongator(HostCsr::alias(A.nrow, A.ncol, A.rowptr, A.colindex, A.val)));
}
void mis_prolongator(sp_matrix_mg &, sp_matrix_mg *&P1)
{
    not_in_build("mis_prolongator (never enabled in the reference; non-deterministic)");
    P1 = nullptr;
}
void C_W_prolongator(sp_matrix_mg &, sp_matrix_mg *&P, int)
{
    not_in_build("C_W_prolongator (never enabled in the reference)");
    P = nullptr;
}
void HEM_Prolongator_2(sp_matrix_mg &, sp_matrix_mg *&P)
{
    not_in_build("HEM_Prolongator_2 (never enabled in the reference)");
    P = nullptr;
}
}  // namespace sequential

namespace parallel
{
void jacobi_smoother(sp_matrix_mg &A, double *&b, double *&x, int iteration) { jacobi_device(A, b, x, iteration); }
void sor_smoother(sp_matrix_mg &, double *&, double *&, int) { not_in_build("sor_smoother"); }
double residual(sp_matrix_mg &A, double *&b, double *&x) { return residual_device(A, b, x); }
void transfer_residual(sp_matrix_mg &P1, double *&r, double *&b) { transfer_residual_device(P1, r, b); }
void transfer_solution(sp_matrix_mg &P1, double *&x, double *&x1) { transfer_solution_device(P1, x, x1); }
void store_residual(sp_matrix_mg &A, double *&b, double *&x, double *&r) { store_residual_device(A, b, x, r); }
void coarsen_matrix(sp_matrix_mg &A, sp_matrix_mg *&Ac, sp_matrix_mg &P1) { coarsen_host(A, Ac, P1); }
void reorder_rhs(sp_matrix_mg &, double *&) { not_in_build("reorder_rhs (multicolour SOR)"); }
void reorder_prolongator(sp_matrix_mg &, sp_matrix_mg *&) { not_in_build("reorder_prolongator (multicolour SOR)"); }
}  // namespace parallel

// ------------------------------------------------------------------ coarse direct solver

namespace {
struct DirectImpl {
    CoarseSolver solver;
    int n = 0;
};
}  // namespace

Direct_Solver_Pardiso::Direct_Solver_Pardiso(sp_matrix_mg &A)
{
    n = A.nrow;
    const HostCsr H = HostCsr::alias(A.nrow, A.ncol, A.rowptr, A.colindex, A.val);
    DirectImpl *d = new DirectImpl();
    d->n = n;
    std::string err;
    bool ok;
    if (n <= 8192) {  // explicit inverse, one GEMV per solve
        std::vector<double> inv;
        if (!sparse_inverse(H, inv)) {
            std::cout << "\nERROR during symbolic factorization: singular matrix" << std::endl;
            error = 1;
            std::exit(1);  // the reference exits on a PARDISO error (src/AMG_coarse_level_solver.cpp:53-57)
        }
        ok = d->solver.setup_dense(n, inv.data(), err);
    } else {  // nested-dissection multifrontal factors, built on the device (what PARDISO does on the host in the reference)
        ok = d->solver.setup_nd(H, nullptr, err);
    }
    if (!ok) {
        std::cout << "sparsh: coarse direct solver failed (" << err << "); without a HIP device there is no CPU fallback" << std::endl;
        error = 1;
        std::exit(1);
    }
    impl_ = d;
}

void Direct_Solver_Pardiso::Direct_Solver_Pardiso_solve(double *&b, double *&x)
{
    DirectImpl *d = static_cast<DirectImpl *>(impl_);
    if (!d) return;
    DevBuf db((size_t)n, b), dx((size_t)n);
    d->solver.solve(db.p, dx.p, nullptr);
    dx.get(x, (size_t)n);
}

Direct_Solver_Pardiso::~Direct_Solver_Pardiso()
{
    DirectImpl *d = static_cast<DirectImpl *>(impl_);
    if (d) {
        delete d;
        impl_ = nullptr;
    }
}
