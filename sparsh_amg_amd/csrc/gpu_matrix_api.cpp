// gpu_matrix_api.cpp -- sp_matrix_gpu and the phase utilities of the drop-in C++ API
// (reference: src/AMG_gpu_matrix.cu:26-142, src/AMG_gpu_phase_utilities.cu:129-167), on this
// library's own kernels.  A stand-alone device operator outside any hierarchy: what user code of
// the reference builds when it writes its own device-side loops (src/AMG_main_solvers.cu:283-300).
#include <cmath>
#include <iostream>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "kernels.hpp"
#include "../../include/AMG_gpu_phase_utilities.hpp"
// include/AMG.hpp is deliberately NOT included here: its short macros (omega, th, ...) would
// rewrite member names.  The relaxation factor is the reference's macro value (include/AMG.hpp:16).
static constexpr double kOmegaJacobi = 0.66667;

using namespace sparsh;

namespace {

struct GpuImpl {
    DevCsr D;
    double *partial = nullptr, *scal = nullptr, *pinned = nullptr;
};

template <class T>
T *dmalloc(size_t count)
{
    T *p = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&p), (count ? count : 1) * sizeof(T)) != hipSuccess) {
        std::cout << "Error in GPU memory allocation" << std::endl;
        return nullptr;
    }
    return p;
}

}  // namespace

sp_matrix_gpu::sp_matrix_gpu(sp_matrix_mg &A)
{
    nrow = A.nrow;
    ncol = A.ncol;
    nnz = A.rowptr ? A.rowptr[A.nrow] : A.nnz;
    rowptr = dmalloc<int>((size_t)nrow + 1);
    colindex = dmalloc<int>((size_t)nnz + kCsrPad);
    val = dmalloc<double>((size_t)nnz + kCsrPad);
    if (nrow == ncol) diag = dmalloc<double>((size_t)nrow);
    GpuImpl *g = new GpuImpl();
    impl_ = g;
    if (colindex) (void)hipMemset(colindex + nnz, 0, kCsrPad * sizeof(int));
    if (val) (void)hipMemset(val + nnz, 0, kCsrPad * sizeof(double));
}

void sp_matrix_gpu::matrix_transfer_gpu(sp_matrix_mg &A, hipStream_t streams)
{
    GpuImpl *g = static_cast<GpuImpl *>(impl_);
    (void)hipMemcpyAsync(rowptr, A.rowptr, sizeof(int) * ((size_t)nrow + 1), hipMemcpyHostToDevice, streams);
    (void)hipMemcpyAsync(colindex, A.colindex, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice, streams);
    (void)hipMemcpyAsync(val, A.val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, streams);
    if (diag && A.diagonal) (void)hipMemcpyAsync(diag, A.diagonal, sizeof(double) * (size_t)nrow, hipMemcpyHostToDevice, streams);
    // row-block schedule of the CSR-stream kernels (host-built from the caller's rowptr)
    std::vector<int> rb((size_t)nrow + 2);
    DevCsr &D = g->D;
    D.nrow = nrow;
    D.ncol = ncol;
    D.nnz = nnz;
    D.rowptr = rowptr;
    D.col = colindex;
    D.val = val;
    const std::vector<int> rec = rowblock_records(nrow, A.rowptr, &D.nblk);
    if (D.rowblk) (void)hipFree(D.rowblk);
    D.rowblk = dmalloc<int>(rec.size());
    (void)hipMemcpy(D.rowblk, rec.data(), sizeof(int) * rec.size(), hipMemcpyHostToDevice);
    if (!g->partial) g->partial = dmalloc<double>((size_t)D.nblk + 8);
    if (!g->scal) g->scal = dmalloc<double>(S_COUNT);
    if (!g->pinned) (void)hipHostMalloc(reinterpret_cast<void **>(&g->pinned), 8 * sizeof(double), hipHostMallocDefault);
}

void sp_matrix_gpu::smooth_jacobi(double *bgpu, double *xgpu, double *hgpu, hipStream_t streams, int steps)
{
    GpuImpl *g = static_cast<GpuImpl *>(impl_);
    if (!g || !g->D.rowblk || !diag) {
        std::cout << "sparsh: smooth_jacobi needs matrix_transfer_gpu of a square matrix first" << std::endl;
        return;
    }
    KernelConfig kc;
    kc.kind = 0;  // stand-alone operator: CSR-stream kernels (no mirror layouts built)
    double *cur = xgpu, *nxt = hgpu;
    for (int k = 0; k < steps; ++k) {
        CsrArgs a;
        a.x = cur;
        a.b = bgpu;
        a.d = diag;
        a.y = nxt;
        a.omega = kOmegaJacobi;
        launch_csr(g->D, OP_JACOBI, a, false, streams, kc);
        std::swap(cur, nxt);
    }
    if (cur != xgpu) launch_copy(nrow, cur, xgpu, streams);
}

sp_matrix_gpu::~sp_matrix_gpu()
{
    GpuImpl *g = static_cast<GpuImpl *>(impl_);
    if (g) {
        if (g->D.rowblk) (void)hipFree(g->D.rowblk);
        if (g->partial) (void)hipFree(g->partial);
        if (g->scal) (void)hipFree(g->scal);
        if (g->pinned) (void)hipHostFree(g->pinned);
        delete g;
        impl_ = nullptr;
    }
    if (rowptr) (void)hipFree(rowptr);
    if (colindex) (void)hipFree(colindex);
    if (val) (void)hipFree(val);
    if (diag) (void)hipFree(diag);
    rowptr = colindex = nullptr;
    val = diag = nullptr;
}

void pin_AMG_MG_matrix(sp_matrix_mg &, double *&, double *&, sp_matrix_mg &) {}
void unpin_AMG_MG_matrix(sp_matrix_mg &, double *&, double *&, sp_matrix_mg &) {}

void gpu_swap_pointers(sp_matrix_gpu *&A1, sp_matrix_gpu *&A2)
{
    sp_matrix_gpu *t = A1;
    A1 = A2;
    A2 = t;
}

double residual(sp_matrix_gpu &A, double *b, double *x, double *h, hipStream_t streams)
{
    (void)h;  // scratch in the reference's three-pass formulation; the fused kernel needs none
    GpuImpl *g = static_cast<GpuImpl *>(A.impl_);
    if (!g || !g->D.rowblk) {
        std::cout << "sparsh: residual needs matrix_transfer_gpu first" << std::endl;
        return NAN;
    }
    KernelConfig kc;
    kc.kind = 0;
    CsrArgs a;
    a.x = x;
    a.b = b;
    a.partial = g->partial;
    const int np = launch_csr(g->D, OP_RESNORM, a, false, streams, kc);
    launch_finalize(FIN_SQRT, g->partial, nullptr, np, g->scal, S_RES, nullptr, 0, streams);
    (void)hipMemcpyAsync(g->pinned, g->scal + S_RES, sizeof(double), hipMemcpyDeviceToHost, streams);
    (void)hipStreamSynchronize(streams);
    return g->pinned[0];
}
