// Feasibility probe for a temporally blocked double Jacobi sweep on a lexicographically ordered box grid (7-point stencil,
// constant coefficients): y = J(J(x)) in ONE pass over x, b and y.  A workgroup owns TY grid lines of every plane of a z-chunk
// and marches through the planes; the first sweep's plane lives in LDS (+ one ring of lines), its neighbours in z in registers,
// so x and b are read once (plus halo) and only the second sweep's result is stored: ~28-30 B per row for two sweeps instead
// of 48.  Every row is computed with the expression and the order of additions of the single-sweep kernels, so the result has
// to be bitwise equal to two single sweeps -- checked here.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/box2_proto.hip -o tools/micro/box2_proto
//   tools/micro/box2_proto [nx ny nz]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

struct Box {
    int nx, ny, nz;
    double c[7];  // -plane, -line, -1, 0, +1, +line, +plane
    double omega;
};

__global__ __launch_bounds__(256) void sweep1_kernel(Box g, const double *__restrict__ x, const double *__restrict__ b, double *__restrict__ y)
{
    const long n = (long)g.nx * g.ny * g.nz;
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int P = g.nx * g.ny;
    const int k = (int)(r / P), rem = (int)(r % P), j = rem / g.nx, i = rem % g.nx;
    double sum = 0.0;
    if (k > 0) sum = sum + g.c[0] * x[r - P];
    if (j > 0) sum = sum + g.c[1] * x[r - g.nx];
    if (i > 0) sum = sum + g.c[2] * x[r - 1];
    const double xi = x[r];
    sum = sum + g.c[3] * xi;
    if (i < g.nx - 1) sum = sum + g.c[4] * x[r + 1];
    if (j < g.ny - 1) sum = sum + g.c[5] * x[r + g.nx];
    if (k < g.nz - 1) sum = sum + g.c[6] * x[r + P];
    const double h = 1.0 * b[r] + (-1.0) * sum;
    y[r] = xi + g.omega * h / g.c[3];
}

constexpr int kT = 1024;  // threads per workgroup

// a / b for a divisor known before the loop: the compiler's own fp64 division sequence (v_div_scale x2, v_rcp + two Newton steps on the
// scaled divisor, q0 = a_s r, rem = fma(-b_s, q0, a_s), v_div_fmas, v_div_fixup) with the part that depends on b alone -- the refined
// reciprocal of the scaled divisor -- computed once.  Where v_div_scale would scale b differently for this a (denormals, exponents
// ~2^1000 apart: never for residual-sized numbers) the whole wave takes the plain division.  Same instructions on the same operands:
// bitwise the plain division's result.
struct DivConst {
    double b, bs0, r0;
};
__device__ __forceinline__ DivConst make_div_const(double b)
{
    DivConst c;
    c.b = b;
    bool f;
    c.bs0 = __builtin_amdgcn_div_scale(1.0, b, false, &f);
    double r = __builtin_amdgcn_rcp(c.bs0);
    double e = __builtin_fma(-c.bs0, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-c.bs0, r, 1.0);
    r = __builtin_fma(r, e, r);
    c.r0 = r;
    return c;
}
__device__ __forceinline__ double div_const(double a, const DivConst &c)
{
#ifdef FASTDIV
    bool fd, fn;
    const double bs = __builtin_amdgcn_div_scale(a, c.b, false, &fd);
    const double as = __builtin_amdgcn_div_scale(a, c.b, true, &fn);
    if (__builtin_amdgcn_ballot_w64(bs != c.bs0) != 0ull) return a / c.b;
    const double q0 = as * c.r0;
    const double rem = __builtin_fma(-bs, q0, as);
    const double q = __builtin_amdgcn_div_fmas(rem, c.r0, q0, fn);
    return __builtin_amdgcn_div_fixup(q, c.b, a);
#else
    return a / c.b;
#endif
}

// region of a workgroup in a plane: lines j0-2 .. j0+TY+1 (TY+4 lines, contiguous in memory), local index p = tid + kT*q
// x0 is needed on all of them, the first sweep runs on lines 1 .. TY+2 of the region, the second on lines 2 .. TY+1.
// Neighbours that do not exist are read as +0.0 (zero pad cell between the lines in LDS, zero lines / planes outside the grid):
// c * 0.0 = +-0.0 and sum + (+-0.0) == sum bit for bit (sum is never -0.0: it starts at +0.0 and x + y = -0.0 only for
// x = y = -0.0), so "skip the missing entry" needs no predicate.
template <int Q, bool XCD_REMAP>
__global__ __launch_bounds__(kT) void box2_kernel(Box g, int TY, int CZ, int ytiles, const double *__restrict__ x, const double *__restrict__ b,
                                                  double *__restrict__ y)
{
    extern __shared__ double lds[];
    const int nx = g.nx, ny = g.ny, nz = g.nz, P = nx * ny;
    const int R0 = (TY + 4) * nx;
    const int pitch = nx + 1;
    const int cells = (TY + 4) * pitch + 1;
    double *X0 = lds;          // x0, plane k, whole region
    double *X1 = lds + cells;  // x1, plane k-1
    // workgroups go round-robin to the 8 XCDs: give every XCD a contiguous run of (z-chunk, y-tile) pairs, so that the tiles
    // sharing halo lines sit behind the same L2
    int wg = blockIdx.x;
    const int nwg = gridDim.x;
    if (XCD_REMAP) {
        const int per = nwg / 8, rem = nwg % 8;  // XCD c owns per + (c < rem) consecutive ids
        const int c = wg % 8, r = wg / 8;
        wg = c * per + min(c, rem) + r;
    }
    const int tile = wg % ytiles, zc = wg / ytiles;
    const int j0 = tile * TY;
    const int z0 = zc * CZ, z1 = min(z0 + CZ, nz);
    const long base = (long)(j0 - 2) * nx;  // + k*P + p
    const int tid = threadIdx.x;
    const double c0 = g.c[0], c1 = g.c[1], c2 = g.c[2], c3 = g.c[3], c4 = g.c[4], c5 = g.c[5], c6 = g.c[6], om = g.omega;
    const DivConst dc = make_div_const(c3);
    for (int i = tid; i < 2 * cells; i += kT) lds[i] = 0.0;
    bool v0[Q], v1[Q], v2[Q];
    int sidx[Q];
    double xm[Q], xc[Q], xp[Q], bk[Q], bp[Q], x1m[Q], x1c[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int p = tid + kT * q;
        const int lr = p / nx;
        const int jr = j0 - 2 + lr;
        sidx[q] = p + lr + 1;
        v0[q] = p < R0 && jr >= 0 && jr < ny;
        v1[q] = v0[q] && lr >= 1 && lr <= TY + 2;
        v2[q] = v0[q] && lr >= 2 && lr <= TY + 1;
        xm[q] = xc[q] = xp[q] = bk[q] = bp[q] = x1m[q] = x1c[q] = 0.0;
    }
    const int ks = z0 - 1;  // first plane of the first sweep (may be -1: does not exist)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int p = tid + kT * q;
        if (v0[q]) {
            if (ks >= 1) xm[q] = x[(long)(ks - 1) * P + base + p];
            if (ks >= 0) xc[q] = x[(long)ks * P + base + p];
            if (ks + 1 < nz) xp[q] = x[(long)(ks + 1) * P + base + p];
        }
        if (v1[q] && ks >= 0) bk[q] = b[(long)ks * P + base + p];
    }
    __syncthreads();
    for (int k = ks; k <= z1; ++k) {
        const bool plane = k >= 0 && k < nz;  // uniform
        double xn[Q], bn[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int p = tid + kT * q;
            xn[q] = 0.0;
            bn[q] = 0.0;
            if (k + 1 <= z1) {
                if (v0[q] && k + 2 < nz) xn[q] = x[(long)(k + 2) * P + base + p];
                if (v1[q] && k + 1 < nz) bn[q] = b[(long)(k + 1) * P + base + p];
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
            if (v0[q]) X0[sidx[q]] = xc[q];
        __syncthreads();
        double x1k[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            x1k[q] = 0.0;
            if (plane && v1[q]) {
                const int s = sidx[q];
                double sum = 0.0;
                sum = sum + c0 * xm[q];
                sum = sum + c1 * X0[s - pitch];
                sum = sum + c2 * X0[s - 1];
                sum = sum + c3 * xc[q];
                sum = sum + c4 * X0[s + 1];
                sum = sum + c5 * X0[s + pitch];
                sum = sum + c6 * xp[q];
                const double h = 1.0 * bk[q] + (-1.0) * sum;
                x1k[q] = xc[q] + div_const(om * h, dc);
            }
        }
        const int k2 = k - 1;
        if (k2 >= z0 && k2 < z1) {  // uniform
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int p = tid + kT * q;
                if (v2[q]) {
                    const int s = sidx[q];
                    double sum = 0.0;
                    sum = sum + c0 * x1m[q];
                    sum = sum + c1 * X1[s - pitch];
                    sum = sum + c2 * X1[s - 1];
                    sum = sum + c3 * x1c[q];
                    sum = sum + c4 * X1[s + 1];
                    sum = sum + c5 * X1[s + pitch];
                    sum = sum + c6 * x1k[q];
                    const double h = 1.0 * bp[q] + (-1.0) * sum;
                    y[(long)k2 * P + base + p] = x1c[q] + div_const(om * h, dc);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (v1[q]) X1[sidx[q]] = x1k[q];
            x1m[q] = x1c[q];
            x1c[q] = x1k[q];
            xm[q] = xc[q];
            xc[q] = xp[q];
            xp[q] = xn[q];
            bp[q] = bk[q];
            bk[q] = bn[q];
        }
    }
}

static bool g_remap = true;
template <int Q>
void launch_box2(Box g, int TY, int CZ, int ytiles, int zch, size_t ldsb, const double *x, const double *b, double *y)
{
    if (g_remap)
        hipLaunchKernelGGL((box2_kernel<Q, true>), dim3(ytiles * zch), dim3(kT), ldsb, 0, g, TY, CZ, ytiles, x, b, y);
    else
        hipLaunchKernelGGL((box2_kernel<Q, false>), dim3(ytiles * zch), dim3(kT), ldsb, 0, g, TY, CZ, ytiles, x, b, y);
}

int main(int argc, char **argv)
{
    Box g;
    g.nx = argc > 3 ? std::atoi(argv[1]) : 216;
    g.ny = argc > 3 ? std::atoi(argv[2]) : 216;
    g.nz = argc > 3 ? std::atoi(argv[3]) : 216;
    const double c[7] = {-1.0, -1.0, -1.0, 6.0, -1.0, -1.0, -1.0};
    std::memcpy(g.c, c, sizeof c);
    g.omega = 0.66667;
    const long n = (long)g.nx * g.ny * g.nz;
    const int Q = argc > 5 ? std::atoi(argv[5]) : 4;
    int TY = kT * Q / g.nx - 4;
    if (TY > g.ny) TY = g.ny;
    if (TY < 2) {
        std::printf("lines too long for the region\n");
        return 1;
    }
    const int ytiles = (g.ny + TY - 1) / TY;
    int zch = (256 + ytiles - 1) / ytiles;
    int CZ = (g.nz + zch - 1) / zch;
    if (argc > 4) CZ = std::atoi(argv[4]);
    zch = (g.nz + CZ - 1) / CZ;
    const size_t ldsb = (size_t)2 * ((TY + 4) * (g.nx + 1) + 1) * 8;
    std::printf("grid %d x %d x %d = %ld rows; Q %d, TY %d, %d y-tiles, CZ %d, %d z-chunks, %d workgroups, LDS %zu B\n", g.nx, g.ny, g.nz, n, Q, TY,
                ytiles, CZ, zch, ytiles * zch, ldsb);
    std::vector<double> hx(n), hb(n);
    unsigned long long s = 88172645463325252ull;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        hx[i] = (double)(s % 2000001) / 1e6 - 1.0;
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        hb[i] = (double)(s % 2000001) / 1e6 - 1.0;
    }
    double *x, *b, *t, *y1, *y2;
    CHECK(hipMalloc(&x, n * 8));
    CHECK(hipMalloc(&b, n * 8));
    CHECK(hipMalloc(&t, n * 8));
    CHECK(hipMalloc(&y1, n * 8));
    CHECK(hipMalloc(&y2, n * 8));
    CHECK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b, hb.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemset(y2, 0, n * 8));
    g_remap = !(argc > 6 && std::atoi(argv[6]) == 0);
    std::printf("XCD remap %d\n", (int)g_remap);
    auto fused = [&](const double *in, double *out) {
        if (Q == 4) launch_box2<4>(g, TY, CZ, ytiles, zch, ldsb, in, b, out);
        else if (Q == 3) launch_box2<3>(g, TY, CZ, ytiles, zch, ldsb, in, b, out);
        else launch_box2<2>(g, TY, CZ, ytiles, zch, ldsb, in, b, out);
    };
    const int g1 = (int)((n + 255) / 256);
    hipLaunchKernelGGL(sweep1_kernel, dim3(g1), dim3(256), 0, 0, g, x, b, t);
    hipLaunchKernelGGL(sweep1_kernel, dim3(g1), dim3(256), 0, 0, g, t, b, y1);
    fused(x, y2);
    CHECK(hipDeviceSynchronize());
    std::vector<double> h1(n), h2(n);
    CHECK(hipMemcpy(h1.data(), y1, n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h2.data(), y2, n * 8, hipMemcpyDeviceToHost));
    long bad = 0;
    for (long i = 0; i < n; ++i)
        if (std::memcmp(&h1[i], &h2[i], 8) != 0) {
            if (bad < 5) std::printf("  row %ld: %.17g vs %.17g\n", i, h1[i], h2[i]);
            ++bad;
        }
    std::printf("bitwise check against two single sweeps: %ld rows differ\n", bad);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 20;
    float ms = 0.f;
    for (int w = 0; w < 2; ++w) {
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(sweep1_kernel, dim3(g1), dim3(256), 0, 0, g, x, b, t);
            hipLaunchKernelGGL(sweep1_kernel, dim3(g1), dim3(256), 0, 0, g, t, b, x);
        }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::printf("two single sweeps : %8.2f us per pair\n", ms * 1e3 / reps);
    for (int w = 0; w < 2; ++w) {
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) {
            fused(x, t);
            fused(t, x);
        }
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::printf("fused double sweep: %8.2f us per launch (= per pair)\n", ms * 1e3 / (2 * reps));
    return bad == 0 ? 0 : 3;
}
