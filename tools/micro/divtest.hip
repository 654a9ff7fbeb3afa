// Is the division by a constant with its reciprocal hoisted out of the loop (div_const in csrc/kernels.hip) bitwise the compiler's own fp64
// division?  Random bit patterns (all exponents, denormals, infinities, NaNs) and ordinary magnitudes as numerators, ten divisors incl. extreme
// ones: compared against the device's a / b and against the host's.  Also counts the lanes that leave the fast path (v_div_scale would scale
// the divisor differently for that numerator).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/divtest.hip -o tools/micro/divtest && tools/micro/divtest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cstdint>
struct DivConst { double b, bs0, r0; };
__device__ inline DivConst make_div_const(double b)
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
__device__ inline double div_const(double a, const DivConst &c)
{
    bool fd, fn;
    const double bs = __builtin_amdgcn_div_scale(a, c.b, false, &fd);
    const double as = __builtin_amdgcn_div_scale(a, c.b, true, &fn);
    if (__builtin_amdgcn_ballot_w64(bs != c.bs0) != 0ull) return a / c.b;
    const double q0 = as * c.r0;
    const double rem = __builtin_fma(-bs, q0, as);
    const double q = __builtin_amdgcn_div_fmas(rem, c.r0, q0, fn);
    return __builtin_amdgcn_div_fixup(q, c.b, a);
}
__global__ void k(const double *a, int n, double b, double *o1, double *o2, unsigned long long *slow)
{
    const DivConst c = make_div_const(b);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o1[i] = a[i] / b;
    bool fd; 
    const double bs = __builtin_amdgcn_div_scale(a[i], b, false, &fd);
    if (bs != c.bs0) atomicAdd(slow, 1ull);
    o2[i] = div_const(a[i], c);
}
int main()
{
    const int n = 1 << 24;
    std::vector<double> h(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint64_t bits = s;
        if (i % 3 == 0) {  // moderate exponents
            bits = (bits & 0x800FFFFFFFFFFFFFull) | ((uint64_t)(1023 - 40 + (s >> 52) % 80) << 52);
        }
        std::memcpy(&h[i], &bits, 8);
    }
    h[0] = 0.0; h[1] = -0.0; h[2] = 1e-320; h[3] = 1e308; h[4] = -1e308; h[5] = 5e-324;
    double *a, *o1, *o2; unsigned long long *slow;
    hipMalloc(&a, n * 8); hipMalloc(&o1, n * 8); hipMalloc(&o2, n * 8); hipMalloc(&slow, 8);
    hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice);
    const double bs[] = {6.0, 10.0, 16.0, 28.0, 3.0, 7.123456789, 1e-300, 1e300, 4.9e-324, 44.0};
    std::vector<double> r1(n), r2(n);
    for (double b : bs) {
        hipMemset(slow, 0, 8);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, n, b, o1, o2, slow);
        hipDeviceSynchronize();
        hipMemcpy(r1.data(), o1, n * 8, hipMemcpyDeviceToHost);
        hipMemcpy(r2.data(), o2, n * 8, hipMemcpyDeviceToHost);
        unsigned long long sl; hipMemcpy(&sl, slow, 8, hipMemcpyDeviceToHost);
        long bad = 0, badcpu = 0;
        for (int i = 0; i < n; ++i) {
            if (std::memcmp(&r1[i], &r2[i], 8)) { if (bad < 3) printf("  a=%a: %a vs %a\n", h[i], r1[i], r2[i]); ++bad; }
            double c = h[i] / b;
            if (std::memcmp(&r1[i], &c, 8) && !(c != c && r1[i] != r1[i])) ++badcpu;
        }
        printf("b=%g: %ld of %d differ (fast vs '/'), %ld differ between device '/' and host '/', %llu lanes off the fast path\n", b, bad, n, badcpu, sl);
    }
    return 0;
}
