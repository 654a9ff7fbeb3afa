// Does rocprofv3's counter-collection mode survive tens of thousands of dispatches from one process?  (the --pmc
// pass segfaults inside hipLaunchKernel when the block-tridiagonal coarse factorisation -- ~32 k launches of bt_gj_kernel
// with dynamic LDS -- runs in the profiled process.)  This program touches none of the solver's code: a trivial kernel,
// optionally with dynamic LDS like bt_gj_kernel, launched N times on one stream.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/many_launches.hip -o tools/micro/many_launches
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- tools/micro/many_launches 40000 8192
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ void tick(double *x, int k)
{
    extern __shared__ double s[];
    if (threadIdx.x == 0) s[0] = (double)k;
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) x[0] += s[0];
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 40000;
    const size_t lds = argc > 2 ? (size_t)std::atoi(argv[2]) : 0;
    double *x = nullptr;
    if (hipMalloc(&x, 8) != hipSuccess) return 1;
    (void)hipMemset(x, 0, 8);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    for (int k = 0; k < n; ++k) {
        hipLaunchKernelGGL(tick, dim3(128), dim3(256), lds, st, x, k);
        if ((k + 1) % 5000 == 0) {
            std::printf("%d launches enqueued\n", k + 1);
            std::fflush(stdout);
        }
    }
    const hipError_t e = hipStreamSynchronize(st);
    double h = 0.0;
    (void)hipMemcpy(&h, x, 8, hipMemcpyDeviceToHost);
    std::printf("done: %d launches, sync %s, sum %.0f (expected %.0f)\n", n, hipGetErrorString(e), h, 0.5 * n * (n - 1.0));
    return e == hipSuccess ? 0 : 2;
}
