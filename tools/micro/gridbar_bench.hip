// Micro-benchmark: cost of a device-wide barrier inside one persistent kernel on MI355X (8 XCDs), against
// the cost of a kernel boundary (back-to-back dependent launches).  hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned nwg, unsigned &epoch, int *fail)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope in HIP
        const unsigned target = (++epoch) * nwg;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1l << 24)) {  // bounded: never hang the GPU
                *fail = 1;
                ok = false;
                break;
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    return ok;
}

// each step: every WG writes its slot of `buf`, barrier, reads its neighbour's slot (checks visibility)
__global__ __launch_bounds__(256) void persistent(unsigned *ctr, int nsteps, double *buf, int *fail, int *bad)
{
    unsigned epoch = 0;
    const unsigned nwg = gridDim.x;
    for (int s = 0; s < nsteps; ++s) {
        if (threadIdx.x == 0) buf[(s & 1) * nwg + blockIdx.x] = (double)(s * 1000 + blockIdx.x);
        if (!grid_barrier(ctr, nwg, epoch, fail)) return;
        if (threadIdx.x == 0) {
            const unsigned nb = (blockIdx.x + 97) % nwg;
            const double v = __builtin_nontemporal_load(&buf[(s & 1) * nwg + nb]);
            if (v != (double)(s * 1000 + nb)) atomicAdd(bad, 1);
        }
    }
}

__global__ __launch_bounds__(256) void tiny(double *buf, int s) { if (threadIdx.x == 0) buf[blockIdx.x] = s; }

int main(int argc, char **argv)
{
    const int nsteps = argc > 1 ? atoi(argv[1]) : 2000;
    for (int nwg : {32, 64, 128, 256, 512}) {
        unsigned *ctr;
        int *fail, *bad;
        double *buf;
        hipMalloc(&ctr, 4); hipMalloc(&fail, 4); hipMalloc(&bad, 4); hipMalloc(&buf, 2 * 1024 * 8);
        hipMemset(ctr, 0, 4); hipMemset(fail, 0, 4); hipMemset(bad, 0, 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(persistent, dim3(nwg), dim3(256), 0, 0, ctr, 10, buf, fail, bad);
        hipMemset(ctr, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(persistent, dim3(nwg), dim3(256), 0, 0, ctr, nsteps, buf, fail, bad);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        int hf = 0, hb = 0;
        hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        hipEventRecord(e0, 0);
        for (int s = 0; s < nsteps; ++s) hipLaunchKernelGGL(tiny, dim3(nwg), dim3(256), 0, 0, buf, s);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms2 = 0;
        hipEventElapsedTime(&ms2, e0, e1);
        printf("nwg %4d: grid barrier %.3f us/step (timeout %d, stale reads %d); kernel boundary %.3f us/launch\n", nwg, ms * 1e3 / nsteps, hf, hb,
               ms2 * 1e3 / nsteps);
        hipFree(ctr); hipFree(fail); hipFree(bad); hipFree(buf);
    }
    return 0;
}
