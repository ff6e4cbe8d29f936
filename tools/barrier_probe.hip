// developer probe: cost of a grid barrier with agent-scope release / acquire on gfx950 (hipcc --offload-arch=gfx950 -O3 -o /tmp/bp tools/barrier_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k_bar(uint32_t *bar, uint32_t *data, uint32_t iters, uint32_t work) {
    uint32_t phase = 0;
    const uint32_t nblk = gridDim.x;
    __shared__ uint32_t s_go;
    for (uint32_t it = 0; it < iters; ++it) {
        // some work: every block writes its slot, reads its neighbour's after the barrier
        for (uint32_t w = 0; w < work; ++w) data[((blockIdx.x + it) % nblk) * 256 + threadIdx.x] += it + w;
        __syncthreads();
        if (threadIdx.x == 0) {
            ++phase;
            if (MODE == 0) { __threadfence(); __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
            else __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * nblk) __builtin_amdgcn_s_sleep(1);
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_go = 1;
        }
        __syncthreads();
    }
}
int main(int argc, char **argv) {
    uint32_t *bar, *data;
    const int nblk = argc > 1 ? atoi(argv[1]) : 512;
    hipMalloc(&bar, 4); hipMalloc(&data, 4 * 256 * 4096);
    hipMemset(data, 0, 4 * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (uint32_t work : {0u, 4u}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(bar, 0, 4);
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k_bar<0>, dim3(nblk), dim3(256), 0, 0, bar, data, 1000u, work);
                else hipLaunchKernelGGL(k_bar<1>, dim3(nblk), dim3(256), 0, 0, bar, data, 1000u, work);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            printf("blocks %d mode %s work %u: %.2f us per barrier\n", nblk, mode == 0 ? "release/acquire(agent)" : "relaxed only", work, best);
        }
    return 0;
}
