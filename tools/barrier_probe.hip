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
            // a kernel that can wait must not be able to hang the GPU: the wait gives up after two seconds (100 MHz wall clock) and the
            // flag makes every later barrier of every block fall through
            const unsigned long long t0 = wall_clock64();
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase * nblk && !__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > 200000000ull) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_go = 1;
        }
        __syncthreads();
    }
}
int main(int argc, char **argv) {
    uint32_t *bar, *data;
    int nblk = argc > 1 ? atoi(argv[1]) : 512;
    // every block of a spin barrier must be resident: the grid is cut to what the device holds at once (with a margin of one block per CU:
    // the occupancy API reads one high near some register counts), and the launch is a cooperative one, which checks it again
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int per_cu0 = 0, per_cu1 = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu0, k_bar<0>, 256, 0);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu1, k_bar<1>, 256, 0);
    const int per_cu = (per_cu0 < per_cu1 ? per_cu0 : per_cu1) - 1;
    const int fit = (per_cu > 0 ? per_cu : 1) * prop.multiProcessorCount;
    if (nblk > fit) { printf("blocks %d do not fit the device at once: cut to %d\n", nblk, fit); nblk = fit; }
    hipMalloc(&bar, 8); hipMalloc(&data, 4 * 256 * 4096);
    hipMemset(data, 0, 4 * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (uint32_t work : {0u, 4u}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(bar, 0, 8);
                hipEventRecord(e0);
                uint32_t iters = 1000u, wk = work;
                void *args[] = {&bar, &data, &iters, &wk};
                const hipError_t le = hipLaunchCooperativeKernel(mode == 0 ? (const void *)k_bar<0> : (const void *)k_bar<1>, dim3(nblk), dim3(256), args, 0, 0);
                if (le != hipSuccess) { printf("cooperative launch refused: %s\n", hipGetErrorString(le)); return 1; }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
            }
            uint32_t gave_up = 0; hipMemcpy(&gave_up, bar + 1, 4, hipMemcpyDeviceToHost);
            if (gave_up) { printf("a barrier gave up after two seconds: not every block was resident\n"); return 1; }
            printf("blocks %d mode %s work %u: %.2f us per barrier\n", nblk, mode == 0 ? "release/acquire(agent)" : "relaxed only", work, best);
        }
    return 0;
}
