// Developer probe: which HIP streams share a hardware queue?  Two long single-workgroup kernels on two streams take one kernel's
// time when the streams sit on different queues and two when they share one.  hipcc --offload-arch=gfx950 -O2 -o queue_probe queue_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(unsigned long long cycles, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 1234567u) *sink = 1;
}
static double pair_ms(hipStream_t a, hipStream_t b, unsigned long long cyc) {
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, a, cyc, (unsigned *)nullptr);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, cyc, (unsigned *)nullptr);
    (void)hipStreamSynchronize(a);
    (void)hipStreamSynchronize(b);
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 10, pre = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<hipStream_t> junk(pre), s(n);
    for (auto &j : junk) { (void)hipStreamCreateWithFlags(&j, hipStreamNonBlocking); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, j, 10ull, (unsigned *)nullptr); }
    (void)hipDeviceSynchronize();
    for (auto &x : s) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    const unsigned long long cyc = 200000ull;      // 100 MHz wall clock: 2 ms
    pair_ms(s[0], s[1], cyc);
    printf("streams %d after %d used ones; pair time in ms (2 = side by side, 4 = one after the other)\n", n, pre);
    for (int i = 0; i < n; ++i) {
        printf("%2d:", i);
        for (int j = 0; j < n; ++j) printf(" %4.1f", i == j ? 0.0 : pair_ms(s[i], s[j], cyc));
        printf("\n");
    }
    return 0;
}
