// valu_peak.hip -- register-only VALU issue-rate microbenchmark (development tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1); } } while (0)
constexpr int R = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double b, double c) {
    double a[8];
    float f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = threadIdx.x * 1e-3 + u; f[u] = (float)a[u]; }
    float fb = (float)b, fc = (float)c;
    for (int i = 0; i < R; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { a[u] = a[u] * b; a[u] = a[u] + c; }           // v_mul_f64 + v_add_f64 (2 instr)
            if (MODE == 1) { a[u] = __builtin_fma(a[u], b, c); }            // v_fma_f64
            if (MODE == 2) { f[u] = __builtin_fmaf(f[u], fb, fc); }         // v_fma_f32
            if (MODE == 3) { a[u] = a[u] < b ? a[u] + c : a[u] * c; }       // cmp + cndmask mix
            if (MODE == 4) { a[u] = fmin(a[u] * b, c + a[u]); }             // mul, add, min
        }
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += a[u] + f[u];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double *out;
    CK(hipMalloc(&out, 8192 * 256 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"mul_f64+add_f64", "fma_f64", "fma_f32", "cmp/select f64", "mul,add,min f64"};
    const double ipi[] = {2, 1, 1, 3, 3};
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        for (int mode = 0; mode < 5; ++mode) {
            auto launch = [&] {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001, 1e-9);
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double instr = 5.0 * blocks * 256.0 * R * 8 * ipi[mode];   // lane-instructions
            double waves_per_simd = blocks * 4 / 1024.0;
            printf("blocks %4d (%.1f waves/SIMD) %-18s %7.2f T lane-instr/s  -> %.2f cycles/wave-instr/SIMD @2.4GHz\n", blocks, waves_per_simd, names[mode],
                   instr / (ms * 1e-3) * 1e-12, 1024.0 * 64 * 2.4e9 / (instr / (ms * 1e-3)));
        }
    }
    return 0;
}
