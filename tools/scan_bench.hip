// scan_bench.hip -- microbenchmark of the K x N scan kernels on synthetic nodes (development tool).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o /tmp/scan_bench tools/scan_bench.hip
// run:   /tmp/scan_bench [N] [K]
#include "../po_rrt_amd/csrc/porrt_device.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>

using namespace porrt;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// variant A: pure math ceiling -- no rare path at all (min of d2 only)
__global__ __launch_bounds__(256) void v_minonly(const double *nxp, const double *nyp, const double *sx, const double *sy, double *out,
                                                  uint32_t N, uint32_t NC) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    uint32_t j0, j1;
    chunk_range(N, NC, c, j0, j1);
    const double qx = sx[k], qy = sy[k];
    cdouble_p px = as_const(nxp) + j0, py = as_const(nyp) + j0;
    double m = 1e300;
    for (uint32_t j = j0; j + 8 <= j1; j += 8, px += 8, py += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { double d2 = dist2(px[u], py[u], qx, qy); m = d2 < m ? d2 : m; }
    }
    out[(size_t)c * gridDim.x * 256 + k] = m;
}

// variant B: nodes staged through LDS (ds_read broadcast) instead of scalar loads
__global__ __launch_bounds__(256) void v_lds(const double *nxp, const double *nyp, const double *sx, const double *sy, double *out,
                                              uint32_t N, uint32_t NC) {
    __shared__ double lx[512], ly[512];
    const uint32_t k = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    uint32_t j0, j1;
    chunk_range(N, NC, c, j0, j1);
    const double qx = sx[k], qy = sy[k];
    double m = 1e300;
    for (uint32_t base = j0; base < j1; base += 512) {
        const uint32_t n = min(512u, j1 - base);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < n; t += 256) { lx[t] = nxp[base + t]; ly[t] = nyp[base + t]; }
        __syncthreads();
        uint32_t t = 0;
        for (; t + 8 <= n; t += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { double d2 = dist2(lx[t + u], ly[t + u], qx, qy); m = d2 < m ? d2 : m; }
        }
    }
    out[(size_t)c * gridDim.x * 256 + k] = m;
}

// variant C: f32 filter keys (2 FMA + min) -- ceiling of a filtered design
__global__ __launch_bounds__(256) void v_f32(const float *nxf, const float *nyf, const float *nn2, const double *sx, const double *sy,
                                              float *out, uint32_t N, uint32_t NC) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    uint32_t j0, j1;
    chunk_range(N, NC, c, j0, j1);
    const float ax = -2.0f * (float)sx[k], ay = -2.0f * (float)sy[k];
    typedef const __attribute__((address_space(4))) float *cf;
    cf px = (cf)(uintptr_t)nxf + j0, py = (cf)(uintptr_t)nyf + j0, pn = (cf)(uintptr_t)nn2 + j0;
    float m = 1e30f;
    for (uint32_t j = j0; j + 8 <= j1; j += 8, px += 8, py += 8, pn += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { float key = __builtin_fmaf(px[u], ax, __builtin_fmaf(py[u], ay, pn[u])); m = key < m ? key : m; }
    }
    out[(size_t)c * gridDim.x * 256 + k] = m;
}

int main(int argc, char **argv) {
    uint32_t N = argc > 1 ? atoi(argv[1]) : 100000, K = argc > 2 ? atoi(argv[2]) : 1024;
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> U(-1, 1);
    std::vector<double> hx(N + 64), hy(N + 64), qx(K), qy(K);
    std::vector<float> fx(N + 64), fy(N + 64), fn(N + 64);
    for (uint32_t i = 0; i < N + 64; ++i) { hx[i] = U(rng); hy[i] = U(rng); fx[i] = hx[i]; fy[i] = hy[i]; fn[i] = hx[i] * hx[i] + hy[i] * hy[i]; }
    for (auto &v : qx) v = U(rng);
    for (auto &v : qy) v = U(rng);
    RunConst rc;
    memset(&rc, 0, sizeof rc);
    uint32_t nat[2] = {N, N};
    float *dfx, *dfy, *dfn, *dof;
    CK(hipMalloc(&rc.nx, (N + 64) * 8)); CK(hipMalloc(&rc.ny, (N + 64) * 8)); CK(hipMalloc(&rc.sx, K * 8)); CK(hipMalloc(&rc.sy, K * 8));
    CK(hipMalloc(&rc.n_at, 8)); CK(hipMalloc(&rc.part_D, (size_t)K * kMaxChunks * 8)); CK(hipMalloc(&rc.part_id, (size_t)K * kMaxChunks * 4));
    CK(hipMalloc(&rc.q_x, K * 8)); CK(hipMalloc(&rc.q_y, K * 8)); CK(hipMalloc(&rc.q_vid, K * 4)); CK(hipMalloc(&rc.cand_cnt, K * 4));
    CK(hipMalloc(&rc.cand_id, (size_t)K * 4096 * 4)); CK(hipMalloc(&rc.cnt, sizeof(Counters))); CK(hipMalloc((void **)&rc.rad_T2, (N + 8) * 8));
    CK(hipMalloc(&dfx, (N + 64) * 4)); CK(hipMalloc(&dfy, (N + 64) * 4)); CK(hipMalloc(&dfn, (N + 64) * 4)); CK(hipMalloc(&dof, (size_t)K * kMaxChunks * 4));
    rc.cand_cap = 4096; rc.part_stride = K;
    CK(hipMemcpy(rc.nx, hx.data(), (N + 64) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(rc.ny, hy.data(), (N + 64) * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(rc.sx, qx.data(), K * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(rc.sy, qy.data(), K * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(rc.q_x, qx.data(), K * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(rc.q_y, qy.data(), K * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfx, fx.data(), (N + 64) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dfy, fy.data(), (N + 64) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dfn, fn.data(), (N + 64) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rc.n_at, nat, 8, hipMemcpyHostToDevice));
    CK(hipMemset(rc.q_vid, 0, K * 4));
    CK(hipMemset(rc.cnt, 0, sizeof(Counters)));
    // radius threshold: expect ~36 neighbours per sample
    std::vector<double> t2(N + 8, 36.0 * 4.0 / (3.14159265 * N));
    CK(hipMemcpy((void *)rc.rad_T2, t2.data(), (N + 8) * 8, hipMemcpyHostToDevice));
    RunConst *drc;
    CK(hipMalloc(&drc, sizeof rc));
    CK(hipMemcpy(drc, &rc, sizeof rc, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, uint32_t NC, auto launch) {
        for (int w = 0; w < 3; ++w) launch();
        CK(hipDeviceSynchronize());
        const int reps = 20;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double us = ms * 1e3 / reps, pairs = (double)N * K;
        printf("%-22s N=%u NC=%3u  %8.2f us  %7.2f Gpair/s  (6 flop/pair: %6.2f TF)\n", name, N, NC, us, pairs / us * 1e-3, 6 * pairs / us * 1e-6);
    };
    for (uint32_t NC : {32u, 64u, 128u, 256u}) {
        dim3 grid(K / 256, NC);
        timeit("nn_scan", NC, [&] { hipLaunchKernelGGL(k_nn_scan<false>, grid, dim3(256), 0, 0, (const RunConst *)drc, 0u, 0u, K, NC); });
        timeit("radius_scan", NC, [&] { CK(hipMemsetAsync(rc.cand_cnt, 0, K * 4)); hipLaunchKernelGGL(k_radius_scan, grid, dim3(256), 0, 0, (const RunConst *)drc, 0u, K, NC); });
        timeit("minonly(sload)", NC, [&] { hipLaunchKernelGGL(v_minonly, grid, dim3(256), 0, 0, (const double *)rc.nx, (const double *)rc.ny, (const double *)rc.sx, (const double *)rc.sy, rc.part_D, N, NC); });
        timeit("minonly(lds)", NC, [&] { hipLaunchKernelGGL(v_lds, grid, dim3(256), 0, 0, (const double *)rc.nx, (const double *)rc.ny, (const double *)rc.sx, (const double *)rc.sy, rc.part_D, N, NC); });
        timeit("f32key(sload)", NC, [&] { hipLaunchKernelGGL(v_f32, grid, dim3(256), 0, 0, (const float *)dfx, (const float *)dfy, (const float *)dfn, (const double *)rc.sx, (const double *)rc.sy, dof, N, NC); });
    }
    return 0;
}
