// scan_bench.hip -- microbenchmark of the K x N scan kernels on synthetic nodes (development tool).
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -o /tmp/scan_bench tools/scan_bench.hip
// run:   /tmp/scan_bench [N] [K]
#include "../po_rrt_amd/csrc/porrt_device.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <cmath>

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

// variant D: f32 keys, nodes staged once per block into LDS with coalesced vector loads, ds_read_b128 broadcast
__global__ __launch_bounds__(256) void v_f32_lds(const float *nxf, const float *nyf, const float *nn2, const double *sx, const double *sy,
                                                  float *out, uint32_t N, uint32_t NC) {
    __shared__ __attribute__((aligned(16))) float lx[1024], ly[1024], l2[1024];
    const uint32_t k = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    uint32_t j0, j1;
    chunk_range(N, NC, c, j0, j1);
    const uint32_t n = j1 - j0;
    for (uint32_t t = threadIdx.x; t < n; t += 256) { lx[t] = nxf[j0 + t]; ly[t] = nyf[j0 + t]; l2[t] = nn2[j0 + t]; }
    __syncthreads();
    const float ax = -2.0f * (float)sx[k], ay = -2.0f * (float)sy[k];
    float m = 1e30f;
    for (uint32_t t = 0; t + 8 <= n; t += 8) {
        const float4 x0 = *(const float4 *)&lx[t], x1 = *(const float4 *)&lx[t + 4];
        const float4 y0 = *(const float4 *)&ly[t], y1 = *(const float4 *)&ly[t + 4];
        const float4 z0 = *(const float4 *)&l2[t], z1 = *(const float4 *)&l2[t + 4];
        float k0 = __builtin_fmaf(x0.x, ax, __builtin_fmaf(y0.x, ay, z0.x)), k1 = __builtin_fmaf(x0.y, ax, __builtin_fmaf(y0.y, ay, z0.y));
        float k2 = __builtin_fmaf(x0.z, ax, __builtin_fmaf(y0.z, ay, z0.z)), k3 = __builtin_fmaf(x0.w, ax, __builtin_fmaf(y0.w, ay, z0.w));
        float k4 = __builtin_fmaf(x1.x, ax, __builtin_fmaf(y1.x, ay, z1.x)), k5 = __builtin_fmaf(x1.y, ax, __builtin_fmaf(y1.y, ay, z1.y));
        float k6 = __builtin_fmaf(x1.z, ax, __builtin_fmaf(y1.z, ay, z1.z)), k7 = __builtin_fmaf(x1.w, ax, __builtin_fmaf(y1.w, ay, z1.w));
        m = fminf(m, fminf(fminf(fminf(k0, k1), fminf(k2, k3)), fminf(fminf(k4, k5), fminf(k6, k7))));
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
    CK(hipMalloc(&rc.n_at, 8)); CK(hipMalloc(&rc.part_D, (size_t)K * 512 * 8)); CK(hipMalloc(&rc.part_id, (size_t)K * 512 * 4));
    CK(hipMalloc(&rc.q_x, K * 8)); CK(hipMalloc(&rc.q_y, K * 8)); CK(hipMalloc(&rc.q_vid, K * 4)); CK(hipMalloc(&rc.cand_cnt, K * 4));
    CK(hipMalloc(&rc.cand_id, (size_t)K * 4096 * 4)); CK(hipMalloc(&rc.cnt, sizeof(Counters))); CK(hipMalloc((void **)&rc.rad_T2, (N + 8) * 8));
    CK(hipMalloc(&dfx, (N + 64) * 4)); CK(hipMalloc(&dfy, (N + 64) * 4)); CK(hipMalloc(&dfn, (N + 64) * 4)); CK(hipMalloc(&dof, (size_t)K * 512 * 4));
    rc.cand_cap = 4096; rc.part_stride = K;
    CK(hipMalloc(&rc.part_mask, (size_t)(K / 64) * 512 * 8));
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
    // filter operands: threshold from a bound 4x the true nearest squared distance (what the pyramid gives)
    rc.fx = dfx; rc.fy = dfy; rc.f2 = dfn;
    rc.filt_E = 32.0 * ldexp(1.0, -24);
    std::vector<float> hax(2 * K), hay(2 * K), hthr(2 * K);
    for (uint32_t k = 0; k < K; ++k) {
        double m = 1e300;
        for (uint32_t j = 0; j < N; ++j) { double dx = hx[j] - qx[k], dy = hy[j] - qy[k]; double d = dx * dx + dy * dy; m = d < m ? d : m; }
        double t = 4.0 * m - (qx[k] * qx[k] + qy[k] * qy[k]) + rc.filt_E;
        hax[k] = hax[K + k] = (float)(-2.0 * qx[k]); hay[k] = hay[K + k] = (float)(-2.0 * qy[k]);
        hthr[k] = hthr[K + k] = nextafterf((float)t, 1e30f);
    }
    CK(hipMalloc(&rc.q_ax, 2 * K * 4)); CK(hipMalloc(&rc.q_ay, 2 * K * 4)); CK(hipMalloc(&rc.q_thr, 2 * K * 4));
    CK(hipMemcpy(rc.q_ax, hax.data(), 2 * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rc.q_ay, hay.data(), 2 * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rc.q_thr, hthr.data(), 2 * K * 4, hipMemcpyHostToDevice));
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
    // radius scan with NO neighbours at all (threshold 0): cost of staging + filter loop alone
    {
        std::vector<double> z(N + 8, 0.0);
        double *dz;
        CK(hipMalloc(&dz, (N + 8) * 8));
        CK(hipMemcpy(dz, z.data(), (N + 8) * 8, hipMemcpyHostToDevice));
        RunConst rc0 = rc;
        rc0.rad_T2 = dz;
        RunConst *drc0;
        CK(hipMalloc(&drc0, sizeof rc0));
        CK(hipMemcpy(drc0, &rc0, sizeof rc0, hipMemcpyHostToDevice));
        dim3 sgrid(K / 1024, 256);
        timeit("radius_scan(no hits)", 256, [&] { hipLaunchKernelGGL(k_radius_scan, sgrid, dim3(1024), 0, 0, (const RunConst *)drc0, 0u, K, 256u); });
    }
    for (uint32_t NC : {256u}) {
        dim3 grid(K / 256, NC);
        dim3 sgrid(K / 1024, NC);
        timeit("nn_scan", NC, [&] { hipLaunchKernelGGL(k_nn_scan<false>, sgrid, dim3(1024), 0, 0, (const RunConst *)drc, 0u, 0u, K, NC); });
        timeit("radius_scan", NC, [&] { CK(hipMemsetAsync(rc.cand_cnt, 0, K * 4)); hipLaunchKernelGGL(k_radius_scan, sgrid, dim3(1024), 0, 0, (const RunConst *)drc, 0u, K, NC); });
        timeit("minonly(sload)", NC, [&] { hipLaunchKernelGGL(v_minonly, grid, dim3(256), 0, 0, (const double *)rc.nx, (const double *)rc.ny, (const double *)rc.sx, (const double *)rc.sy, rc.part_D, N, NC); });
        timeit("minonly(lds)", NC, [&] { hipLaunchKernelGGL(v_lds, grid, dim3(256), 0, 0, (const double *)rc.nx, (const double *)rc.ny, (const double *)rc.sx, (const double *)rc.sy, rc.part_D, N, NC); });
        timeit("f32key(lds)", NC, [&] { hipLaunchKernelGGL(v_f32_lds, grid, dim3(256), 0, 0, (const float *)dfx, (const float *)dfy, (const float *)dfn, (const double *)rc.sx, (const double *)rc.sy, dof, N, NC); });
        timeit("f32key(sload)", NC, [&] { hipLaunchKernelGGL(v_f32, grid, dim3(256), 0, 0, (const float *)dfx, (const float *)dfy, (const float *)dfn, (const double *)rc.sx, (const double *)rc.sy, dof, N, NC); });
    }
    return 0;
}
