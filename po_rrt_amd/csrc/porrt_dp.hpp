// porrt_dp.hpp -- expected costs over the belief graph on the device: conditional_dijkstra
// (src/belief_graph.rs:89-175) as called by PTO::compute_expected_costs_to_goals (src/pto.rs:261-275).
//
// The reference relaxes with a priority queue: popping v, every parent u gets
//     Action node       alternative = cost(u, v) + dist[v]
//     Observation node  alternative = sum over u's children vv of p(u -> vv) * (cost(u, vv) + dist[vv])
// and keeps it when it is smaller.  Every operation in there is monotone in f64 (a + b, p * x with p > 0, min), so the
// loop computes the greatest fixpoint below the start (finals 0, everything else +inf) of
//     dist[u] = min over children (cost + dist[child])          for action nodes
//     dist[u] = the sum above over all children                 for observation nodes
// and ANY fair order of relaxations ends in the same bits (tested on the CPU: the queue-driven restatement against plain
// sweeps, tests/test_oracle_dp.py).  The device uses the order that suits it: sweeps.  A belief node is re-evaluated
// (pull: from its children list) in the sweep after one of its children improved; improving marks the parents.  Two
// byte-flag arrays alternate; a sweep over a quiet graph costs one pass over the flags.
//
// Two layouts share the kernels: the context's belief graph (node i = graph node i / B with belief i % B, coordinates
// from the PTO graph) and explicit graphs (per-node coordinates and belief rows), which is how the reference's own
// known-answer graphs (belief_graph.rs:278-567) run through the same code.
#pragma once
#include "porrt_belief.hpp"

namespace porrt {

struct DpConst {
    unsigned long long n;             // belief nodes
    uint32_t B, nw;                   // implicit layout: beliefs per graph node; worlds
    const double *nx, *ny;            // implicit: graph node coordinates; explicit: per belief node
    const uint32_t *bvec;             // explicit: row of `beliefs` per node (nullptr = implicit: i % B)
    const double *beliefs;            // [rows][nw]
    const uint8_t *types;
    const unsigned long long *child_off, *par_off;
    const uint32_t *child_id, *par_id;
    double *dist;
    uint32_t *flags;                  // [0] a failed assert / panic of the reference, [1 + s] updates made by sweep s of a group
};

enum : uint32_t { DP_ERR_UNKNOWN_TYPE = 1, DP_ERR_ZERO_PROBABILITY = 2 };

__global__ __launch_bounds__(256) void k_dp_fill(double *__restrict__ dist, unsigned long long n, double v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) as_global(dist)[i] = v;
}

// finals: dist 0, parents to be evaluated by the first sweep
__global__ __launch_bounds__(256) void k_dp_set_finals(DpConst g, const unsigned long long *__restrict__ finals, unsigned long long n_final,
                                                       uint8_t *__restrict__ dirty) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_final) return;
    const unsigned long long i = as_global(finals)[k];
    as_global(g.dist)[i] = 0.0;
    for (unsigned long long e = as_global(g.par_off)[i]; e < as_global(g.par_off)[i + 1]; ++e) as_global(dirty)[as_global(g.par_id)[e]] = 1;
}

template <bool IMPLICIT>
__device__ __forceinline__ void dp_state(const DpConst &g, unsigned long long i, double &x, double &y, uint32_t &row) {
    if (IMPLICIT) {
        const uint32_t node = (uint32_t)(i / g.B);
        x = as_global(g.nx)[node]; y = as_global(g.ny)[node];
        row = (uint32_t)(i % g.B);
    } else {
        x = as_global(g.nx)[i]; y = as_global(g.ny)[i];
        row = as_global(g.bvec)[i];
    }
}

// One thread per belief node: re-evaluate it if one of its children improved in the previous sweep.
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void k_dp_sweep(DpConst g, uint8_t *__restrict__ dirty_in, uint8_t *__restrict__ dirty_out, uint32_t slot) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.n) return;
    if (!as_global(dirty_in)[i]) return;
    as_global(dirty_in)[i] = 0;
    const uint8_t type = as_global(g.types)[i];
    double ux, uy;
    uint32_t urow;
    dp_state<IMPLICIT>(g, i, ux, uy, urow);
    const unsigned long long c0 = as_global(g.child_off)[i], c1 = as_global(g.child_off)[i + 1];
    double alt;
    if (type == BG_ACTION) {
        alt = __builtin_huge_val();
        for (unsigned long long c = c0; c < c1; ++c) {
            const unsigned long long v = as_global(g.child_id)[c];
            double vx, vy;
            uint32_t vrow;
            dp_state<IMPLICIT>(g, v, vx, vy, vrow);
            const double cost = sqrt(dist2(ux, uy, vx, vy));          // norm2(u.state, v.state)
            const double a = cost + as_global(g.dist)[v];
            alt = a < alt ? a : alt;
        }
    } else if (type == BG_OBSERVATION) {
        alt = 0.0;
        for (unsigned long long c = c0; c < c1; ++c) {
            const unsigned long long v = as_global(g.child_id)[c];
            double vx, vy;
            uint32_t vrow;
            dp_state<IMPLICIT>(g, v, vx, vy, vrow);
            double p = 0.0;                                           // transition_probability (common.rs:187-190)
            for (uint32_t w = 0; w < g.nw; ++w)
                p = p + (as_global(g.beliefs)[(size_t)vrow * g.nw + w] > 0.0 ? as_global(g.beliefs)[(size_t)urow * g.nw + w] : 0.0);
            if (!(p > 0.0)) atomicOr(&g.flags[0], DP_ERR_ZERO_PROBABILITY);   // assert!(p > 0.0)
            const double cost = sqrt(dist2(ux, uy, vx, vy));
            alt = alt + p * (cost + as_global(g.dist)[v]);
        }
    } else {
        atomicOr(&g.flags[0], DP_ERR_UNKNOWN_TYPE);                   // "node type should be know at this stage!"
        return;
    }
    if (alt < as_global(g.dist)[i]) {
        as_global(g.dist)[i] = alt;
        for (unsigned long long e = as_global(g.par_off)[i]; e < as_global(g.par_off)[i + 1]; ++e) as_global(dirty_out)[as_global(g.par_id)[e]] = 1;
        as_global(g.flags)[1 + slot] = 1;
    }
}

// ------------------------------------------------------------------------------------------------ host side

struct DpState {
    bool valid = false;
    double *d_dist = nullptr;
    size_t n = 0;
    uint32_t sweeps = 0;
    double t_total = 0, t_device = 0;
    std::vector<void *> owned;                        // scratch of the last run (flags, finals); dist lives in a grow-only slot
    size_t dist_cap = 0;
    uint8_t *d_dirty[2] = {nullptr, nullptr};
    size_t dirty_cap = 0;
    uint32_t *d_flags = nullptr;
    unsigned long long *d_finals = nullptr;
    size_t finals_cap = 0;
    void release() { valid = false; }
    void free_device() {
        release();
        if (d_dist) (void)hipFree(d_dist);
        for (int k = 0; k < 2; ++k) if (d_dirty[k]) (void)hipFree(d_dirty[k]);
        if (d_flags) (void)hipFree(d_flags);
        if (d_finals) (void)hipFree(d_finals);
        d_dist = nullptr; d_dirty[0] = d_dirty[1] = nullptr; d_flags = nullptr; d_finals = nullptr;
        dist_cap = dirty_cap = finals_cap = 0;
    }
    ~DpState() { free_device(); }
};

constexpr uint32_t kDpGroup = 8;                      // sweeps between two looks at the "anything changed" flags

#define DP_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return PORRT_ERR_DEVICE; } \
    } while (0)

// Runs the sweeps on a graph whose arrays are already on the device (c.dist / c.flags are filled in here).
static int dp_run(DpState &st, DpConst c, bool implicit, const std::vector<unsigned long long> &finals, hipStream_t s, std::string &err) {
    st.release();
    const double t0 = bg_now();
    const size_t n = (size_t)c.n;
    if (st.dist_cap < n) {
        if (st.d_dist) (void)hipFree(st.d_dist);
        st.d_dist = nullptr; st.dist_cap = 0;
        DP_HIP(hipMalloc((void **)&st.d_dist, (n + n / 8 + 1) * sizeof(double)));
        st.dist_cap = n + n / 8 + 1;
    }
    if (st.dirty_cap < n) {
        for (int k = 0; k < 2; ++k) { if (st.d_dirty[k]) (void)hipFree(st.d_dirty[k]); st.d_dirty[k] = nullptr; }
        st.dirty_cap = 0;
        for (int k = 0; k < 2; ++k) DP_HIP(hipMalloc((void **)&st.d_dirty[k], n + n / 8 + 1));
        st.dirty_cap = n + n / 8 + 1;
    }
    if (!st.d_flags) DP_HIP(hipMalloc((void **)&st.d_flags, (1 + kDpGroup) * sizeof(uint32_t)));
    if (st.finals_cap < finals.size() + 1) {
        if (st.d_finals) (void)hipFree(st.d_finals);
        st.d_finals = nullptr; st.finals_cap = 0;
        DP_HIP(hipMalloc((void **)&st.d_finals, (finals.size() + 1) * 2 * sizeof(unsigned long long)));
        st.finals_cap = (finals.size() + 1) * 2;
    }
    c.dist = st.d_dist;
    c.flags = st.d_flags;
    hipEvent_t ev0, ev1;
    DP_HIP(hipEventCreate(&ev0)); DP_HIP(hipEventCreate(&ev1));
    DP_HIP(hipEventRecord(ev0, s));
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_dp_fill, grid, block, 0, s, st.d_dist, (unsigned long long)n, __builtin_huge_val());
    DP_HIP(hipMemsetAsync(st.d_dirty[0], 0, n, s));
    DP_HIP(hipMemsetAsync(st.d_dirty[1], 0, n, s));
    DP_HIP(hipMemsetAsync(st.d_flags, 0, (1 + kDpGroup) * sizeof(uint32_t), s));
    if (!finals.empty()) {
        DP_HIP(hipMemcpyAsync(st.d_finals, finals.data(), finals.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_dp_set_finals, dim3((unsigned)((finals.size() + 255) / 256)), block, 0, s, c, (const unsigned long long *)st.d_finals,
                           (unsigned long long)finals.size(), st.d_dirty[0]);
    }
    uint32_t sweeps = 0, h_flags[1 + kDpGroup];
    int cur = 0;
    for (bool more = !finals.empty(); more;) {
        DP_HIP(hipMemsetAsync(st.d_flags + 1, 0, kDpGroup * sizeof(uint32_t), s));
        for (uint32_t k = 0; k < kDpGroup; ++k, cur ^= 1) {
            if (implicit) hipLaunchKernelGGL(k_dp_sweep<true>, grid, block, 0, s, c, st.d_dirty[cur], st.d_dirty[cur ^ 1], k);
            else hipLaunchKernelGGL(k_dp_sweep<false>, grid, block, 0, s, c, st.d_dirty[cur], st.d_dirty[cur ^ 1], k);
        }
        sweeps += kDpGroup;
        DP_HIP(hipMemcpyAsync(h_flags, st.d_flags, sizeof h_flags, hipMemcpyDeviceToHost, s));
        DP_HIP(hipStreamSynchronize(s));
        if (h_flags[0]) break;
        more = h_flags[kDpGroup] != 0;                  // the last sweep of the group still improved something
        if (sweeps > 4u * 1000u * 1000u) { err = "conditional_dijkstra: no fixpoint after 4M sweeps"; return PORRT_ERR_DEVICE; }
    }
    DP_HIP(hipEventRecord(ev1, s));
    DP_HIP(hipMemcpyAsync(h_flags, st.d_flags, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    DP_HIP(hipStreamSynchronize(s));
    DP_HIP(hipGetLastError());
    float ms = 0;
    DP_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
    if (h_flags[0] & DP_ERR_UNKNOWN_TYPE) { err = "node type should be know at this stage! (belief_graph.rs:138)"; return PORRT_ERR_INVALID; }
    if (h_flags[0] & DP_ERR_ZERO_PROBABILITY) { err = "assert!(p > 0.0) failed (belief_graph.rs:128)"; return PORRT_ERR_INVALID; }
    st.n = n;
    st.sweeps = sweeps;
    st.t_device = 1e-3 * (double)ms;
    st.t_total = bg_now() - t0;
    st.valid = true;
    return PORRT_OK;
}

} // namespace porrt
