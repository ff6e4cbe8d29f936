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
        const uint32_t node = (uint32_t)i / g.B;                  // the context's graph has fewer than 2^32 belief nodes
        x = as_global(g.nx)[node]; y = as_global(g.ny)[node];
        row = (uint32_t)i - node * g.B;
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


// One belief node's children with what get_best_expected_children (belief_graph.rs:214-263) needs of each: id, expected
// cost, edge cost.  One workgroup; the policy walk on the host asks for one row at a time.
struct DpRowItem { double dist, cost; uint32_t child, pad; };
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void k_dp_row(DpConst g, unsigned long long bn, DpRowItem *__restrict__ out, uint32_t cap, uint32_t *__restrict__ count) {
    const unsigned long long c0 = as_global(g.child_off)[bn], c1 = as_global(g.child_off)[bn + 1];
    if (threadIdx.x == 0) { *count = (uint32_t)(c1 - c0); DpRowItem h; h.child = (uint32_t)(c1 - c0); h.pad = 0; h.dist = 0.0; h.cost = 0.0; out[0] = h; }   // item 0 = the count
    double ux, uy;
    uint32_t urow;
    dp_state<IMPLICIT>(g, bn, ux, uy, urow);
    for (unsigned long long c = c0 + threadIdx.x; c < c1 && c - c0 + 1 < cap; c += blockDim.x) {
        const unsigned long long v = as_global(g.child_id)[c];
        double vx, vy;
        uint32_t vrow;
        dp_state<IMPLICIT>(g, v, vx, vy, vrow);
        DpRowItem it;
        it.child = (uint32_t)v; it.pad = 0;
        it.dist = as_global(g.dist)[v];
        it.cost = sqrt(dist2(ux, uy, vx, vy));
        out[1 + (c - c0)] = it;
    }
}

// ------------------------------------------------------------------------------------------------------------
// The context's belief graph has more structure than an arbitrary one, and the fast path uses it:
//   * action edges stay inside one belief ("layer": the PTO graph filtered by what that belief allows), observation
//     edges lead to beliefs with fewer possible worlds;
//   * so the layers are solved level by level, by growing number of possible worlds: when a level starts, the value of
//     each of its observation nodes is a finished sum over deeper levels, and what remains is a shortest-path problem
//     per layer with those nodes and the finals as sources -- no layer ever relaxes on unfinished inputs;
//   * all layers of a level share the PTO adjacency: with the belief index fastest (beliefs renumbered so that a level is
//     a contiguous range), a wave works on one graph node for 64 beliefs -- the neighbour list and the edge weights are
//     the same for all lanes, the neighbours' costs are 64 consecutive doubles.
// Same fixpoint, same bits as the general sweeps; far fewer relaxations and coalesced ones.
// One adjacency entry as a sweep wants it: 16 bytes, one (scalar) load -- the neighbour, what decides whether a belief may use the edge
// (the neighbour's validity and the edge's, belief_graph.rs:114-121) and the edge's cost.
struct __attribute__((aligned(16))) DpEdge {
    uint32_t c;
    uint32_t vv;                       // vid[c] | adj_val[k] << 8
    double w;
};

struct DpLevelConst {
    BgConst g;                         // tables, adjacency, bit planes of the belief graph
    const struct DpEdge *adj_e;        // per adjacency entry: neighbour, the two validities it depends on, norm2(node, neighbour)
    const uint32_t *rank;              // [B] belief -> position in the level order
    const uint32_t *belief_at;         // [B] position -> belief
    double *dist_p;                    // [N][B] by position
    uint32_t *flags;                   // [1 + s]: sweep s of a group improved something
};

__global__ __launch_bounds__(256) void k_dp_edge_weights(BgConst g, DpEdge *__restrict__ e) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= g.N) return;
    const double x = as_global(g.nx)[n], y = as_global(g.ny)[n];
    for (unsigned long long k = as_global(g.adj_off)[n]; k < as_global(g.adj_off)[n + 1]; ++k) {
        const uint32_t c = as_global(g.adj_id)[k];
        DpEdge o;
        o.c = c;
        o.vv = (uint32_t)as_global(g.vid)[c] | (uint32_t)as_global(g.adj_val)[k] << 8;
        o.w = sqrt(dist2(x, y, as_global(g.nx)[c], as_global(g.ny)[c]));            // norm2(u.state, v.state), u = n
        e[k] = o;
    }
}

__global__ __launch_bounds__(256) void k_dp_level_finals(DpLevelConst L, const unsigned long long *__restrict__ finals, unsigned long long n_final) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_final) return;
    const unsigned long long i = as_global(finals)[k];
    as_global(L.dist_p)[(size_t)(i / L.g.B) * L.g.B + as_global(L.rank)[i % L.g.B]] = 0.0;
}

// A sweep's workgroup looks first at ONE byte: whether any row of its item was marked for this sweep (the byte holds the stamp of the sweep
// it was last marked for; two arrays alternate like the rows' marks, so nothing needs clearing) -- most workgroups of most sweeps end there.
__host__ __device__ inline uint8_t dp_stamp(uint32_t sweep) { return (uint8_t)(sweep % 255u + 1u); }

// thread -> (graph node n, position p in [p0, p0 + W)), position fastest
__device__ __forceinline__ bool dp_level_thread(const DpLevelConst &L, uint32_t p0, uint32_t W, uint32_t &n, uint32_t &p) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)L.g.N * W) return false;
    n = (uint32_t)t / W;                                      // N * W < 2^32
    p = p0 + ((uint32_t)t - n * W);
    return true;
}

// Start of a level: observation nodes get their value (a sum over finished, deeper levels).  Every node with a finite
// value (those and the finals) is a source: its neighbours are marked for the first sweep.
// (and the byte of each marked row's item -- parts = 1: an item is 256 beliefs of one node; 4: 64 rows -- gets the first sweep's stamp)
__global__ __launch_bounds__(256) void k_dp_level_init(DpLevelConst L, uint32_t p0, uint32_t W, uint8_t *__restrict__ dirty, uint8_t *__restrict__ item,
                                                       uint32_t parts, uint32_t sweep) {
    uint32_t n, p;
    if (!dp_level_thread(L, p0, W, n, p)) return;
    const BgConst &g = L.g;
    const size_t ip = (size_t)n * g.B + p;
    const uint32_t b = as_global(L.belief_at)[p], vn = as_global(g.vid)[n];
    const unsigned long long cb = as_global(g.compat)[b];
    if (!((cb >> vn) & 1ull)) return;
    double v = as_global(L.dist_p)[ip];
    const size_t bit = (size_t)n * g.B + b;
    if (v != 0.0 && ((as_global(g.obs_bits)[bit >> 6] >> (bit & 63)) & 1ull)) {   // (a final node stays at 0)
        const size_t trow = (size_t)as_global(g.mask_idx)[n] * g.B + b;
        double alt = 0.0;
        for (uint32_t k = as_global(g.obs_off)[trow]; k < as_global(g.obs_off)[trow + 1]; ++k) {
            const uint32_t c = as_global(g.obs_child)[k];
            if ((as_global(g.compat)[c] >> vn) & 1ull)                              // cost(u, vv) = norm2 of equal states = 0
                alt = alt + as_global(g.obs_p)[k] * (0.0 + as_global(L.dist_p)[(size_t)n * g.B + as_global(L.rank)[c]]);
        }
        as_global(L.dist_p)[ip] = alt;
        v = alt;
    }
    if (v < __builtin_huge_val())
        for (unsigned long long k = as_global(g.adj_off)[n]; k < as_global(g.adj_off)[n + 1]; ++k) {
            const uint32_t c = as_global(g.adj_id)[k];
            as_global(dirty)[(size_t)c * g.B + p] = 1;
            if (parts == 1u) as_global(item)[c * ((W + 255u) / 256u) + (p - p0) / 256u] = dp_stamp(sweep);
            else as_global(item)[((size_t)c * W + (p - p0)) >> 6] = dp_stamp(sweep);
        }
}

// One sweep over a level: action nodes whose neighbours improved take the best of (edge + neighbour).  Four lanes share a
// row (lane = part * 16 + row-in-wave: the 16 rows of a wave are consecutive beliefs of one node, so each part still
// reads 128 contiguous bytes per neighbour) and split its neighbour list, which quarters the chain of dependent loads
// that sets the duration of a sweep over a small level (13.8 ms instead of 23 for the 255-belief problem); a large
// level is bandwidth bound and keeps one lane per row (PARTS = 1: 512 contiguous bytes per neighbour).
// (`item` = what one workgroup of a sweep takes: 256 beliefs of one node, or 64 rows in four parts; a byte per item says whether any of
// its rows is marked for the next sweep.)
template <uint32_t kDpParts>
__device__ __forceinline__ bool dp_sweep_item(const DpLevelConst &L, uint32_t p0, uint32_t W, uint8_t *__restrict__ dirty_in, uint8_t *__restrict__ dirty_out,
                                              uint8_t *__restrict__ item_out, uint8_t stamp, uint32_t item, uint32_t &evaluated) {
    const BgConst &g = L.g;
    constexpr uint32_t kDpRowsPerWave = 64 / kDpParts;
    const uint32_t lane = threadIdx.x & 63u, part = lane / kDpRowsPerWave;
    bool in_range;
    uint32_t n, p;
    if (kDpParts == 1) {
        // a workgroup = 256 consecutive beliefs of ONE graph node: the node index is a scalar, no per-thread division
        const uint32_t chunks = (W + 255u) / 256u, off = (item % chunks) * 256u + threadIdx.x;
        n = item / chunks;
        in_range = off < W;
        p = p0 + (in_range ? off : 0u);
    } else {
        const size_t wave = ((size_t)item * 256u + threadIdx.x) / 64;
        const size_t row = wave * kDpRowsPerWave + (lane % kDpRowsPerWave);
        in_range = row < (size_t)g.N * W;
        n = in_range ? (uint32_t)row / W : 0u;
        p = p0 + (in_range ? (uint32_t)row - n * W : 0u);
    }
    // first trip: everything that depends on (n, p) alone -- the row's mark, its belief, the node's validity and list
    const size_t ip = (size_t)n * g.B + p;
    const bool was_dirty = in_range && as_global(dirty_in)[ip] != 0;
    const uint32_t b = as_global(L.belief_at)[p], vn = as_global(g.vid)[n];
    const unsigned long long a0 = as_global(g.adj_off)[n], a1 = as_global(g.adj_off)[n + 1];
    if (!__ballot(was_dirty)) return false;
    // second trip, marked rows only: what decides whether the row moves at all (observation nodes and finals do not), its cost -- and,
    // below, the first batch of its list, which waits for none of these
    bool active = false;
    double old = 0.0;
    unsigned long long cb = 0;
    if (was_dirty) {
        cb = as_global(g.compat)[b];
        const size_t bit = (size_t)n * g.B + b;
        old = as_global(L.dist_p)[ip];
        active = ((cb >> vn) & 1ull) && !((as_global(g.obs_bits)[bit >> 6] >> (bit & 63)) & 1ull) && old != 0.0;
    }
    evaluated += (active && part == 0) ? 1u : 0u;
    const DpEdge *__restrict__ E = L.adj_e;
    double best = __builtin_huge_val();
    // kDpBatch neighbours at a time: their entries first (one load each; scalar when the node is the workgroup's), then their costs, all in
    // flight together -- nothing in a batch waits for anything else in it, whatever the number of validities.  (An index past the row's
    // end repeats its last entry: min does not mind.)
    constexpr uint32_t kDpBatch = kDpParts == 1 ? 12u : 10u;
    // (a small level is a chain of trips: there the list is fetched for every marked row, `active` -- known a trip later -- gates the
    // result only; a wide level is bandwidth: rows that cannot move fetch nothing)
    if ((kDpParts == 1 ? active : was_dirty) && a1 > a0) {
        const unsigned long long alast = a1 - 1;
        for (unsigned long long k = a0 + part; k < a1; k += (unsigned long long)kDpParts * kDpBatch) {
            uint32_t ec[kDpBatch], ev[kDpBatch];
            double ew[kDpBatch], dv[kDpBatch];
#pragma unroll
            for (uint32_t u = 0; u < kDpBatch; ++u) {
                const unsigned long long kk = k + (unsigned long long)u * kDpParts;
                const auto ep = as_global(E) + (kk < alast ? kk : alast);
                ec[u] = ep->c; ev[u] = ep->vv; ew[u] = ep->w;
            }
#pragma unroll
            for (uint32_t u = 0; u < kDpBatch; ++u) dv[u] = as_global(L.dist_p)[(size_t)ec[u] * g.B + p];
#pragma unroll
            for (uint32_t u = 0; u < kDpBatch; ++u) {
                const bool ok = ((cb >> (ev[u] & 0xFFu)) & (cb >> (ev[u] >> 8)) & 1ull) != 0;     // (one validity: bit 0 of every belief)
                const double a = ew[u] + dv[u];
                best = (ok && a < best) ? a : best;
            }
        }
    }
    for (uint32_t d = kDpRowsPerWave; d < 64; d <<= 1) {                            // the parts of a row sit 16 lanes apart
        const double o = __shfl_xor(best, d, 64);
        best = o < best ? o : best;
    }
    const bool improved = active && best < old;
    if (improved) {
        if (part == 0) as_global(L.dist_p)[ip] = best;
        for (unsigned long long k = a0 + part; k < a1; k += kDpParts) {             // whoever may use this node as a child
            const uint32_t c = as_global(E)[k].c;
            as_global(dirty_out)[(size_t)c * g.B + p] = 1;
            if (kDpParts == 1) as_global(item_out)[c * ((W + 255u) / 256u) + (p - p0) / 256u] = stamp;       // ... and the item that row belongs to
            else as_global(item_out)[((size_t)c * W + (p - p0)) >> 6] = stamp;
        }
    }
    if (was_dirty && part == 0) as_global(dirty_in)[ip] = 0;
    return improved;
}

template <uint32_t kDpParts>
__global__ __launch_bounds__(256) void k_dp_level_sweep(DpLevelConst L, uint32_t p0, uint32_t W, uint8_t *__restrict__ dirty_in,
                                                        uint8_t *__restrict__ dirty_out, const uint8_t *__restrict__ item_in, uint8_t *__restrict__ item_out,
                                                        uint32_t sweep, uint32_t slot) {
    if (as_global(item_in)[blockIdx.x] != dp_stamp(sweep)) return;
    uint32_t ev = 0;
    if (dp_sweep_item<kDpParts>(L, p0, W, dirty_in, dirty_out, item_out, dp_stamp(sweep + 1u), blockIdx.x, ev)) as_global(L.flags)[1 + slot] = 1;
}

// dist[n * B + b] = dist_p[n * B + rank[b]]
__global__ __launch_bounds__(256) void k_dp_unpermute(DpLevelConst L, double *__restrict__ dist) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)L.g.N * L.g.B) return;
    const uint32_t n = (uint32_t)i / L.g.B, b = (uint32_t)i - n * L.g.B;
    as_global(dist)[i] = as_global(L.dist_p)[(size_t)n * L.g.B + as_global(L.rank)[b]];
}

// ------------------------------------------------------------------------------------------------ host side

struct DpState {
    bool valid = false;
    double *d_dist = nullptr;
    size_t n = 0;
    uint32_t sweeps = 0;
    unsigned long long sweep_rows = 0;                // rows (belief nodes) the sweeps passed over, summed over the sweeps
    double t_total = 0, t_device = 0;
    std::vector<void *> owned;                        // scratch of the last run (flags, finals); dist lives in a grow-only slot
    size_t dist_cap = 0;
    uint8_t *d_dirty[2] = {nullptr, nullptr};
    size_t dirty_cap = 0;
    uint32_t *d_flags = nullptr;
    uint32_t *h_pin = nullptr;                        // pinned: where the flags are read (a copy into pageable memory costs several times as much)
    uint8_t *d_item[2] = {nullptr, nullptr};          // a byte per item (= workgroup of a sweep) of a level: the stamp of the sweep it was last marked for
    size_t item_cap = 0;
    unsigned long long *d_finals = nullptr;
    size_t finals_cap = 0;
    DpConst last{};                                   // the graph of the last run (device pointers), for the policy walk
    DpRowItem *d_row = nullptr;
    double *d_dist_l = nullptr;                       // layered evaluation: [B][N]
    size_t layer_cap = 0;
    void *d_aux = nullptr;                            // edge weights, belief order, flag scratch
    size_t aux_cap = 0;
    bool layered = false;
    uint32_t *d_row_count = nullptr;
    std::vector<uint64_t> pol_original;               // extract_policy: per policy node, in add_node order
    std::vector<int64_t> pol_parent;
    std::vector<uint8_t> pol_leaf;
    bool have_policy = false;
    void release() { valid = false; have_policy = false; }
    void free_device() {
        release();
        if (d_dist) (void)hipFree(d_dist);
        for (int k = 0; k < 2; ++k) if (d_dirty[k]) (void)hipFree(d_dirty[k]);
        if (d_flags) (void)hipFree(d_flags);
        for (int k = 0; k < 2; ++k) { if (d_item[k]) (void)hipFree(d_item[k]); d_item[k] = nullptr; }
        item_cap = 0;
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr;
        if (d_finals) (void)hipFree(d_finals);
        if (d_row) (void)hipFree(d_row);
        if (d_dist_l) (void)hipFree(d_dist_l);
        if (d_aux) (void)hipFree(d_aux);
        d_dist_l = nullptr; d_aux = nullptr; layer_cap = aux_cap = 0;
        if (d_row_count) (void)hipFree(d_row_count);
        d_row = nullptr; d_row_count = nullptr;
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
    ScopedEvents<2> evs;
    DP_HIP(evs.create());
    hipEvent_t ev0 = evs.e[0], ev1 = evs.e[1];
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
    if (!st.h_pin) DP_HIP(hipHostMalloc((void **)&st.h_pin, (1 + kDpGroup) * sizeof(uint32_t), hipHostMallocDefault));
    uint32_t sweeps = 0, *h_flags = st.h_pin;
    unsigned long long sweep_rows = 0;
    int cur = 0;
    for (bool more = !finals.empty(); more;) {
        DP_HIP(hipMemsetAsync(st.d_flags + 1, 0, kDpGroup * sizeof(uint32_t), s));
        for (uint32_t k = 0; k < kDpGroup; ++k, cur ^= 1) {
            if (implicit) hipLaunchKernelGGL(k_dp_sweep<true>, grid, block, 0, s, c, st.d_dirty[cur], st.d_dirty[cur ^ 1], k);
            else hipLaunchKernelGGL(k_dp_sweep<false>, grid, block, 0, s, c, st.d_dirty[cur], st.d_dirty[cur ^ 1], k);
        }
        sweeps += kDpGroup;
        sweep_rows += (unsigned long long)n * kDpGroup;
        DP_HIP(hipMemcpyAsync(h_flags, st.d_flags, (1 + kDpGroup) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
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
    if (h_flags[0] & DP_ERR_UNKNOWN_TYPE) { err = "node type should be know at this stage! (belief_graph.rs:138)"; return PORRT_ERR_INVALID; }
    if (h_flags[0] & DP_ERR_ZERO_PROBABILITY) { err = "assert!(p > 0.0) failed (belief_graph.rs:128)"; return PORRT_ERR_INVALID; }
    st.n = n;
    st.last = c;
    st.layered = false;
    st.sweeps = sweeps;
    st.sweep_rows = sweep_rows;
    st.t_device = 1e-3 * (double)ms;
    st.t_total = bg_now() - t0;
    st.valid = true;
    return PORRT_OK;
}

// Level-by-level evaluation on the context's belief graph (see above).  Needs what the build left on the device.
static int dp_run_layered(DpState &st, BeliefGraphState &bg, DpConst c, const std::vector<unsigned long long> &finals, hipStream_t s, std::string &err) {
    st.release();
    const double t0 = bg_now();
    const size_t N = bg.N, B = bg.B, n = N * B;
    if (st.dist_cap < n) {
        if (st.d_dist) (void)hipFree(st.d_dist);
        st.d_dist = nullptr; st.dist_cap = 0;
        DP_HIP(hipMalloc((void **)&st.d_dist, (n + n / 8 + 1) * sizeof(double)));
        st.dist_cap = n + n / 8 + 1;
    }
    if (st.layer_cap < n) {
        if (st.d_dist_l) (void)hipFree(st.d_dist_l);
        st.d_dist_l = nullptr; st.layer_cap = 0;
        DP_HIP(hipMalloc((void **)&st.d_dist_l, (n + n / 8 + 1) * sizeof(double)));
        st.layer_cap = n + n / 8 + 1;
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
    // beliefs ordered by number of possible worlds, fewest first; a level = beliefs of equal count = a range of positions
    std::vector<uint32_t> belief_at(B), rank(B);
    for (size_t b = 0; b < B; ++b) belief_at[b] = (uint32_t)b;
    std::stable_sort(belief_at.begin(), belief_at.end(), [&](uint32_t a, uint32_t b2) { return bg.support[a] < bg.support[b2]; });
    for (size_t p = 0; p < B; ++p) rank[belief_at[p]] = (uint32_t)p;
    std::vector<std::pair<uint32_t, uint32_t>> levels;              // [first, last) positions
    for (size_t i = 0; i < B;) {
        size_t j = i;
        while (j < B && bg.support[belief_at[j]] == bg.support[belief_at[i]]) ++j;
        levels.push_back({(uint32_t)i, (uint32_t)j});
        i = j;
    }
    unsigned long long n_adj = 0;
    DP_HIP(hipMemcpy(&n_adj, bg.last.adj_off + N, sizeof n_adj, hipMemcpyDeviceToHost));
    const size_t aux_bytes = (size_t)n_adj * sizeof(DpEdge) + 2 * B * sizeof(uint32_t) + 64;
    if (st.aux_cap < aux_bytes) {
        if (st.d_aux) (void)hipFree(st.d_aux);
        st.d_aux = nullptr; st.aux_cap = 0;
        DP_HIP(hipMalloc(&st.d_aux, aux_bytes + aux_bytes / 8));
        st.aux_cap = aux_bytes + aux_bytes / 8;
    }
    DpEdge *d_w = (DpEdge *)st.d_aux;
    uint32_t *d_rank = (uint32_t *)(d_w + n_adj), *d_belief_at = d_rank + B;
    ScopedEvents<2> evs;
    DP_HIP(evs.create());
    hipEvent_t ev0 = evs.e[0], ev1 = evs.e[1];
    DP_HIP(hipEventRecord(ev0, s));
    DP_HIP(hipMemcpyAsync(d_rank, rank.data(), B * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    DP_HIP(hipMemcpyAsync(d_belief_at, belief_at.data(), B * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    DP_HIP(hipMemsetAsync(st.d_dirty[0], 0, n, s));
    DP_HIP(hipMemsetAsync(st.d_dirty[1], 0, n, s));
    DP_HIP(hipMemsetAsync(st.d_flags, 0, (1 + kDpGroup) * sizeof(uint32_t), s));
    hipLaunchKernelGGL(k_dp_edge_weights, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, bg.last, d_w);
    hipLaunchKernelGGL(k_dp_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, st.d_dist_l, (unsigned long long)n, __builtin_huge_val());
    DpLevelConst L{};
    L.g = bg.last; L.adj_e = d_w; L.rank = d_rank; L.belief_at = d_belief_at; L.dist_p = st.d_dist_l; L.flags = st.d_flags;
    if (!finals.empty()) {
        DP_HIP(hipMemcpyAsync(st.d_finals, finals.data(), finals.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_dp_level_finals, dim3((unsigned)((finals.size() + 255) / 256)), dim3(256), 0, s, L, (const unsigned long long *)st.d_finals,
                           (unsigned long long)finals.size());
    }
    if (!st.h_pin) DP_HIP(hipHostMalloc((void **)&st.h_pin, (1 + kDpGroup) * sizeof(uint32_t), hipHostMallocDefault));
    uint32_t sweeps = 0, *h_flags = st.h_pin, prev_level_sweeps = 0;
    unsigned long long sweep_rows = 0;
    {
        size_t max_items = 0;
        for (auto &lv : levels) {
            const size_t W = lv.second - lv.first, rows = N * W;
            const size_t items = rows < (2u << 20) ? (rows * 4 + 255) / 256 : N * ((W + 255) / 256);
            max_items = items > max_items ? items : max_items;
        }
        if (st.item_cap < max_items) {
            for (int k = 0; k < 2; ++k) { if (st.d_item[k]) (void)hipFree(st.d_item[k]); st.d_item[k] = nullptr; }
            st.item_cap = 0;
            for (int k = 0; k < 2; ++k) DP_HIP(hipMalloc((void **)&st.d_item[k], max_items + max_items / 8 + 64));
            st.item_cap = max_items + max_items / 8 + 64;
        }
        // (stamps are never 0: a cleared byte matches no sweep.  The stamp of a sweep repeats after 255 sweeps and the bytes outlive levels --
        // an old byte that matches costs its workgroup a look at its rows' marks, which decide)
        DP_HIP(hipMemsetAsync(st.d_item[0], 0, st.item_cap, s));
        DP_HIP(hipMemsetAsync(st.d_item[1], 0, st.item_cap, s));
    }
    for (auto &lv : levels) {
        const uint32_t p0 = lv.first, W = lv.second - lv.first;
        const dim3 grid((unsigned)(((size_t)N * W + 255) / 256)), block(256);
        const bool split = (size_t)N * W < (2u << 20);                                 // small level: latency bound
        const dim3 sgrid(split ? (unsigned)(((size_t)N * W * 4 + 255) / 256) : (unsigned)(N * ((W + 255) / 256)));
        hipLaunchKernelGGL(k_dp_level_init, grid, block, 0, s, L, p0, W, st.d_dirty[0], st.d_item[sweeps & 1u], split ? 4u : 1u, sweeps);
        // Groups of sweeps between two looks from the host at "did the group's last sweep still improve something".  Levels of one graph
        // take about as many sweeps as the level before them (the graph's depth in hops): the first group of a level is that count less
        // one group, the following ones kDpGroup -- a look costs a synchronisation, a sweep past the fixpoint ends at its item bytes.
        int cur = 0;
        uint32_t level_sweeps = 0;
        uint32_t G = prev_level_sweeps > 2u * kDpGroup ? ((prev_level_sweeps - kDpGroup) / kDpGroup) * kDpGroup : kDpGroup;
        for (bool more = true; more; G = kDpGroup) {
            DP_HIP(hipMemsetAsync(st.d_flags + 1, 0, kDpGroup * sizeof(uint32_t), s));
            for (uint32_t k = 0; k < G; ++k, cur ^= 1) {
                const uint32_t slot = k + 1u == G ? kDpGroup - 1u : 0u;            // (the host reads the last sweep's word only)
                if (split) hipLaunchKernelGGL(k_dp_level_sweep<4>, sgrid, block, 0, s, L, p0, W, st.d_dirty[cur], st.d_dirty[cur ^ 1],
                                              (const uint8_t *)st.d_item[(sweeps + k) & 1u], st.d_item[(sweeps + k + 1u) & 1u], sweeps + k, slot);
                else hipLaunchKernelGGL(k_dp_level_sweep<1>, sgrid, block, 0, s, L, p0, W, st.d_dirty[cur], st.d_dirty[cur ^ 1],
                                        (const uint8_t *)st.d_item[(sweeps + k) & 1u], st.d_item[(sweeps + k + 1u) & 1u], sweeps + k, slot);
            }
            sweeps += G;
            level_sweeps += G;
            sweep_rows += (unsigned long long)N * W * G;
            DP_HIP(hipMemcpyAsync(h_flags, st.d_flags, (1 + kDpGroup) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            DP_HIP(hipStreamSynchronize(s));
            more = h_flags[kDpGroup] != 0;
            if (sweeps > 4u * 1000u * 1000u) { err = "conditional_dijkstra: no fixpoint after 4M sweeps"; return PORRT_ERR_DEVICE; }
        }
        prev_level_sweeps = level_sweeps;
    }
    hipLaunchKernelGGL(k_dp_unpermute, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, L, st.d_dist);
    DP_HIP(hipEventRecord(ev1, s));
    DP_HIP(hipStreamSynchronize(s));
    DP_HIP(hipGetLastError());
    float ms = 0;
    DP_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    c.dist = st.d_dist;
    c.flags = st.d_flags;
    st.n = n;
    st.last = c;
    st.sweeps = sweeps;
    st.sweep_rows = sweep_rows;
    st.layered = true;
    st.t_device = 1e-3 * (double)ms;
    st.t_total = bg_now() - t0;
    st.valid = true;
    return PORRT_OK;
}

constexpr uint32_t kDpRowCap = 1u << 16;

// extract_policy (belief_graph.rs:177-212) with get_best_expected_children (214-263): a depth-first walk from belief
// node 0 that keeps, per target belief, the child of least p * (cost + expected cost).  Sequential pointer chasing over
// a few hundred nodes: host code, one row of the device graph fetched per step (k_dp_row).
// belief_of(i) = clustering key of node i; p_of(parent, child) = transition_probability of their belief vectors.
template <class BeliefOf, class ProbOf>
static int dp_extract_policy(DpState &st, bool implicit, BeliefOf belief_of, ProbOf p_of, hipStream_t s, std::string &err) {
    if (!st.valid) { err = "extract_policy: compute the expected costs first"; return PORRT_ERR_INVALID; }
    if (!st.d_row) {
        DP_HIP(hipMalloc((void **)&st.d_row, kDpRowCap * sizeof(DpRowItem)));
        DP_HIP(hipMalloc((void **)&st.d_row_count, sizeof(uint32_t)));
    }
    st.pol_original.assign(1, 0);
    st.pol_parent.assign(1, -1);
    st.pol_leaf.assign(1, 0);
    std::vector<std::pair<uint64_t, uint64_t>> lifo{{0, 0}};       // (policy node, belief node)
    std::vector<DpRowItem> row, fetch;
    double root_cost = 0.0;
    DP_HIP(hipMemcpy(&root_cost, st.d_dist, sizeof(double), hipMemcpyDeviceToHost));
    if (!(root_cost < __builtin_huge_val())) { err = "extract_policy: no policy from the root (its expected cost is infinite; the reference does not terminate here)"; return PORRT_ERR_INVALID; }
    std::map<uint64_t, double> dist_of{{0, root_cost}};
    while (!lifo.empty()) {
        const auto [pol, bn] = lifo.back();
        lifo.pop_back();
        // The reference's walk has no memory (belief_graph.rs:193-213): where the best child of a node leads back to a node on the
        // walk's own path -- two graph nodes at one place are joined by edges of cost 0 and may each be the other's first best child --
        // it never ends.  Here that is an error at once, not a policy of 2^24 nodes.
        for (int64_t up = st.pol_parent[pol]; up >= 0; up = st.pol_parent[(size_t)up])
            if (st.pol_original[(size_t)up] == bn) {
                err = "extract_policy: the walk returns to a belief node on its own path (zero-cost edges between nodes at one place); the reference does not terminate here";
                return PORRT_ERR_INVALID;
            }
        if (implicit) hipLaunchKernelGGL(k_dp_row<true>, dim3(1), dim3(256), 0, s, st.last, (unsigned long long)bn, st.d_row, kDpRowCap, st.d_row_count);
        else hipLaunchKernelGGL(k_dp_row<false>, dim3(1), dim3(256), 0, s, st.last, (unsigned long long)bn, st.d_row, kDpRowCap, st.d_row_count);
        // one copy brings the count (item 0) and the first 255 children; longer rows need a second one
        constexpr uint32_t kFirst = 256;
        std::vector<DpRowItem> &buf = fetch;
        buf.resize(kFirst);
        DP_HIP(hipMemcpyAsync(buf.data(), st.d_row, kFirst * sizeof(DpRowItem), hipMemcpyDeviceToHost, s));
        DP_HIP(hipStreamSynchronize(s));
        const uint32_t cnt = buf[0].child;
        if (cnt + 1 > kDpRowCap) { err = "extract_policy: a belief node with more than 65535 children"; return PORRT_ERR_CAPACITY; }
        row.assign(buf.begin() + 1, buf.begin() + 1 + std::min<uint32_t>(cnt, kFirst - 1));
        if (cnt > kFirst - 1) {
            row.resize(cnt);
            DP_HIP(hipMemcpy(row.data() + (kFirst - 1), st.d_row + kFirst, (cnt - (kFirst - 1)) * sizeof(DpRowItem), hipMemcpyDeviceToHost));
        }
        // BTreeMap<belief_id, Vec<ChildWithCosts>>: clusters in ascending key order, members in children order
        std::map<uint32_t, std::vector<uint32_t>> clusters;
        for (uint32_t k = 0; k < cnt; ++k) clusters[belief_of(row[k].child)].push_back(k);
        const double dist_bn = dist_of.count(bn) ? dist_of[bn] : 0.0;
        for (auto &kv : clusters) {
            uint32_t best = kv.second[0];
            const double p = p_of(bn, row[best].child);
            if (!(p > 0.0)) { err = "assert!(p > 0.0) failed (belief_graph.rs:244)"; return PORRT_ERR_INVALID; }
            double best_cost = __builtin_huge_val();
            for (uint32_t k : kv.second) {
                const double cost = p * (row[k].cost + row[k].dist);
                if (cost < best_cost) { best_cost = cost; best = k; }
            }
            if (!(p * row[best].dist <= dist_bn)) { err = "assert!(p * expected_costs_to_goals[best_id] <= expected_costs_to_goals[belief_node_id]) failed (belief_graph.rs:257)"; return PORRT_ERR_INVALID; }
            const bool leaf = row[best].dist == 0.0;
            const uint64_t id = st.pol_original.size();
            st.pol_original.push_back(row[best].child);
            st.pol_parent.push_back((int64_t)pol);
            st.pol_leaf.push_back(leaf ? 1 : 0);
            dist_of[row[best].child] = row[best].dist;
            if (!leaf) lifo.push_back({id, (uint64_t)row[best].child});
            if (st.pol_original.size() > (1u << 24)) { err = "extract_policy: more than 2^24 policy nodes"; return PORRT_ERR_CAPACITY; }
        }
    }
    st.have_policy = true;
    return PORRT_OK;
}

} // namespace porrt
