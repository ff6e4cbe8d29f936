// porrt_belief.hpp -- belief-space expansion on the device: PTO::build_belief_graph (src/pto.rs:185-259).
//
// The reference builds, for every (graph node, reachable belief state) pair, a belief node with its children and
// parents lists: observation edges where the node sees a zone that splits the belief (observe, src/map_io.rs:281-300 /
// src/map_shelves_io.rs:242-265), geometric ("action") edges along the PTO graph where belief and edge validity are
// compatible (common.rs:256-276).  N_nodes x N_beliefs x degree work, the reference's measured bottleneck.
//
// Split here:
//   host   reachable_belief_states (map_io.rs:515-546), hash (common.rs:352-355), compatibility bits, and the
//          observation fold as a TABLE over (set of visible zones, belief): what observe() returns depends on the
//          node only through the set of zones it sees, and a graph has a handful of distinct sets;
//   device k_bg_vismask   which zones each node sees (distance test + one raycast per node and zone),
//          k_bg_children_count / k_bg_parents_count   node type, number of children / parents of every belief node,
//          k_scan_*       offsets,
//          k_bg_fill      the children / parents lists, written as contiguous runs per wave.
// Lists come out in the reference's Vec::push order: children = observation children in fold order, or the PTO
// adjacency order filtered; parents = observation parents by ascending belief id, then action parents by ascending
// graph node id (the order of the two loops at pto.rs:211-257).  Belief node id = node * n_beliefs + belief
// (pto.rs:198-201 adds them in that order); node_to_belief_nodes[id][b] is Some(id * B + b) iff compatible.
//
// The result stays on the device (CSR: 64-bit offsets, 32-bit ids) for the rows that consume it next
// (conditional_dijkstra); the getters of include/porrt_hip.h copy it out.
#pragma once
#include "porrt_device.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace porrt {

enum : uint8_t { BG_UNKNOWN = 0, BG_ACTION = 1, BG_OBSERVATION = 2 };     // belief_graph.rs:13-17

// what the kernels read; all pointers are device memory
struct BgConst {
    uint32_t N, B, nz, n_masks, n_validities;
    const double *nx, *ny;            // node coordinates
    const uint8_t *vid;               // node validity id
    const double *zone_xy;            // [nz][2]
    double visibility;
    const unsigned long long *compat; // [B]: bit v = belief compatible with world validity v
    const uint32_t *mask_idx;         // [N]: index of the node's visible-zone set
    const uint32_t *obs_off, *obs_child;      // [(n_masks*B)+1], children beliefs of (mask, belief) in fold order
    const double *obs_p;                      // per table entry: transition_probability(belief, posterior) (common.rs:187-190)
    const uint32_t *robs_off, *robs_par;      // reverse: parent beliefs of (mask, belief), ascending
    const unsigned long long *adj_off;        // [N+1] PTO adjacency in push order
    const uint32_t *adj_id;
    const uint8_t *adj_val;
    const unsigned long long *radj_off;       // [N+1] the same edges by ascending neighbour id
    const uint32_t *radj_id;
    const uint8_t *radj_val;
    uint8_t *types;                   // [N*B]
    unsigned long long *obs_bits;     // [N*B/64+2]: bit node*B + b = belief node (node, b) is an observation node
    unsigned long long *obs_bits_t;   // the same bits transposed: bit b*N + node (the parents fill reads one belief's row)
    uint32_t *deg;                    // [N*B] scratch of the count passes
    unsigned long long *child_off, *par_off;  // [N*B+1]
    uint32_t *child_id, *par_id;
};

// zones seen from each node: bit z of vis[node]  (map_io.rs:287-289, map_shelves_io.rs:259-265)
__global__ __launch_bounds__(256) void k_bg_vismask(const RunConst *__restrict__ rcp, BgConst g, unsigned long long *__restrict__ vis,
                                                    uint32_t *__restrict__ err_out) {
    const RunConst &rc = *rcp;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)g.N * g.nz) return;
    const uint32_t node = (uint32_t)(t / g.nz), z = (uint32_t)(t % g.nz);
    const double x = as_global(g.nx)[node], y = as_global(g.ny)[node];
    const double zx = as_global(g.zone_xy)[2 * z], zy = as_global(g.zone_xy)[2 * z + 1];
    const double D = sqrt(dist2(x, y, zx, zy));
    if (!(D < g.visibility)) return;
    uint32_t err = 0;
    GlobalGrid grid{rc.cls, rc.W};
    const int c = traversed_class(rc, grid, x, y, zx, zy, &err);
    if (err) atomicOr(err_out, err);
    if (c != CLS_HIGH) atomicOr(&vis[node], 1ull << z);
}

// Sets the observation bit of belief node i = node * B + b in both planes (zeroed before the launch).  The lanes of a
// wave hold consecutive i, so their bits fall into at most two words of the (node, b) plane: one ballot, two atomics.
__device__ __forceinline__ void mark_obs(const BgConst &g, size_t i, uint32_t node, uint32_t b, bool is_obs) {
    const unsigned long long ballot = __ballot(is_obs);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long first = __ballot(true);
    if (ballot && lane == (uint32_t)__ffsll((long long)first) - 1u) {
        const size_t i0 = i - lane;                         // flat index lane 0 holds (or would hold)
        const uint32_t sh = (uint32_t)(i0 & 63);
        const unsigned long long lo = ballot << sh, hi = sh ? ballot >> (64 - sh) : 0ull;
        if (lo) atomicOr(&g.obs_bits[i0 >> 6], lo);
        if (hi) atomicOr(&g.obs_bits[(i0 >> 6) + 1], hi);
    }
    if (is_obs) {
        const size_t t = (size_t)b * g.N + node;
        atomicOr(&g.obs_bits_t[t >> 6], 1ull << (t & 63));
    }
}

// Count pass, one thread per belief node (node, b): its type and the number of its children.
__global__ __launch_bounds__(256) void k_bg_children_count(BgConst g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)g.N * g.B) return;
    const uint32_t i32 = (uint32_t)i, node = i32 / g.B, b = i32 - node * g.B;          // N * B < 2^32 (checked by the build)
    const uint32_t v = as_global(g.vid)[node];
    const unsigned long long cb = as_global(g.compat)[b];
    uint32_t cnt = 0;
    uint8_t type = BG_UNKNOWN;
    if ((cb >> v) & 1ull) {
        // observation edges (pto.rs:211-233): (node, b) -> (node, b') for every posterior b' != b of observe(node, b)
        const size_t row = (size_t)as_global(g.mask_idx)[node] * g.B + b;
        const uint32_t o0 = as_global(g.obs_off)[row], o1 = as_global(g.obs_off)[row + 1];
        for (uint32_t k = o0; k < o1; ++k) cnt += (uint32_t)((as_global(g.compat)[as_global(g.obs_child)[k]] >> v) & 1ull);
        if (cnt) type = BG_OBSERVATION;
        else {
            // action edges (pto.rs:235-257): (node, b) -> (child, b) where the child node and the edge are compatible with b
            const unsigned long long a0 = as_global(g.adj_off)[node], a1 = as_global(g.adj_off)[node + 1];
#pragma unroll 4
            for (unsigned long long k = a0; k < a1; ++k) {
                const uint32_t ev = as_global(g.adj_val)[k], cv = as_global(g.vid)[as_global(g.adj_id)[k]];
                cnt += (uint32_t)(((cb >> cv) & 1ull) & ((cb >> ev) & 1ull));
            }
            if (cnt) type = BG_ACTION;
        }
    }
    as_global(g.types)[i] = type;
    as_global(g.deg)[i] = cnt;
    mark_obs(g, i, node, b, type == BG_OBSERVATION);
}

// Count pass, one thread per belief node (node, b): how many point at it.  Needs the types of k_bg_children_count.
__global__ __launch_bounds__(256) void k_bg_parents_count(BgConst g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)g.N * g.B) return;
    const uint32_t i32 = (uint32_t)i, node = i32 / g.B, b = i32 - node * g.B;          // N * B < 2^32 (checked by the build)
    const uint32_t v = as_global(g.vid)[node];
    const unsigned long long cb = as_global(g.compat)[b];
    uint32_t cnt = 0;
    if ((cb >> v) & 1ull) {
        const size_t row = (size_t)as_global(g.mask_idx)[node] * g.B + b;
        const uint32_t o0 = as_global(g.robs_off)[row], o1 = as_global(g.robs_off)[row + 1];
        for (uint32_t k = o0; k < o1; ++k) cnt += (uint32_t)((as_global(g.compat)[as_global(g.robs_par)[k]] >> v) & 1ull);
        const unsigned long long a0 = as_global(g.radj_off)[node], a1 = as_global(g.radj_off)[node + 1];
#pragma unroll 4
        for (unsigned long long k = a0; k < a1; ++k) {
            const uint32_t p = as_global(g.radj_id)[k];
            const uint32_t ev = as_global(g.radj_val)[k], pv = as_global(g.vid)[p];
            const size_t bit = (size_t)p * g.B + b;
            const unsigned long long word = as_global(g.obs_bits)[bit >> 6];
            cnt += (uint32_t)(((cb >> pv) & 1ull) & ((cb >> ev) & 1ull) & (~(word >> (bit & 63)) & 1ull));
        }
    }
    as_global(g.deg)[i] = cnt;
}

// The same two count passes for many beliefs per node (B >= 256): a workgroup takes 256 beliefs of ONE graph node,
// stages the node's neighbour list in LDS (id and the two validity bits an edge needs, 256 neighbours at a time) and
// every thread walks the staged list for its belief -- the per-neighbour loads that all threads shared become one.
template <bool PARENTS>
__global__ __launch_bounds__(256) void k_bg_count_tiled(BgConst g, uint32_t chunks) {
    __shared__ uint32_t s_id[256];
    __shared__ unsigned long long s_req[256];
    const uint32_t node = blockIdx.x / chunks, b = (blockIdx.x % chunks) * 256u + threadIdx.x;
    const bool valid = b < g.B;
    const uint32_t v = as_global(g.vid)[node];
    const unsigned long long cb = valid ? as_global(g.compat)[b] : 0ull;
    const bool rowok = valid && ((cb >> v) & 1ull);
    const size_t i = (size_t)node * g.B + b;
    uint32_t cnt = 0;
    if (rowok) {
        const size_t row = (size_t)as_global(g.mask_idx)[node] * g.B + b;
        const uint32_t *off = PARENTS ? g.robs_off : g.obs_off, *lst = PARENTS ? g.robs_par : g.obs_child;
        const uint32_t o0 = as_global(off)[row], o1 = as_global(off)[row + 1];
        for (uint32_t k = o0; k < o1; ++k) cnt += (uint32_t)((as_global(g.compat)[as_global(lst)[k]] >> v) & 1ull);
    }
    const bool is_obs = !PARENTS && cnt > 0;
    const unsigned long long a0 = as_global(PARENTS ? g.radj_off : g.adj_off)[node], a1 = as_global(PARENTS ? g.radj_off : g.adj_off)[node + 1];
    uint32_t acnt = 0;
    if (!PARENTS && g.n_validities == 1) {
        if (rowok && !is_obs) acnt = (uint32_t)(a1 - a0);   // one validity: every edge of a compatible row passes
    } else {
        for (unsigned long long t0 = a0; t0 < a1; t0 += 256) {
            __syncthreads();
            if (t0 + threadIdx.x < a1) {
                const uint32_t id = as_global(PARENTS ? g.radj_id : g.adj_id)[t0 + threadIdx.x];
                const uint32_t ev = as_global(PARENTS ? g.radj_val : g.adj_val)[t0 + threadIdx.x];
                s_id[threadIdx.x] = id;
                s_req[threadIdx.x] = (1ull << as_global(g.vid)[id]) | (1ull << ev);
            }
            __syncthreads();
            const uint32_t n = (uint32_t)(a1 - t0 < 256 ? a1 - t0 : 256);
            if (rowok && !is_obs) {
#pragma unroll 4
                for (uint32_t j = 0; j < n; ++j) {
                    const unsigned long long req = s_req[j];
                    uint32_t ok = (cb & req) == req;
                    if (PARENTS) {
                        const size_t bit = (size_t)s_id[j] * g.B + b;
                        ok &= (uint32_t)(~(as_global(g.obs_bits)[bit >> 6] >> (bit & 63)) & 1ull);
                    }
                    acnt += ok;
                }
            }
        }
    }
    if (PARENTS) {
        if (valid) as_global(g.deg)[i] = cnt + acnt;
    } else {
        if (valid) {
            as_global(g.types)[i] = is_obs ? BG_OBSERVATION : (acnt ? BG_ACTION : BG_UNKNOWN);
            as_global(g.deg)[i] = is_obs ? cnt : acnt;
        }
        mark_obs(g, i, node, b, is_obs);
    }
}

// Fill pass.  A wave owns 64 consecutive belief nodes (rows) and walks the concatenation of their SOURCE lists, one
// source element per lane: for the children of an observation node the posterior table row, for those of an action
// node the PTO adjacency of its graph node; for parents the reverse table row followed by the neighbours by ascending
// id.  Each lane evaluates the edge condition of its element; a ballot turns the survivors into positions (rank
// inside the row = survivors of the same row in lower lanes + what the row carried over from the previous 64
// elements), so what a wave stores per step is one contiguous run of the output array: the lists are written at
// streaming rate however ragged the rows are.
struct BgRow {
    unsigned long long out;       // first output slot of the row
    unsigned long long adj0;      // first adjacency element (action part)
    unsigned long long cb;        // compatibility bits of the row's belief
    uint32_t obs0, n_obs;         // table part: [obs0, obs0 + n_obs)
    uint32_t node, b;
    uint32_t v, pad;
};
constexpr uint32_t kFillWaves = 4;

template <bool PARENTS>
__global__ __launch_bounds__(64 * kFillWaves) void k_bg_fill(BgConst g) {
    __shared__ BgRow rows[kFillWaves][64];
    __shared__ uint32_t soff[kFillWaves][65];
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const size_t NB = (size_t)g.N * g.B;
    const size_t i = ((size_t)blockIdx.x * kFillWaves + w) * 64 + lane;
    // ---- this lane's row: where its source list lives and how long it is
    uint32_t n_src = 0;
    BgRow r{};
    if (i < NB) {
        r.node = (uint32_t)i / g.B; r.b = (uint32_t)i - r.node * g.B;              // N * B < 2^32 (checked by the build)
        r.v = as_global(g.vid)[r.node];
        r.cb = as_global(g.compat)[r.b];
        r.out = as_global(PARENTS ? g.par_off : g.child_off)[i];
        if ((r.cb >> r.v) & 1ull) {
            const size_t trow = (size_t)as_global(g.mask_idx)[r.node] * g.B + r.b;
            if (PARENTS) {
                r.obs0 = as_global(g.robs_off)[trow];
                r.n_obs = as_global(g.robs_off)[trow + 1] - r.obs0;
                r.adj0 = as_global(g.radj_off)[r.node];
                n_src = r.n_obs + (uint32_t)(as_global(g.radj_off)[r.node + 1] - r.adj0);
            } else {
                const uint8_t type = as_global(g.types)[i];
                if (type == BG_OBSERVATION) {
                    r.obs0 = as_global(g.obs_off)[trow];
                    r.n_obs = as_global(g.obs_off)[trow + 1] - r.obs0;
                    n_src = r.n_obs;
                } else if (type == BG_ACTION) {
                    r.adj0 = as_global(g.adj_off)[r.node];
                    n_src = (uint32_t)(as_global(g.adj_off)[r.node + 1] - r.adj0);
                }
            }
        }
    }
    rows[w][lane] = r;
    uint32_t inc = n_src;                                  // inclusive scan over the wave's rows
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    soff[w][lane + 1] = inc;
    if (lane == 0) soff[w][0] = 0;
    const uint32_t total = __shfl(inc, 63, 64);
    __syncthreads();                                       // (each wave only reads its own rows; every thread reaches this)
    uint32_t carry_row = 0xFFFFFFFFu, carry_cnt = 0;
    constexpr uint32_t U = 4;                               // steps in flight: their loads are independent, only the ranks chain
    // Common case, served without the search: the 64 rows are beliefs of ONE graph node and all of them walk its whole
    // adjacency (no table part) -- row and element index are a division by the common length, node data are scalars.
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(n_src), node0 = __builtin_amdgcn_readfirstlane(r.node);
    const bool same = __ballot(i < NB && n_src == d0 && r.node == node0 && r.n_obs == 0) == ~0ull;
    if (same && d0 > 0 && d0 < (1u << 17)) {
        const unsigned long long adj0 = __builtin_amdgcn_readfirstlane((uint32_t)r.adj0) |
                                        ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(r.adj0 >> 32)) << 32);
        const uint32_t b0 = __builtin_amdgcn_readfirstlane(r.b);
        const float inv_d = 1.0f / (float)d0;
        const bool one_validity = g.n_validities == 1;      // then every belief is compatible with every node and edge
        for (uint32_t q0 = 0; q0 < total; q0 += 64 * U) {
            uint32_t row[U], k[U], c[U], ev[U];
            bool live[U], pass[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t q = q0 + u * 64 + lane;
                live[u] = q < total;
                const uint32_t qq = live[u] ? q : total - 1;
                uint32_t rr = (uint32_t)((float)qq * inv_d);          // qq < 2^23: off by one at most
                if (rr * d0 > qq) --rr;
                else if ((rr + 1) * d0 <= qq) ++rr;
                row[u] = rr;
                k[u] = qq - rr * d0;
                c[u] = as_global(PARENTS ? g.radj_id : g.adj_id)[adj0 + k[u]];
                ev[u] = one_validity ? 0u : (uint32_t)as_global(PARENTS ? g.radj_val : g.adj_val)[adj0 + k[u]];
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                bool ok = live[u];
                if (!one_validity) {
                    const unsigned long long cb = rows[w][row[u]].cb;
                    const uint32_t cv = as_global(g.vid)[c[u]];
                    ok = ok && ((cb >> cv) & 1ull) && ((cb >> ev[u]) & 1ull);
                }
                if (PARENTS) {
                    const size_t bit = (size_t)(b0 + row[u]) * g.N + c[u];
                    ok = ok && !((as_global(g.obs_bits_t)[bit >> 6] >> (bit & 63)) & 1ull);
                }
                pass[u] = ok;
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                if (q0 + u * 64 >= total) break;            // wave-uniform
                const unsigned long long ballot = __ballot(pass[u]);
                const uint32_t seg = k[u] < lane ? lane - k[u] : 0u;
                const unsigned long long below = (1ull << lane) - 1ull, before_seg = (1ull << seg) - 1ull;
                uint32_t rank = (uint32_t)__popcll(ballot & below & ~before_seg);
                if (row[u] == carry_row) rank += carry_cnt;
                if (pass[u]) as_global(PARENTS ? g.par_id : g.child_id)[rows[w][row[u]].out + rank] = c[u] * g.B + (b0 + row[u]);
                carry_row = __shfl(row[u], 63, 64);
                carry_cnt = __shfl(rank + (pass[u] ? 1u : 0u), 63, 64);
            }
        }
        return;
    }
    for (uint32_t q0 = 0; q0 < total; q0 += 64 * U) {
        uint32_t row[U], k[U], value[U];
        bool live[U], pass[U];
        unsigned long long out0[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const uint32_t q = q0 + u * 64 + lane;
            live[u] = q < total;
            const uint32_t qq = live[u] ? q : total - 1;
            uint32_t lo = 0, hi = 64;                       // last row whose first source index is <= q (empty rows are skipped by the "<=")
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (soff[w][mid] <= qq) lo = mid; else hi = mid;
            }
            row[u] = lo;
            k[u] = qq - soff[w][lo];
        }
        uint32_t c[U], ev[U];
        bool table[U];
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const BgRow &R = rows[w][row[u]];
            table[u] = k[u] < R.n_obs;
            out0[u] = R.out;
            c[u] = 0; ev[u] = 0;
            if (live[u]) {
                if (table[u]) c[u] = as_global(PARENTS ? g.robs_par : g.obs_child)[R.obs0 + k[u]];
                else {
                    const unsigned long long e = R.adj0 + (k[u] - R.n_obs);
                    c[u] = as_global(PARENTS ? g.radj_id : g.adj_id)[e];
                    ev[u] = as_global(PARENTS ? g.radj_val : g.adj_val)[e];
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            const BgRow &R = rows[w][row[u]];
            pass[u] = false;
            value[u] = 0;
            if (live[u]) {
                if (table[u]) {
                    pass[u] = (as_global(g.compat)[c[u]] >> R.v) & 1ull;
                    value[u] = R.node * g.B + c[u];
                } else {
                    const uint32_t cv = as_global(g.vid)[c[u]];
                    bool ok = ((R.cb >> cv) & 1ull) && ((R.cb >> ev[u]) & 1ull);
                    if (PARENTS) {
                        const size_t bit = (size_t)R.b * g.N + c[u];
                        ok = ok && !((as_global(g.obs_bits_t)[bit >> 6] >> (bit & 63)) & 1ull);
                    }
                    pass[u] = ok;
                    value[u] = c[u] * g.B + R.b;
                }
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            if (q0 + u * 64 >= total) break;                // wave-uniform
            const unsigned long long ballot = __ballot(pass[u]);
            const uint32_t seg = k[u] < lane ? lane - k[u] : 0u;     // first lane of this row in this step
            const unsigned long long below = (1ull << lane) - 1ull, before_seg = (1ull << seg) - 1ull;
            uint32_t rank = (uint32_t)__popcll(ballot & below & ~before_seg);
            if (row[u] == carry_row) rank += carry_cnt;
            if (pass[u]) as_global(PARENTS ? g.par_id : g.child_id)[out0[u] + rank] = value[u];
            carry_row = __shfl(row[u], 63, 64);
            carry_cnt = __shfl(rank + (pass[u] ? 1u : 0u), 63, 64);
        }
    }
}

// ---- exclusive scan of deg[n] (u32) into off[n+1] (u64): block totals, one block over the totals, apply
constexpr uint32_t kScanItems = 16, kScanBlock = 256, kScanTile = kScanItems * kScanBlock;

__device__ __forceinline__ unsigned long long block_exclusive(unsigned long long v, unsigned long long *lds, unsigned long long &total) {
    // 256 threads: wave scan by shuffles, then the 4 wave totals through LDS
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    unsigned long long inc = v;
    for (uint32_t d = 1; d < 64; d <<= 1) {
        unsigned long long o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    unsigned long long base = 0;
    for (uint32_t k = 0; k < w; ++k) base += lds[k];
    total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_totals(const uint32_t *__restrict__ deg, size_t n, unsigned long long *__restrict__ tot) {
    __shared__ unsigned long long lds[4];
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
    unsigned long long s = 0;
    for (uint32_t k = 0; k < kScanItems; ++k)
        if (base + k < n) s += as_global(deg)[base + k];
    unsigned long long total;
    (void)block_exclusive(s, lds, total);
    if (threadIdx.x == 0) tot[blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_top(unsigned long long *__restrict__ tot, size_t nblk) {
    __shared__ unsigned long long lds[4];
    unsigned long long carry = 0;
    for (size_t c0 = 0; c0 < nblk; c0 += kScanBlock) {
        const size_t i = c0 + threadIdx.x;
        const unsigned long long v = i < nblk ? tot[i] : 0;
        unsigned long long total;
        const unsigned long long ex = block_exclusive(v, lds, total);
        if (i < nblk) tot[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) tot[nblk] = carry;
}

__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const uint32_t *__restrict__ deg, size_t n, const unsigned long long *__restrict__ tot,
                                                           unsigned long long *__restrict__ off, size_t nblk) {
    __shared__ unsigned long long lds[4];
    const size_t base = (size_t)blockIdx.x * kScanTile + (size_t)threadIdx.x * kScanItems;
    uint32_t d[kScanItems];
    unsigned long long s = 0;
    for (uint32_t k = 0; k < kScanItems; ++k) {
        d[k] = base + k < n ? as_global(deg)[base + k] : 0u;
        s += d[k];
    }
    unsigned long long total;
    unsigned long long run = tot[blockIdx.x] + block_exclusive(s, lds, total);
    for (uint32_t k = 0; k < kScanItems; ++k) {
        if (base + k < n) as_global(off)[base + k] = run;
        run += d[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) off[n] = tot[nblk];
}

// --------------------------------------------------------------------------------------------- host side

// Belief states of one problem: the reachable set, their hashes and the successor rule of the domain.
struct BeliefSpace {
    int domain = 0, nz = 0;
    uint32_t nw = 0;
    const uint64_t *validities = nullptr;
    std::vector<double> vec;                          // [B][nw]
    std::vector<uint64_t> hash;                       // [B]
    std::unordered_map<uint64_t, uint32_t> by_hash;
    size_t size() const { return hash.size(); }
    const double *at(size_t b) const { return vec.data() + b * nw; }

    // common.rs:352-355 (usize arithmetic; a release build wraps)
    static uint64_t hash_of(const double *p, uint32_t n) {
        uint64_t h = 0, p10 = 1;
        for (uint32_t i = 0; i < n; ++i, p10 *= 10) {
            const double x = p[i] * 1000.0;
            if (!(x > 0.0)) continue;                        // zero, negative and NaN all cast to 0
            uint64_t q;
            if (x >= 18446744073709551615.0) q = ~0ull;      // `as usize` saturates
            else {
                q = (uint64_t)x;                             // f64::round = half away from zero: trunc, then the exact remainder
                if (x - (double)q >= 0.5) ++q;
            }
            h += (p10 + 1) * q;
        }
        return h;
    }
    // get_successor_belief_states (map_io.rs:244-279 doors: zone closed, zone open; map_shelves_io.rs:206-240
    // shelves: object there, object not there); a posterior that normalises to NaN is dropped
    void successors(const double *b, int zone, std::vector<double> &out) const {
        for (int pass = 0; pass < 2; ++pass) {
            const size_t at0 = out.size();
            out.resize(at0 + nw);
            double *o = out.data() + at0;
            double sum = 0.0;
            for (uint32_t w = 0; w < nw; ++w) {
                bool keep;
                if (domain == PORRT_DOMAIN_DOOR) keep = (((validities[zone] >> w) & 1ull) != 0) == (pass == 1);
                else keep = (w == (uint32_t)zone) == (pass == 0);
                o[w] = keep ? b[w] : 0.0;
                sum = sum + o[w];
            }
            bool nan = sum != sum || sum == 0.0;             // x / 0 with x == 0 somewhere, or NaN inputs: a NaN posterior, dropped
            if (!nan)
                for (uint32_t w = 0; w < nw; ++w) {
                    o[w] /= sum;
                    nan |= o[w] != o[w];
                }
            if (nan) out.resize(at0);
        }
    }
    // Vec::contains over the reachable list: exact equality of vectors, as an open-addressing table of list indices
    struct ExactSet {
        const BeliefSpace *bs;
        std::vector<uint32_t> slot;                       // index + 1, 0 = empty
        size_t used = 0;
        static uint64_t mix(const double *p, uint32_t n) {
            uint64_t h = 1469598103934665603ull;
            for (uint32_t w = 0; w < n; ++w) {
                const double v = p[w] == 0.0 ? 0.0 : p[w];  // -0.0 == 0.0
                uint64_t u;
                std::memcpy(&u, &v, 8);
                h = (h ^ u) * 1099511628211ull;
                h ^= h >> 29;
            }
            return h;
        }
        bool equal(uint32_t id, const double *p) const {
            const double *q = bs->at(id);
            for (uint32_t w = 0; w < bs->nw; ++w) if (!(q[w] == p[w])) return false;
            return true;
        }
        bool contains(const double *p) const {
            const size_t m = slot.size() - 1;
            for (size_t k = mix(p, bs->nw) & m;; k = (k + 1) & m) {
                if (!slot[k]) return false;
                if (equal(slot[k] - 1, p)) return true;
            }
        }
        void insert(uint32_t id) {
            if (2 * (used + 1) > slot.size()) {
                std::vector<uint32_t> old;
                old.swap(slot);
                slot.assign(std::max<size_t>(64, 2 * old.size()), 0);
                used = 0;
                for (uint32_t v : old) if (v) insert(v - 1);
            }
            const size_t m = slot.size() - 1;
            size_t k = mix(bs->at(id), bs->nw) & m;
            while (slot[k]) k = (k + 1) & m;
            slot[k] = id + 1;
            ++used;
        }
    };
    // reachable_belief_states (map_io.rs:515-546): depth-first over (belief, zones still to check); a successor is new
    // when neither its exact vector nor its hash is known; note that the start's hash is never entered in the set
    int reach_from(const double *start, std::string &err) {
        vec.assign(start, start + nw);
        ExactSet exact{this};
        exact.slot.assign(64, 0);
        exact.insert(0);
        std::unordered_set<uint64_t> hashes;
        std::vector<double> pool(start, start + nw);      // the LIFO's belief vectors, one after the other
        std::vector<uint64_t> zones_of{nz >= 64 ? ~0ull : ((1ull << nz) - 1)};
        std::vector<double> succ, cur(nw);
        while (!zones_of.empty()) {
            const uint64_t zones = zones_of.back();
            std::memcpy(cur.data(), pool.data() + pool.size() - nw, nw * sizeof(double));
            zones_of.pop_back();
            pool.resize(pool.size() - nw);
            for (int z = 0; z < nz; ++z) {
                if (!((zones >> z) & 1ull)) continue;
                succ.clear();
                successors(cur.data(), z, succ);
                for (size_t s = 0; s * nw < succ.size(); ++s) {
                    const double *v = succ.data() + s * nw;
                    if (exact.contains(v)) continue;
                    const uint64_t h = hash_of(v, nw);
                    if (hashes.insert(h).second) {
                        vec.insert(vec.end(), v, v + nw);
                        exact.insert((uint32_t)(vec.size() / nw - 1));
                    }
                    pool.insert(pool.end(), v, v + nw);
                    zones_of.push_back(zones & ~(1ull << z));
                }
            }
        }
        const size_t B = vec.size() / nw;
        hash.resize(B);
        by_hash.clear();
        by_hash.reserve(2 * B);
        for (size_t b = 0; b < B; ++b) {
            hash[b] = hash_of(at(b), nw);
            if (!by_hash.emplace(hash[b], (uint32_t)b).second) { err = "collision when hashing the belief states! (belief_graph.rs:82)"; return PORRT_ERR_INVALID; }
        }
        return PORRT_OK;
    }
    // observe_impl for a node that sees exactly the zones of `mask`: posterior belief ids, in the reference's order,
    // without those that hash like the prior (pto.rs:217)
    int posteriors(uint32_t b, uint64_t mask, std::vector<uint32_t> &out, std::string &err, std::vector<double> &cur, std::vector<double> &nxt) const {
        cur.assign(at(b), at(b) + nw);
        for (int z = 0; z < nz; ++z) {
            if (!((mask >> z) & 1ull)) continue;
            nxt.clear();
            for (size_t s = 0; s * nw < cur.size(); ++s) successors(cur.data() + s * nw, z, nxt);
            cur.swap(nxt);
            if (cur.size() / nw > (1u << 20)) { err = "observe: more than 2^20 posteriors"; return PORRT_ERR_INVALID; }
        }
        for (size_t s = 0; s * nw < cur.size(); ++s) {
            const uint64_t h = hash_of(cur.data() + s * nw, nw);
            if (h == hash[b]) continue;
            auto f = by_hash.find(h);
            if (f == by_hash.end()) { err = "no id corresponding to this belief state! (belief_graph.rs:68)"; return PORRT_ERR_INVALID; }
            out.push_back(f->second);
        }
        return PORRT_OK;
    }
};

// What depends on the problem (domain, worlds, prior) but not on the graph: the reachable beliefs and, per set of
// visible zones met so far, the posterior table.  Kept across builds of one context: replanning on a new graph
// with the same prior pays for neither again.
struct BeliefCache {
    int domain = -1, nz = 0, nv = 0;
    uint32_t nw = 0;
    std::vector<uint64_t> validities;
    std::vector<double> start;
    BeliefSpace space;
    std::vector<unsigned long long> compat;
    struct Fold { std::vector<uint32_t> cnt, child; };        // per belief: number of posteriors; their ids one after the other
    std::map<uint64_t, Fold> folds;
    bool matches(int d, int z, int v, uint32_t w, const uint64_t *val, const double *st) const {
        if (d != domain || z != nz || v != nv || w != nw) return false;
        for (int k = 0; k < v; ++k) if (validities[k] != val[k]) return false;
        return std::memcmp(start.data(), st, w * sizeof(double)) == 0;
    }
};

// Result of the last build; owns its device memory.
struct BeliefGraphState {
    bool valid = false;
    BeliefCache cache;
    uint32_t nw = 0;
    size_t N = 0, B = 0;
    uint64_t n_edges = 0;
    std::vector<double> beliefs;
    std::vector<uint64_t> h_vis;                      // zones seen per node
    double t_total = 0, t_device = 0, t_tables = 0, t_reach = 0, t_post = 0, t_adj = 0, t_alloc = 0, t_edges = 0;
    // device buffers: a build asks for them in a fixed order; each slot keeps its allocation for the next build and
    // only grows (hipFree / hipMalloc of GB-sized lists cost more than filling them)
    struct Slot { void *p = nullptr; size_t bytes = 0; };
    std::vector<Slot> slots;
    size_t next_slot = 0;
    uint8_t *d_types = nullptr;
    unsigned long long *d_child_off = nullptr, *d_par_off = nullptr;
    uint32_t *d_child_id = nullptr, *d_par_id = nullptr;
    const double *d_beliefs = nullptr;                // [B][nw]
    BgConst last{};                                   // device pointers of the last build (tables, adjacency, bit planes)
    bool support_shrinks = false;                     // every posterior has fewer possible worlds than its prior (expected; checked)
    std::vector<uint32_t> support;                    // per belief: number of worlds with p > 0
    void release() {                                   // the result is gone, the memory stays for the next build
        next_slot = 0;
        valid = false;
        d_types = nullptr; d_child_off = d_par_off = nullptr; d_child_id = d_par_id = nullptr;
    }
    void free_device() {
        release();
        for (Slot &sl : slots) if (sl.p) (void)hipFree(sl.p);
        slots.clear();
    }
    ~BeliefGraphState() { free_device(); }
};

struct BeliefInputs {
    int domain, n_zones, n_worlds, n_validities;
    const uint64_t *validities;
    const double (*zone_pos)[2];
    double visibility;
    const RunConst *d_rc;
    size_t N, E;
    const double *d_nx, *d_ny;
    const uint8_t *d_vid, *h_vid;
    // PTO adjacency on the device (porrt_edges.hpp): children lists in push order, the same edges by ascending neighbour
    const unsigned long long *d_adj_off;
    const uint32_t *d_adj_id, *d_radj_id;
    const uint8_t *d_adj_val, *d_radj_val;
    uint64_t graph_tag;
    hipStream_t stream;
};

#define BG_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return PORRT_ERR_DEVICE; } \
    } while (0)

template <class T>
static int bg_alloc(BeliefGraphState &g, T *&p, size_t n, std::string &err) {
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (g.next_slot == g.slots.size()) g.slots.emplace_back();
    BeliefGraphState::Slot &sl = g.slots[g.next_slot++];
    if (sl.bytes < bytes) {
        if (sl.p) (void)hipFree(sl.p);
        sl.p = nullptr; sl.bytes = 0;
        const size_t want = bytes + bytes / 8;
        BG_HIP(hipMalloc(&sl.p, want));
        sl.bytes = want;
    }
    p = (T *)sl.p;
    return PORRT_OK;
}
template <class T>
static int bg_upload(BeliefGraphState &g, const T *&p, const std::vector<T> &h, hipStream_t s, std::string &err) {
    T *q = nullptr;
    int r = bg_alloc(g, q, h.size(), err);
    if (r) return r;
    if (!h.empty()) BG_HIP(hipMemcpyAsync(q, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
    p = q;
    return PORRT_OK;
}

static int bg_scan(const uint32_t *deg, size_t n, unsigned long long *tot, unsigned long long *off, hipStream_t s) {
    const size_t nblk = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(k_scan_totals, dim3((unsigned)nblk), dim3(kScanBlock), 0, s, deg, n, tot);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(kScanBlock), 0, s, tot, nblk);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nblk), dim3(kScanBlock), 0, s, deg, n, (const unsigned long long *)tot, off, nblk);
    return 0;
}

static double bg_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// PTO::build_belief_graph.  The PTO graph is the one of the context's last grow (mode PTO).
static int belief_graph_build(BeliefGraphState &g, const BeliefInputs &in, const double *start_belief, std::string &err) {
    g.release();
    const double t0 = bg_now();
    if (in.n_validities > 64) { err = "build_belief_graph: more than 64 world validities"; return PORRT_ERR_INVALID; }
    // assert_belief_state_validity (common.rs:279-281), checked by plan_belief_space before anything else (pto.rs:153)
    {
        double s = 0.0;
        for (int w = 0; w < in.n_worlds; ++w) s = start_belief[w] + s;
        if (!(std::fabs(s - 1.0) < 0.000001)) { err = "start belief state does not sum to 1"; return PORRT_ERR_INVALID; }
    }
    int r = PORRT_OK;
    BgConst c{};
    const size_t N = in.N;
    c.N = (uint32_t)N; c.B = 0; c.nz = (uint32_t)in.n_zones; c.n_validities = (uint32_t)in.n_validities;
    c.nx = in.d_nx; c.ny = in.d_ny; c.vid = in.d_vid; c.visibility = in.visibility;
    hipStream_t s = in.stream;
    ScopedEvents<4> evs;
    BG_HIP(evs.create());
    hipEvent_t ev0 = evs.e[0], ev1 = evs.e[1], ev2 = evs.e[2], ev3 = evs.e[3];

    // 1. visible zones per node
    std::vector<double> zxy(2 * std::max(in.n_zones, 1));
    for (int z = 0; z < in.n_zones; ++z) { zxy[2 * z] = in.zone_pos[z][0]; zxy[2 * z + 1] = in.zone_pos[z][1]; }
    if ((r = bg_upload(g, c.zone_xy, zxy, s, err))) return r;
    unsigned long long *d_vis = nullptr;
    uint32_t *d_err = nullptr;
    if ((r = bg_alloc(g, d_vis, N, err)) || (r = bg_alloc(g, d_err, 1, err))) return r;
    BG_HIP(hipMemsetAsync(d_vis, 0, N * sizeof(unsigned long long), s));
    BG_HIP(hipMemsetAsync(d_err, 0, sizeof(uint32_t), s));
    BG_HIP(hipEventRecord(ev0, s));
    if (in.n_zones > 0) {
        const size_t th = N * (size_t)in.n_zones;
        hipLaunchKernelGGL(k_bg_vismask, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, s, in.d_rc, c, d_vis, d_err);
    }
    BG_HIP(hipEventRecord(ev1, s));
    g.h_vis.assign(N, 0);
    uint32_t h_err = 0;
    BG_HIP(hipMemcpyAsync(g.h_vis.data(), d_vis, N * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    BG_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    // ... while the raycasts run: the belief states and the adjacency lists, which do not depend on them
    const double tr0 = bg_now();
    BeliefCache &bc = g.cache;
    if (!bc.matches(in.domain, in.n_zones, in.n_validities, (uint32_t)in.n_worlds, in.validities, start_belief)) {
        bc = BeliefCache();
        bc.domain = in.domain; bc.nz = in.n_zones; bc.nv = in.n_validities; bc.nw = (uint32_t)in.n_worlds;
        bc.validities.assign(in.validities, in.validities + in.n_validities);
        bc.start.assign(start_belief, start_belief + in.n_worlds);
        bc.space.domain = in.domain; bc.space.nz = in.n_zones; bc.space.nw = bc.nw; bc.space.validities = bc.validities.data();
        r = bc.space.reach_from(start_belief, err);
        if (r) { bc = BeliefCache(); (void)hipStreamSynchronize(s); return r; }          // the copies above target this frame
        bc.compat.assign(bc.space.size(), 0);                       // compute_compatibility (common.rs:266-276)
        for (size_t b = 0; b < bc.space.size(); ++b)
            for (int v = 0; v < in.n_validities; ++v) {
                bool ok = true;
                for (uint32_t w = 0; w < bc.nw && ok; ++w) ok = !(bc.space.at(b)[w] > 0.0) || ((in.validities[v] >> w) & 1ull);
                if (ok) bc.compat[b] |= 1ull << v;
            }
    }
    const BeliefSpace &bs = bc.space;
    const std::vector<unsigned long long> &compat = bc.compat;
    g.t_reach = bg_now() - tr0;
    const size_t B = bs.size();
    c.B = (uint32_t)B;
    if (N * B >= 0xFFFFFFFFull) { (void)hipStreamSynchronize(s); err = "build_belief_graph: more than 2^32 belief nodes"; return PORRT_ERR_INVALID; }
    g.nw = bs.nw; g.N = N; g.B = B;
    g.beliefs = bs.vec;

    const double tj0 = bg_now();
    g.t_adj = bg_now() - tj0;
    BG_HIP(hipStreamSynchronize(s));
    if (h_err) { err = "observe: raster access the reference would panic on (image::get_pixel / two zones on one ray, map_io.rs:233)"; return PORRT_ERR_RASTER; }

    // 2. host tables: distinct zone sets, the observation fold per (set, belief), PTO adjacency in both orders
    const double tt0 = bg_now();
    std::map<uint64_t, uint32_t> mask_id;
    std::vector<uint64_t> masks;
    std::vector<uint32_t> mask_idx(N);
    for (size_t i = 0; i < N; ++i) {
        auto f = mask_id.find(g.h_vis[i]);
        if (f == mask_id.end()) { f = mask_id.emplace(g.h_vis[i], (uint32_t)masks.size()).first; masks.push_back(g.h_vis[i]); }
        mask_idx[i] = f->second;
    }
    const size_t M = masks.size();
    c.n_masks = (uint32_t)M;
    std::vector<uint32_t> obs_off(M * B + 1, 0), obs_child, robs_off(M * B + 1, 0), robs_par;
    std::vector<double> obs_p;
    {
        // the fold of every (new zone set, belief) pair, pairs split evenly over a few host threads; each thread
        // appends to its own list, the lists are cut into the per-set tables in pair order
        std::vector<uint64_t> fresh;
        for (uint64_t m : masks) if (m && !bc.folds.count(m)) fresh.push_back(m);
        const size_t pairs = fresh.size() * B;
        if (pairs) {
            unsigned nt = std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
            if (pairs < 8192) nt = 1;
            std::vector<std::vector<uint32_t>> part(nt);
            std::vector<uint32_t> cnt(pairs, 0);
            std::vector<std::string> perr(nt);
            std::vector<int> prc(nt, PORRT_OK);
            auto work = [&](unsigned t) {
                std::vector<double> cur, nxt;
                const size_t p0 = pairs * t / nt, p1 = pairs * (t + 1) / nt;
                for (size_t pr = p0; pr < p1; ++pr) {
                    const size_t before = part[t].size();
                    if ((prc[t] = bs.posteriors((uint32_t)(pr % B), fresh[pr / B], part[t], perr[t], cur, nxt))) return;
                    cnt[pr] = (uint32_t)(part[t].size() - before);
                }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
            for (unsigned t = 0; t < nt; ++t) if (prc[t]) { err = perr[t]; return prc[t]; }
            std::vector<uint32_t> all;
            for (unsigned t = 0; t < nt; ++t) all.insert(all.end(), part[t].begin(), part[t].end());
            size_t at = 0;
            for (size_t f = 0; f < fresh.size(); ++f) {
                BeliefCache::Fold &fd = bc.folds[fresh[f]];
                fd.cnt.assign(cnt.begin() + f * B, cnt.begin() + (f + 1) * B);
                size_t n = 0;
                for (uint32_t v : fd.cnt) n += v;
                fd.child.assign(all.begin() + at, all.begin() + at + n);
                at += n;
            }
        }
        for (size_t m = 0; m < M; ++m) {
            if (!masks[m]) { for (size_t b = 0; b < B; ++b) obs_off[m * B + b + 1] = obs_off[m * B + b]; continue; }
            const BeliefCache::Fold &fd = bc.folds[masks[m]];
            for (size_t b = 0; b < B; ++b) obs_off[m * B + b + 1] = obs_off[m * B + b] + fd.cnt[b];
            obs_child.insert(obs_child.end(), fd.child.begin(), fd.child.end());
        }
        // transition probabilities of the table entries, and the order the dynamic programming may rely on
        obs_p.resize(obs_child.size());
        g.support.assign(B, 0);
        for (size_t b = 0; b < B; ++b)
            for (uint32_t w = 0; w < bs.nw; ++w) g.support[b] += bs.at(b)[w] > 0.0;
        g.support_shrinks = true;
        for (size_t m = 0; m < M; ++m)
            for (size_t b = 0; b < B; ++b)
                for (uint32_t k = obs_off[m * B + b]; k < obs_off[m * B + b + 1]; ++k) {
                    const double *pb = bs.at(b), *cbv = bs.at(obs_child[k]);
                    double pr = 0.0;
                    for (uint32_t w = 0; w < bs.nw; ++w) pr = pr + (cbv[w] > 0.0 ? pb[w] : 0.0);
                    obs_p[k] = pr;
                    if (g.support[obs_child[k]] >= g.support[b]) g.support_shrinks = false;
                }
        // reverse table: for every posterior its priors, ascending (counting sort per zone set)
        robs_par.resize(obs_child.size());
        std::vector<uint32_t> cur(B);
        for (size_t m = 0; m < M; ++m) {
            const uint32_t lo = obs_off[m * B], hi = obs_off[(m + 1) * B];
            std::fill(cur.begin(), cur.end(), 0u);
            for (uint32_t k = lo; k < hi; ++k) cur[obs_child[k]]++;
            uint32_t run = lo;
            for (size_t b = 0; b < B; ++b) { robs_off[m * B + b] = run; const uint32_t n = cur[b]; cur[b] = run; run += n; }
            robs_off[(m + 1) * B] = hi;
            for (size_t b = 0; b < B; ++b)
                for (uint32_t k = obs_off[m * B + b]; k < obs_off[m * B + b + 1]; ++k) robs_par[cur[obs_child[k]]++] = (uint32_t)b;
        }
    }
    g.t_post = bg_now() - tt0;
    g.t_tables = g.t_post + g.t_adj + g.t_reach;
    const double ta0 = bg_now();
    if ((r = bg_upload(g, g.d_beliefs, bs.vec, s, err))) return r;
    if ((r = bg_upload(g, c.compat, compat, s, err)) || (r = bg_upload(g, c.mask_idx, mask_idx, s, err)) ||
        (r = bg_upload(g, c.obs_off, obs_off, s, err)) || (r = bg_upload(g, c.obs_child, obs_child, s, err)) || (r = bg_upload(g, c.obs_p, obs_p, s, err)) ||
        (r = bg_upload(g, c.robs_off, robs_off, s, err)) || (r = bg_upload(g, c.robs_par, robs_par, s, err)) ||
        false)
        return r;
    c.adj_off = in.d_adj_off; c.adj_id = in.d_adj_id; c.adj_val = in.d_adj_val; c.radj_id = in.d_radj_id; c.radj_val = in.d_radj_val;
    c.radj_off = c.adj_off;

    // 3. children and parents lists
    const size_t NB = N * B;
    const size_t nblk = (NB + kScanTile - 1) / kScanTile;
    unsigned long long *d_tot = nullptr;
    if ((r = bg_alloc(g, c.types, NB, err)) || (r = bg_alloc(g, c.obs_bits, NB / 64 + 2, err)) || (r = bg_alloc(g, c.obs_bits_t, NB / 64 + 2, err)) || (r = bg_alloc(g, c.deg, NB, err)) || (r = bg_alloc(g, c.child_off, NB + 1, err)) ||
        (r = bg_alloc(g, c.par_off, NB + 1, err)) || (r = bg_alloc(g, d_tot, nblk + 1, err)))
        return r;
    const dim3 grid((unsigned)((NB + 255) / 256)), block(256);
    g.t_alloc = bg_now() - ta0;
    BG_HIP(hipEventRecord(ev2, s));
    BG_HIP(hipMemsetAsync(c.obs_bits, 0, (NB / 64 + 2) * sizeof(unsigned long long), s));
    BG_HIP(hipMemsetAsync(c.obs_bits_t, 0, (NB / 64 + 2) * sizeof(unsigned long long), s));
    const uint32_t chunks = (uint32_t)((B + 255) / 256);
    const bool tiled = B >= 256 && (size_t)N * chunks < 0x7FFFFFFFull;
    const dim3 tgrid((unsigned)(N * chunks));
    if (tiled) hipLaunchKernelGGL(k_bg_count_tiled<false>, tgrid, block, 0, s, c, chunks);
    else hipLaunchKernelGGL(k_bg_children_count, grid, block, 0, s, c);
    bg_scan(c.deg, NB, d_tot, c.child_off, s);
    unsigned long long n_edges = 0;
    BG_HIP(hipMemcpyAsync(&n_edges, c.child_off + NB, sizeof n_edges, hipMemcpyDeviceToHost, s));
    BG_HIP(hipStreamSynchronize(s));
    {
        const double tb0 = bg_now();
        if ((r = bg_alloc(g, c.child_id, n_edges, err)) || (r = bg_alloc(g, c.par_id, n_edges, err))) return r;
        g.t_alloc += bg_now() - tb0;
    }
    const dim3 fgrid((unsigned)((NB + 64 * kFillWaves - 1) / (64 * kFillWaves))), fblock(64 * kFillWaves);
    hipLaunchKernelGGL(k_bg_fill<false>, fgrid, fblock, 0, s, c);
    // (the staged variant of the parents count measured 3.6 ms against 1.0 ms for the plain one at 18M belief nodes: not used)
    hipLaunchKernelGGL(k_bg_parents_count, grid, block, 0, s, c);
    bg_scan(c.deg, NB, d_tot, c.par_off, s);
    hipLaunchKernelGGL(k_bg_fill<true>, fgrid, fblock, 0, s, c);
    BG_HIP(hipEventRecord(ev3, s));
    unsigned long long n_par = 0;
    BG_HIP(hipMemcpyAsync(&n_par, c.par_off + NB, sizeof n_par, hipMemcpyDeviceToHost, s));
    BG_HIP(hipStreamSynchronize(s));
    BG_HIP(hipGetLastError());
    if (n_par != n_edges) { err = "build_belief_graph: children and parents lists disagree"; return PORRT_ERR_DEVICE; }
    float ms_a = 0, ms_b = 0;
    BG_HIP(hipEventElapsedTime(&ms_a, ev0, ev1));
    BG_HIP(hipEventElapsedTime(&ms_b, ev2, ev3));
    g.n_edges = n_edges;
    g.last = c;
    g.d_types = c.types; g.d_child_off = c.child_off; g.d_par_off = c.par_off; g.d_child_id = c.child_id; g.d_par_id = c.par_id;
    g.t_device = 1e-3 * (double)(ms_a + ms_b);
    g.t_total = bg_now() - t0;
    g.valid = true;
    return PORRT_OK;
}

} // namespace porrt
