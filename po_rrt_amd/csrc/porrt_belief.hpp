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
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace porrt {

enum : uint8_t { BG_UNKNOWN = 0, BG_ACTION = 1, BG_OBSERVATION = 2 };     // belief_graph.rs:13-17

// what the kernels read; all pointers are device memory
struct BgConst {
    uint32_t N, B, nz, n_masks;
    const double *nx, *ny;            // node coordinates
    const uint8_t *vid;               // node validity id
    const double *zone_xy;            // [nz][2]
    double visibility;
    const unsigned long long *compat; // [B]: bit v = belief compatible with world validity v
    const uint32_t *mask_idx;         // [N]: index of the node's visible-zone set
    const uint32_t *obs_off, *obs_child;      // [(n_masks*B)+1], children beliefs of (mask, belief) in fold order
    const uint32_t *robs_off, *robs_par;      // reverse: parent beliefs of (mask, belief), ascending
    const unsigned long long *adj_off;        // [N+1] PTO adjacency in push order
    const uint32_t *adj_id;
    const uint8_t *adj_val;
    const unsigned long long *radj_off;       // [N+1] the same edges by ascending neighbour id
    const uint32_t *radj_id;
    const uint8_t *radj_val;
    uint8_t *types;                   // [N*B]
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

// Count pass, one thread per belief node (node, b): its type and the number of its children.
__global__ __launch_bounds__(256) void k_bg_children_count(BgConst g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)g.N * g.B) return;
    const uint32_t node = (uint32_t)(i / g.B), b = (uint32_t)(i % g.B);
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
            for (unsigned long long k = a0; k < a1; ++k) {
                const uint32_t ev = as_global(g.adj_val)[k], cv = as_global(g.vid)[as_global(g.adj_id)[k]];
                cnt += (uint32_t)(((cb >> cv) & 1ull) & ((cb >> ev) & 1ull));
            }
            if (cnt) type = BG_ACTION;
        }
    }
    as_global(g.types)[i] = type;
    as_global(g.deg)[i] = cnt;
}

// Count pass, one thread per belief node (node, b): how many point at it.  Needs the types of k_bg_children_count.
__global__ __launch_bounds__(256) void k_bg_parents_count(BgConst g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)g.N * g.B) return;
    const uint32_t node = (uint32_t)(i / g.B), b = (uint32_t)(i % g.B);
    const uint32_t v = as_global(g.vid)[node];
    const unsigned long long cb = as_global(g.compat)[b];
    uint32_t cnt = 0;
    if ((cb >> v) & 1ull) {
        const size_t row = (size_t)as_global(g.mask_idx)[node] * g.B + b;
        const uint32_t o0 = as_global(g.robs_off)[row], o1 = as_global(g.robs_off)[row + 1];
        for (uint32_t k = o0; k < o1; ++k) cnt += (uint32_t)((as_global(g.compat)[as_global(g.robs_par)[k]] >> v) & 1ull);
        const unsigned long long a0 = as_global(g.radj_off)[node], a1 = as_global(g.radj_off)[node + 1];
        for (unsigned long long k = a0; k < a1; ++k) {
            const uint32_t p = as_global(g.radj_id)[k];
            const uint32_t ev = as_global(g.radj_val)[k], pv = as_global(g.vid)[p];
            if (((cb >> pv) & 1ull) && ((cb >> ev) & 1ull) && as_global(g.types)[(size_t)p * g.B + b] != BG_OBSERVATION) ++cnt;
        }
    }
    as_global(g.deg)[i] = cnt;
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
        r.node = (uint32_t)(i / g.B); r.b = (uint32_t)(i % g.B);
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
    for (uint32_t q0 = 0; q0 < total; q0 += 64) {
        const uint32_t q = q0 + lane;
        const bool live = q < total;
        uint32_t row = 0;
        {   // last row whose first source index is <= q (rows with no sources are skipped by the "<=")
            uint32_t lo = 0, hi = 64;
            const uint32_t qq = live ? q : total - 1;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (soff[w][mid] <= qq) lo = mid; else hi = mid;
            }
            row = lo;
        }
        const BgRow &R = rows[w][row];
        const uint32_t k = (live ? q : total - 1) - soff[w][row];
        bool pass = false;
        uint32_t value = 0;
        if (live) {
            if (k < R.n_obs) {
                const uint32_t c = as_global(PARENTS ? g.robs_par : g.obs_child)[R.obs0 + k];
                pass = (as_global(g.compat)[c] >> R.v) & 1ull;
                value = R.node * g.B + c;
            } else {
                const unsigned long long e = R.adj0 + (k - R.n_obs);
                const uint32_t c = as_global(PARENTS ? g.radj_id : g.adj_id)[e];
                const uint32_t ev = as_global(PARENTS ? g.radj_val : g.adj_val)[e], cv = as_global(g.vid)[c];
                pass = ((R.cb >> cv) & 1ull) && ((R.cb >> ev) & 1ull);
                if (PARENTS && pass) pass = as_global(g.types)[(size_t)c * g.B + R.b] != BG_OBSERVATION;
                value = c * g.B + R.b;
            }
        }
        const unsigned long long ballot = __ballot(pass);
        const uint32_t seg = k < lane ? lane - k : 0u;     // first lane of this row in this step
        const unsigned long long below = (1ull << lane) - 1ull, before_seg = (1ull << seg) - 1ull;
        uint32_t rank = (uint32_t)__popcll(ballot & below & ~before_seg);
        if (row == carry_row) rank += carry_cnt;
        if (pass) as_global(PARENTS ? g.par_id : g.child_id)[R.out + rank] = value;
        carry_row = __shfl(row, 63, 64);
        carry_cnt = __shfl(rank + (pass ? 1u : 0u), 63, 64);
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
        for (uint32_t i = 0; i < n; ++i) {
            const double r = std::round(p[i] * 1000.0);
            uint64_t q = 0;
            if (r == r && r > 0.0) q = r >= 18446744073709551615.0 ? ~0ull : (uint64_t)r;
            h += (p10 + 1) * q;
            p10 *= 10;
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
            bool nan = false;
            for (uint32_t w = 0; w < nw; ++w) {
                o[w] /= sum;
                nan |= o[w] != o[w];
            }
            if (nan) out.resize(at0);
        }
    }
    static std::string key_of(const double *p, uint32_t n) {          // exact equality of vectors (Vec::contains)
        std::string k(n * sizeof(double), '\0');
        for (uint32_t w = 0; w < n; ++w) {
            const double v = p[w] == 0.0 ? 0.0 : p[w];
            std::memcpy(&k[w * sizeof(double)], &v, sizeof(double));
        }
        return k;
    }
    // reachable_belief_states (map_io.rs:515-546): depth-first over (belief, zones still to check); a successor is new
    // when neither its exact vector nor its hash is known; note that the start's hash is never entered in the set
    int reach_from(const double *start, std::string &err) {
        vec.assign(start, start + nw);
        std::unordered_set<std::string> exact{key_of(start, nw)};
        std::unordered_set<uint64_t> hashes;
        struct Item { std::vector<double> b; uint64_t zones; };
        std::vector<Item> lifo;
        lifo.push_back({std::vector<double>(start, start + nw), nz >= 64 ? ~0ull : ((1ull << nz) - 1)});
        std::vector<double> succ;
        while (!lifo.empty()) {
            Item it = std::move(lifo.back());
            lifo.pop_back();
            for (int z = 0; z < nz; ++z) {
                if (!((it.zones >> z) & 1ull)) continue;
                succ.clear();
                successors(it.b.data(), z, succ);
                for (size_t s = 0; s * nw < succ.size(); ++s) {
                    const double *v = succ.data() + s * nw;
                    std::string k = key_of(v, nw);
                    if (exact.count(k)) continue;
                    const uint64_t h = hash_of(v, nw);
                    if (!hashes.count(h)) {
                        vec.insert(vec.end(), v, v + nw);
                        exact.insert(std::move(k));
                        hashes.insert(h);
                    }
                    lifo.push_back({std::vector<double>(v, v + nw), it.zones & ~(1ull << z)});
                }
            }
        }
        const size_t B = vec.size() / nw;
        hash.resize(B);
        by_hash.clear();
        for (size_t b = 0; b < B; ++b) {
            hash[b] = hash_of(at(b), nw);
            if (!by_hash.emplace(hash[b], (uint32_t)b).second) { err = "collision when hashing the belief states! (belief_graph.rs:82)"; return PORRT_ERR_INVALID; }
        }
        return PORRT_OK;
    }
    // observe_impl for a node that sees exactly the zones of `mask`: posterior belief ids, in the reference's order,
    // without those that hash like the prior (pto.rs:217)
    int posteriors(uint32_t b, uint64_t mask, std::vector<uint32_t> &out, std::string &err) const {
        std::vector<double> cur(at(b), at(b) + nw), nxt;
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

// Result of the last build; owns its device memory.
struct BeliefGraphState {
    bool valid = false;
    uint32_t nw = 0;
    size_t N = 0, B = 0;
    uint64_t n_edges = 0;
    std::vector<double> beliefs;
    std::vector<uint64_t> h_vis;                      // zones seen per node
    double t_total = 0, t_device = 0, t_tables = 0;
    std::vector<void *> owned;
    uint8_t *d_types = nullptr;
    unsigned long long *d_child_off = nullptr, *d_par_off = nullptr;
    uint32_t *d_child_id = nullptr, *d_par_id = nullptr;
    void release() {
        for (void *p : owned) (void)hipFree(p);
        owned.clear();
        valid = false;
        d_types = nullptr; d_child_off = d_par_off = nullptr; d_child_id = d_par_id = nullptr;
    }
    ~BeliefGraphState() { release(); }
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
    const uint32_t *ef, *et, *ev;                     // forward edges, to ascending, from in kd pre-order (porrt_get_edges)
    hipStream_t stream;
};

#define BG_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return PORRT_ERR_DEVICE; } \
    } while (0)

template <class T>
static int bg_alloc(BeliefGraphState &g, T *&p, size_t n, std::string &err) {
    void *q = nullptr;
    BG_HIP(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    g.owned.push_back(q);
    p = (T *)q;
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
    BeliefSpace bs;
    bs.domain = in.domain; bs.nz = in.n_zones; bs.nw = (uint32_t)in.n_worlds; bs.validities = in.validities;
    int r = bs.reach_from(start_belief, err);
    if (r) return r;
    const size_t B = bs.size(), N = in.N;
    if (N * B >= 0xFFFFFFFFull) { err = "build_belief_graph: more than 2^32 belief nodes"; return PORRT_ERR_INVALID; }
    g.nw = bs.nw; g.N = N; g.B = B;
    g.beliefs = bs.vec;
    std::vector<unsigned long long> compat(B, 0);                   // compute_compatibility (common.rs:266-276)
    for (size_t b = 0; b < B; ++b)
        for (int v = 0; v < in.n_validities; ++v) {
            bool ok = true;
            for (uint32_t w = 0; w < bs.nw && ok; ++w) ok = !(bs.at(b)[w] > 0.0) || ((in.validities[v] >> w) & 1ull);
            if (ok) compat[b] |= 1ull << v;
        }

    BgConst c{};
    c.N = (uint32_t)N; c.B = (uint32_t)B; c.nz = (uint32_t)in.n_zones;
    c.nx = in.d_nx; c.ny = in.d_ny; c.vid = in.d_vid; c.visibility = in.visibility;
    hipStream_t s = in.stream;
    hipEvent_t ev0, ev1, ev2, ev3;
    BG_HIP(hipEventCreate(&ev0)); BG_HIP(hipEventCreate(&ev1)); BG_HIP(hipEventCreate(&ev2)); BG_HIP(hipEventCreate(&ev3));

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
    for (size_t m = 0; m < M; ++m) {
        std::vector<std::vector<uint32_t>> rev(B);
        for (size_t b = 0; b < B; ++b) {
            if (masks[m]) {
                if ((r = bs.posteriors((uint32_t)b, masks[m], obs_child, err))) return r;
            }
            obs_off[m * B + b + 1] = (uint32_t)obs_child.size();
            for (uint32_t k = obs_off[m * B + b]; k < obs_off[m * B + b + 1]; ++k) rev[obs_child[k]].push_back((uint32_t)b);
        }
        for (size_t b = 0; b < B; ++b) {
            robs_par.insert(robs_par.end(), rev[b].begin(), rev[b].end());
            robs_off[m * B + b + 1] = (uint32_t)robs_par.size();
        }
    }
    // PTOGraph::children in push order (pto.rs:111-120): per new node, first every add_edge(nbr, new), then every add_edge(new, nbr)
    std::vector<unsigned long long> adj_off(N + 1, 0);
    for (size_t e = 0; e < in.E; ++e) { adj_off[in.ef[e] + 1]++; adj_off[in.et[e] + 1]++; }
    for (size_t i = 0; i < N; ++i) adj_off[i + 1] += adj_off[i];
    std::vector<uint32_t> adj_id(2 * in.E), radj_id(2 * in.E);
    std::vector<uint8_t> adj_val(2 * in.E), radj_val(2 * in.E);
    {
        std::vector<unsigned long long> fill(adj_off.begin(), adj_off.end() - 1);
        for (size_t e = 0; e < in.E;) {
            size_t e1 = e;
            while (e1 < in.E && in.et[e1] == in.et[e]) ++e1;
            for (size_t k = e; k < e1; ++k) { const auto p = fill[in.ef[k]]++; adj_id[p] = in.et[k]; adj_val[p] = (uint8_t)in.ev[k]; }
            for (size_t k = e; k < e1; ++k) { const auto p = fill[in.et[k]]++; adj_id[p] = in.ef[k]; adj_val[p] = (uint8_t)in.ev[k]; }
            e = e1;
        }
        // the same lists by ascending neighbour id: the order in which the action-edge loop reaches a node's parents
        std::vector<std::pair<uint32_t, uint8_t>> tmp;
        for (size_t i = 0; i < N; ++i) {
            tmp.clear();
            for (auto k = adj_off[i]; k < adj_off[i + 1]; ++k) tmp.emplace_back(adj_id[k], adj_val[k]);
            std::stable_sort(tmp.begin(), tmp.end(), [](const auto &a, const auto &b2) { return a.first < b2.first; });
            for (size_t k = 0; k < tmp.size(); ++k) { radj_id[adj_off[i] + k] = tmp[k].first; radj_val[adj_off[i] + k] = tmp[k].second; }
        }
    }
    g.t_tables = bg_now() - tt0;
    if ((r = bg_upload(g, c.compat, compat, s, err)) || (r = bg_upload(g, c.mask_idx, mask_idx, s, err)) ||
        (r = bg_upload(g, c.obs_off, obs_off, s, err)) || (r = bg_upload(g, c.obs_child, obs_child, s, err)) ||
        (r = bg_upload(g, c.robs_off, robs_off, s, err)) || (r = bg_upload(g, c.robs_par, robs_par, s, err)) ||
        (r = bg_upload(g, c.adj_off, adj_off, s, err)) || (r = bg_upload(g, c.adj_id, adj_id, s, err)) ||
        (r = bg_upload(g, c.adj_val, adj_val, s, err)) || (r = bg_upload(g, c.radj_id, radj_id, s, err)) ||
        (r = bg_upload(g, c.radj_val, radj_val, s, err)))
        return r;
    c.radj_off = c.adj_off;

    // 3. children and parents lists
    const size_t NB = N * B;
    const size_t nblk = (NB + kScanTile - 1) / kScanTile;
    unsigned long long *d_tot = nullptr;
    if ((r = bg_alloc(g, c.types, NB, err)) || (r = bg_alloc(g, c.deg, NB, err)) || (r = bg_alloc(g, c.child_off, NB + 1, err)) ||
        (r = bg_alloc(g, c.par_off, NB + 1, err)) || (r = bg_alloc(g, d_tot, nblk + 1, err)))
        return r;
    const dim3 grid((unsigned)((NB + 255) / 256)), block(256);
    BG_HIP(hipEventRecord(ev2, s));
    hipLaunchKernelGGL(k_bg_children_count, grid, block, 0, s, c);
    bg_scan(c.deg, NB, d_tot, c.child_off, s);
    unsigned long long n_edges = 0;
    BG_HIP(hipMemcpyAsync(&n_edges, c.child_off + NB, sizeof n_edges, hipMemcpyDeviceToHost, s));
    BG_HIP(hipStreamSynchronize(s));
    if ((r = bg_alloc(g, c.child_id, n_edges, err)) || (r = bg_alloc(g, c.par_id, n_edges, err))) return r;
    const dim3 fgrid((unsigned)((NB + 64 * kFillWaves - 1) / (64 * kFillWaves))), fblock(64 * kFillWaves);
    hipLaunchKernelGGL(k_bg_fill<false>, fgrid, fblock, 0, s, c);
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
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1); (void)hipEventDestroy(ev2); (void)hipEventDestroy(ev3);
    g.n_edges = n_edges;
    g.d_types = c.types; g.d_child_off = c.child_off; g.d_par_off = c.par_off; g.d_child_id = c.child_id; g.d_par_id = c.par_id;
    g.t_device = 1e-3 * (double)(ms_a + ms_b);
    g.t_total = bg_now() - t0;
    g.valid = true;
    return PORRT_OK;
}

} // namespace porrt
