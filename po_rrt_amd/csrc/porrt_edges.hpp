// porrt_edges.hpp -- the adjacency order of a PTO graph / PRM roadmap, restored on the device.
//
// The growth kernels keep the edge SET (neighbour -> new node, appended in no particular order).  The reference's
// PTOGraph keeps adjacency LISTS: for a new node first add_edge(nbr, new) for its neighbours in the order
// KdTree::nearest_neighbors lists them -- kd pre-order (nearest_neighbor.rs:101-117) --, then add_edge(new, nbr)
// (pto.rs:103-120, prm.rs:96-103).  So the children (== parents) list of node X is
//     [neighbours at its creation, in kd pre-order]  ++  [later nodes that found X, in id order],
// and everything downstream (porrt_get_edges, the belief graph's lists, extract_path) inherits that order.
// The pre-order rank of every node comes from the host (a kd-tree of the coordinates built in id order: sequential
// pointer chasing, 2 ms for 18k nodes); the edges never leave the device: they are dealt into one bucket per node (degree counts,
// a scan, a scatter) and every bucket -- a node's neighbours, a few hundred at most -- is ordered by one wave (k_eo_segsort: each
// element counts the keys below its own): by the pre-order rank of the neighbour, by the neighbour's id, and the later nodes
// that found the node by their id.
#pragma once
#include "porrt_belief.hpp"

namespace porrt {

struct EdgeOrderState {
    uint64_t tag = ~0ull;             // results_tag of the graph these arrays belong to
    size_t N = 0, E = 0;
    // forward edges in the order porrt_get_edges returns them: new node ascending, neighbours in kd pre-order
    uint32_t *d_from = nullptr, *d_to = nullptr, *d_val = nullptr;
    // adjacency in push order and by ascending neighbour id (what the belief-graph kernels read)
    unsigned long long *d_adj_off = nullptr;
    uint32_t *d_adj_id = nullptr, *d_radj_id = nullptr;
    uint8_t *d_adj_val = nullptr, *d_radj_val = nullptr;
    struct Slot { void *p = nullptr; size_t bytes = 0; };
    std::vector<Slot> slots;
    size_t next_slot = 0;
    double t_total = 0;
    void free_device() {
        for (Slot &sl : slots) if (sl.p) (void)hipFree(sl.p);
        slots.clear();
        tag = ~0ull;
    }
    ~EdgeOrderState() { free_device(); }
};

#define EO_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return PORRT_ERR_DEVICE; } \
    } while (0)

template <class T>
static int eo_alloc(EdgeOrderState &st, T *&p, size_t n, std::string &err) {
    const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
    if (st.next_slot == st.slots.size()) st.slots.emplace_back();
    EdgeOrderState::Slot &sl = st.slots[st.next_slot++];
    if (sl.bytes < bytes) {
        if (sl.p) (void)hipFree(sl.p);
        sl.p = nullptr; sl.bytes = 0;
        EO_HIP(hipMalloc(&sl.p, bytes + bytes / 8));
        sl.bytes = bytes + bytes / 8;
    }
    p = (T *)sl.p;
    return PORRT_OK;
}

// Edges into buckets: bucket of node t (its creation neighbours: edges f -> t) and bucket of node f (the later nodes that found f),
// in arrival order -- with the keys the segment sorts below order them by.  Bucket offsets are the scanned degrees.
__global__ __launch_bounds__(256) void k_eo_degrees(const uint32_t *__restrict__ from, const uint32_t *__restrict__ to, size_t E, uint32_t *__restrict__ deg_to,
                                                    uint32_t *__restrict__ deg_from) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    atomicAdd(&deg_to[as_global(to)[e]], 1u);
    atomicAdd(&deg_from[as_global(from)[e]], 1u);
}
__global__ __launch_bounds__(256) void k_eo_scatter(const uint32_t *__restrict__ from, const uint32_t *__restrict__ to, const uint32_t *__restrict__ rank, size_t E,
                                                    const unsigned long long *__restrict__ off_to, const unsigned long long *__restrict__ off_from,
                                                    uint32_t *__restrict__ cur_to, uint32_t *__restrict__ cur_from, uint32_t *__restrict__ bt_e,
                                                    uint32_t *__restrict__ bt_rank, uint32_t *__restrict__ bt_from, uint32_t *__restrict__ bf_e,
                                                    uint32_t *__restrict__ bf_to) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint32_t f = as_global(from)[e], t = as_global(to)[e];
    const unsigned long long p = off_to[t] + atomicAdd(&cur_to[t], 1u), q = off_from[f] + atomicAdd(&cur_from[f], 1u);
    bt_e[p] = (uint32_t)e; bt_rank[p] = as_global(rank)[f]; bt_from[p] = f;
    bf_e[q] = (uint32_t)e; bf_to[q] = t;
}

// One wave per node: its bucket [off[x], off[x + 1]) ordered by key -- every element counts the keys below its own, which is its place.
// Keys are distinct inside a bucket for a graph this engine grew (an edge f -> t exists once, ranks are a permutation); equal keys
// (a repeated edge in a loaded graph) are ordered by position, so the order is total and stable and every slot is written once.  A node has at most a few
// hundred neighbours (the radius search's hits); buckets of up to kSegLds entries are staged in LDS, longer ones (the dense
// start of a belief-space graph) are counted against the keys in memory.
constexpr uint32_t kSegLds = 512;
__global__ __launch_bounds__(256) void k_eo_segsort(const unsigned long long *__restrict__ off, const uint32_t *__restrict__ key, const uint32_t *__restrict__ val,
                                                    uint32_t N, uint32_t *__restrict__ out) {
    __shared__ uint32_t s_key[4][kSegLds];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = blockIdx.x * 4u + wv;
    if (x >= N) return;
    const unsigned long long a = off[x];
    const uint32_t L = (uint32_t)(off[x + 1] - a);
    if (L == 0) return;
    auto gk = as_global(key) + a;
    auto gv = as_global(val) + a;
    if (L <= kSegLds) {
        for (uint32_t i = lane; i < L; i += 64u) s_key[wv][i] = gk[i];
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < L; i += 64u) {
            const uint32_t ki = s_key[wv][i];
            uint32_t below = 0;
            for (uint32_t j = 0; j < L; ++j) { const uint32_t kj = s_key[wv][j]; below += (kj < ki || (kj == ki && j < i)) ? 1u : 0u; }       // (all lanes read one word: a broadcast)
            out[a + below] = gv[i];
        }
    } else {
        for (uint32_t i = lane; i < L; i += 64u) {
            const uint32_t ki = gk[i];
            uint32_t below = 0;
            for (uint32_t j = 0; j < L; ++j) { const uint32_t kj = gk[j]; below += (kj < ki || (kj == ki && j < i)) ? 1u : 0u; }
            out[a + below] = gv[i];
        }
    }
}

// ordered forward edge list from sort 0
__global__ __launch_bounds__(256) void k_eo_gather(const uint32_t *__restrict__ from, const uint32_t *__restrict__ to, const uint32_t *__restrict__ val,
                                                   const uint32_t *__restrict__ order, size_t E, uint32_t *__restrict__ ofrom, uint32_t *__restrict__ oto,
                                                   uint32_t *__restrict__ oval) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= E) return;
    const uint32_t e = as_global(order)[k];
    ofrom[k] = as_global(from)[e]; oto[k] = as_global(to)[e]; oval[k] = as_global(val)[e];
}

// adjacency of node x: [creation neighbours in kd pre-order: ord0's run of x] ++ [later nodes ascending: ord1's run of x];
// radj (by ascending neighbour id): [creation neighbours by id: ordf's run of x -- all older than x] ++ [later nodes: ord1's run]
__global__ __launch_bounds__(256) void k_eo_adjacency(const uint32_t *__restrict__ from, const uint32_t *__restrict__ to, const uint32_t *__restrict__ val,
                                                      const uint32_t *__restrict__ ord0, const uint32_t *__restrict__ ord1, const uint32_t *__restrict__ ordf,
                                                      const unsigned long long *__restrict__ off_to, const unsigned long long *__restrict__ off_from,
                                                      uint32_t N, unsigned long long *__restrict__ adj_off, uint32_t *__restrict__ adj_id,
                                                      uint8_t *__restrict__ adj_val, uint32_t *__restrict__ radj_id, uint8_t *__restrict__ radj_val) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x > N) return;
    if (x == N) { adj_off[N] = off_to[N] + off_from[N]; return; }
    const unsigned long long a = off_to[x], b = off_to[x + 1], c = off_from[x], d = off_from[x + 1];
    const unsigned long long base = a + c;                       // both offset arrays are prefix sums over the same node order
    adj_off[x] = base;
    unsigned long long w = base;
    for (unsigned long long k = a; k < b; ++k, ++w) {
        const uint32_t e = ord0[k], e2 = ordf[k];
        adj_id[w] = from[e]; adj_val[w] = (uint8_t)val[e];
        radj_id[w] = from[e2]; radj_val[w] = (uint8_t)val[e2];
    }
    for (unsigned long long k = c; k < d; ++k, ++w) {
        const uint32_t e = ord1[k];
        adj_id[w] = to[e]; adj_val[w] = (uint8_t)val[e];
        radj_id[w] = to[e]; radj_val[w] = (uint8_t)val[e];
    }
}

// ---- the pre-order rank of every node in the reference's kd-tree of the coordinates (KdTree::add in id order, nearest_neighbor.rs:29-46;
// the order of :101-117), on the device.  The tree is built a level per round by the whole GPU -- every node not yet placed bids for
// the empty child slot its descent has reached (atomicMin of its id: sequential insertion gives a slot to the lowest id among the
// nodes whose paths reach it, and those all reach it in the same round), the next round places the winners and lets the others step
// below them -- then every node counts itself into its ancestors (subtree sizes) and walks to the root once more for its rank.
// (k_seg_kd_ranks, porrt_prm.hpp, is the same for many small trees, a workgroup each.)
constexpr int kKd1Empty = 0x7FFFFFFF;
constexpr uint32_t kKd1Placed = 0x80000000u;
__global__ __launch_bounds__(256) void k_kd1_init(uint32_t N, int *__restrict__ child, int *__restrict__ par, uint32_t *__restrict__ dep, uint32_t *__restrict__ size) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    child[2 * (size_t)t] = kKd1Empty; child[2 * (size_t)t + 1] = kKd1Empty;
    par[t] = t ? 0 : -1;                       // (until a node is placed: the node its descent stands at)
    dep[t] = t ? 0u : kKd1Placed;              // depth of that node; the root is placed
    size[t] = 1u;
}
__device__ __forceinline__ size_t kd1_slot(const double *x, const double *y, uint32_t t, int c, uint32_t d) {
    const bool left = (d & 1u) ? (as_global(y)[t] < as_global(y)[c]) : (as_global(x)[t] < as_global(x)[c]);      // strictly less goes left (nearest_neighbor.rs:33-35)
    return 2 * (size_t)c + (left ? 0 : 1);
}
// one round: a node that bid in the round before looks at its slot -- its own id: placed; another's: it steps below that node --, and
// every node still on its way bids for the slot below the node it stands at.  All such nodes stand at the same depth (they start at
// the root together and go down a level per round), so the slots read in a round (that depth's) are never the slots written in it (the
// next depth's): one kernel per round is enough.
constexpr uint32_t kKd1Bid = 0x40000000u;
__global__ __launch_bounds__(256) void k_kd1_round(const double *__restrict__ x, const double *__restrict__ y, uint32_t N, int *__restrict__ child,
                                                   int *__restrict__ par, uint32_t *__restrict__ dep, uint32_t *__restrict__ more) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    uint32_t d = as_global(dep)[t];
    if (d & kKd1Placed) return;
    int c = as_global(par)[t];
    if (d & kKd1Bid) {
        d &= ~kKd1Bid;
        const int w = as_global(child)[kd1_slot(x, y, t, c, d)];
        if (w == (int)t) { dep[t] = kKd1Placed | (d + 1u); return; }
        c = w; d += 1u;
        par[t] = c;
    }
    atomicMin(&child[kd1_slot(x, y, t, c, d)], (int)t);
    dep[t] = d | kKd1Bid;
    *more = 1u;
}
__global__ __launch_bounds__(256) void k_kd1_sizes(uint32_t N, const int *__restrict__ par, uint32_t *__restrict__ size) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    for (int a = as_global(par)[t]; a >= 0; a = as_global(par)[a]) atomicAdd(&size[a], 1u);
}
__global__ __launch_bounds__(256) void k_kd1_ranks(uint32_t N, const int *__restrict__ child, const int *__restrict__ par, const uint32_t *__restrict__ size,
                                                   uint32_t *__restrict__ rank) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    uint32_t r = 0;
    int v = (int)t;
    for (int a = as_global(par)[v]; a >= 0; v = a, a = as_global(par)[a]) {
        r += 1u;
        const int l = as_global(child)[2 * (size_t)a];
        if (l != v && l != kKd1Empty) r += as_global(size)[l];            // the path turns right at a: a's left subtree comes first
    }
    rank[t] = r;
}
// ranks of the N nodes at (d_x, d_y) into `rank` (device), scratch from the state's slots
static int kd_ranks_device(EdgeOrderState &st, size_t N, const double *d_x, const double *d_y, uint32_t *rank, hipStream_t s, std::string &err) {
    int *child, *par;
    uint32_t *dep, *size, *more;
    int r;
    constexpr uint32_t kGroup = 32;
    if ((r = eo_alloc(st, child, 2 * N, err)) || (r = eo_alloc(st, par, N, err)) || (r = eo_alloc(st, dep, N, err)) || (r = eo_alloc(st, size, N, err)) ||
        (r = eo_alloc(st, more, kGroup, err)))
        return r;
    const dim3 block(256), grid((unsigned)((std::max<size_t>(N, 1) + 255) / 256));
    hipLaunchKernelGGL(k_kd1_init, grid, block, 0, s, (uint32_t)N, child, par, dep, size);
    uint32_t h_more[kGroup];
    for (size_t rounds = 0; rounds <= N; rounds += kGroup) {
        EO_HIP(hipMemsetAsync(more, 0, kGroup * sizeof(uint32_t), s));
        for (uint32_t k = 0; k < kGroup; ++k)
            hipLaunchKernelGGL(k_kd1_round, grid, block, 0, s, d_x, d_y, (uint32_t)N, child, par, dep, more + k);
        EO_HIP(hipMemcpyAsync(h_more, more, sizeof h_more, hipMemcpyDeviceToHost, s));
        EO_HIP(hipStreamSynchronize(s));
        if (!h_more[kGroup - 1]) break;          // the group's last round placed the last node (or an earlier one did)
    }
    hipLaunchKernelGGL(k_kd1_sizes, grid, block, 0, s, (uint32_t)N, (const int *)par, size);
    hipLaunchKernelGGL(k_kd1_ranks, grid, block, 0, s, (uint32_t)N, (const int *)child, (const int *)par, (const uint32_t *)size, rank);
    return PORRT_OK;
}

// Orders the E edges (d_from, d_to, d_val: the growth kernels' arrays) of a graph of N nodes; the nodes' pre-order ranks come from
// h_rank (host) or, when that is null, are made on the device from the coordinates (d_x, d_y).
static int edge_order_build(EdgeOrderState &st, uint64_t tag, size_t N, size_t E, const uint32_t *d_from, const uint32_t *d_to, const uint32_t *d_val,
                            const std::vector<uint32_t> *h_rank, const double *d_x, const double *d_y, hipStream_t s, std::string &err) {
    const double t0 = bg_now();
    st.next_slot = 0;
    st.tag = ~0ull;
    st.N = N; st.E = E;
    uint32_t *rank, *deg_to, *deg_from, *cur_to, *cur_from, *bt_e, *bt_rank, *bt_from, *bf_e, *bf_to, *ord0, *ord1, *ordf;
    unsigned long long *off_to, *off_from, *tot;
    int r;
    const size_t nblk = (N + 1 + kScanTile - 1) / kScanTile;
    if ((r = eo_alloc(st, st.d_from, E, err)) || (r = eo_alloc(st, st.d_to, E, err)) || (r = eo_alloc(st, st.d_val, E, err)) ||
        (r = eo_alloc(st, st.d_adj_off, N + 1, err)) || (r = eo_alloc(st, st.d_adj_id, 2 * E, err)) || (r = eo_alloc(st, st.d_radj_id, 2 * E, err)) ||
        (r = eo_alloc(st, st.d_adj_val, 2 * E, err)) || (r = eo_alloc(st, st.d_radj_val, 2 * E, err)) ||
        (r = eo_alloc(st, rank, N, err)) || (r = eo_alloc(st, deg_to, N + 1, err)) || (r = eo_alloc(st, deg_from, N + 1, err)) ||
        (r = eo_alloc(st, cur_to, N + 1, err)) || (r = eo_alloc(st, cur_from, N + 1, err)) ||
        (r = eo_alloc(st, bt_e, E, err)) || (r = eo_alloc(st, bt_rank, E, err)) || (r = eo_alloc(st, bt_from, E, err)) || (r = eo_alloc(st, bf_e, E, err)) ||
        (r = eo_alloc(st, bf_to, E, err)) || (r = eo_alloc(st, ord0, E, err)) || (r = eo_alloc(st, ord1, E, err)) || (r = eo_alloc(st, ordf, E, err)) ||
        (r = eo_alloc(st, off_to, N + 2, err)) || (r = eo_alloc(st, off_from, N + 2, err)) || (r = eo_alloc(st, tot, nblk + 2, err)))
        return r;
    if (h_rank) EO_HIP(hipMemcpyAsync(rank, h_rank->data(), N * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    else if ((r = kd_ranks_device(st, N, d_x, d_y, rank, s, err))) return r;
    EO_HIP(hipMemsetAsync(deg_to, 0, (N + 1) * sizeof(uint32_t), s));
    EO_HIP(hipMemsetAsync(deg_from, 0, (N + 1) * sizeof(uint32_t), s));
    EO_HIP(hipMemsetAsync(cur_to, 0, (N + 1) * sizeof(uint32_t), s));
    EO_HIP(hipMemsetAsync(cur_from, 0, (N + 1) * sizeof(uint32_t), s));
    const dim3 block(256), egrid((unsigned)((std::max<size_t>(E, 1) + 255) / 256));
    if (E) hipLaunchKernelGGL(k_eo_degrees, egrid, block, 0, s, d_from, d_to, E, deg_to, deg_from);
    bg_scan(deg_to, N + 1, tot, off_to, s);                       // off[x] = edges whose new node (neighbour) is below x; off[N] = E
    bg_scan(deg_from, N + 1, tot, off_from, s);
    if (E) {
        hipLaunchKernelGGL(k_eo_scatter, egrid, block, 0, s, d_from, d_to, (const uint32_t *)rank, E, (const unsigned long long *)off_to,
                           (const unsigned long long *)off_from, cur_to, cur_from, bt_e, bt_rank, bt_from, bf_e, bf_to);
        const dim3 ngrid((unsigned)((N + 3) / 4));
        hipLaunchKernelGGL(k_eo_segsort, ngrid, block, 0, s, (const unsigned long long *)off_to, (const uint32_t *)bt_rank, (const uint32_t *)bt_e, (uint32_t)N, ord0);
        hipLaunchKernelGGL(k_eo_segsort, ngrid, block, 0, s, (const unsigned long long *)off_to, (const uint32_t *)bt_from, (const uint32_t *)bt_e, (uint32_t)N, ordf);
        hipLaunchKernelGGL(k_eo_segsort, ngrid, block, 0, s, (const unsigned long long *)off_from, (const uint32_t *)bf_to, (const uint32_t *)bf_e, (uint32_t)N, ord1);
        hipLaunchKernelGGL(k_eo_gather, egrid, block, 0, s, d_from, d_to, d_val, (const uint32_t *)ord0, E, st.d_from, st.d_to, st.d_val);
    }
    hipLaunchKernelGGL(k_eo_adjacency, dim3((unsigned)((N + 1 + 255) / 256)), block, 0, s, d_from, d_to, d_val, (const uint32_t *)ord0,
                       (const uint32_t *)ord1, (const uint32_t *)ordf, (const unsigned long long *)off_to,
                       (const unsigned long long *)off_from, (uint32_t)N, st.d_adj_off, st.d_adj_id, st.d_adj_val, st.d_radj_id, st.d_radj_val);
    EO_HIP(hipStreamSynchronize(s));
    EO_HIP(hipGetLastError());
    st.tag = tag;
    st.t_total = bg_now() - t0;
    return PORRT_OK;
}

} // namespace porrt
