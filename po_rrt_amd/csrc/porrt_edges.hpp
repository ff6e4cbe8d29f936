// porrt_edges.hpp -- the adjacency order of a PTO graph / PRM roadmap, restored on the device.
//
// The growth kernels keep the edge SET (neighbour -> new node, appended in no particular order).  The reference's
// PTOGraph keeps adjacency LISTS: for a new node first add_edge(nbr, new) for its neighbours in the order
// KdTree::nearest_neighbors lists them -- kd pre-order (nearest_neighbor.rs:101-117) --, then add_edge(new, nbr)
// (pto.rs:103-120, prm.rs:96-103).  So the children (== parents) list of node X is
//     [neighbours at its creation, in kd pre-order]  ++  [later nodes that found X, in id order],
// and everything downstream (porrt_get_edges, the belief graph's lists, extract_path) inherits that order.
// The pre-order rank of every node comes from the host (a kd-tree of the coordinates built in id order: sequential
// pointer chasing, 2 ms for 18k nodes); the edges never leave the device: three radix sorts (rocPRIM through hipCUB)
// by (new node, rank of neighbour), (neighbour, new node) and (node, other end), counts and scans for the offsets.
#pragma once
#include "porrt_belief.hpp"

#include <hipcub/hipcub.hpp>

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

// keys of the three orders; `which`: 0 = (to, rank[from]) per edge, 1 = (from, to) per edge, 2 = (node, other end) per direction
__global__ __launch_bounds__(256) void k_eo_keys(const uint32_t *__restrict__ from, const uint32_t *__restrict__ to, const uint32_t *__restrict__ rank,
                                                 size_t E, unsigned long long *__restrict__ k0, unsigned long long *__restrict__ k1,
                                                 unsigned long long *__restrict__ k2, uint32_t *__restrict__ idx, uint32_t *__restrict__ idx2,
                                                 uint32_t *__restrict__ deg_to, uint32_t *__restrict__ deg_from) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint32_t f = as_global(from)[e], t = as_global(to)[e];
    k0[e] = ((unsigned long long)t << 32) | as_global(rank)[f];
    k1[e] = ((unsigned long long)f << 32) | t;
    k2[2 * e] = ((unsigned long long)t << 32) | f;               // t's neighbour f
    k2[2 * e + 1] = ((unsigned long long)f << 32) | t;           // f's neighbour t
    idx[e] = (uint32_t)e;
    idx2[2 * e] = (uint32_t)e; idx2[2 * e + 1] = (uint32_t)e;
    atomicAdd(&deg_to[t], 1u);
    atomicAdd(&deg_from[f], 1u);
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

// adjacency of node x: [creation neighbours: sort 0's run of x] ++ [later nodes: sort 1's run of x]; radj: sort 2's run
__global__ __launch_bounds__(256) void k_eo_adjacency(const uint32_t *__restrict__ from, const uint32_t *__restrict__ to, const uint32_t *__restrict__ val,
                                                      const uint32_t *__restrict__ ord0, const uint32_t *__restrict__ ord1,
                                                      const unsigned long long *__restrict__ key2, const uint32_t *__restrict__ ord2,
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
    for (unsigned long long k = a; k < b; ++k, ++w) { const uint32_t e = ord0[k]; adj_id[w] = from[e]; adj_val[w] = (uint8_t)val[e]; }
    for (unsigned long long k = c; k < d; ++k, ++w) { const uint32_t e = ord1[k]; adj_id[w] = to[e]; adj_val[w] = (uint8_t)val[e]; }
    for (unsigned long long k = base; k < base + (b - a) + (d - c); ++k) {       // sort 2 is grouped by node in the same order
        radj_id[k] = (uint32_t)(key2[k] & 0xFFFFFFFFull);
        radj_val[k] = (uint8_t)val[ord2[k]];
    }
}

// Orders the E edges (d_from, d_to, d_val: the growth kernels' arrays) of a graph of N nodes whose pre-order ranks are h_rank.
static int edge_order_build(EdgeOrderState &st, uint64_t tag, size_t N, size_t E, const uint32_t *d_from, const uint32_t *d_to, const uint32_t *d_val,
                            const std::vector<uint32_t> &h_rank, hipStream_t s, std::string &err) {
    const double t0 = bg_now();
    st.next_slot = 0;
    st.tag = ~0ull;
    st.N = N; st.E = E;
    uint32_t *rank, *idx, *idx_out, *idx2, *idx2_out, *deg_to, *deg_from, *ord1;
    unsigned long long *k0, *k0o, *k1, *k1o, *k2, *k2o, *off_to, *off_from, *tot;
    int r;
    const size_t nblk = (N + 1 + kScanTile - 1) / kScanTile;
    if ((r = eo_alloc(st, st.d_from, E, err)) || (r = eo_alloc(st, st.d_to, E, err)) || (r = eo_alloc(st, st.d_val, E, err)) ||
        (r = eo_alloc(st, st.d_adj_off, N + 1, err)) || (r = eo_alloc(st, st.d_adj_id, 2 * E, err)) || (r = eo_alloc(st, st.d_radj_id, 2 * E, err)) ||
        (r = eo_alloc(st, st.d_adj_val, 2 * E, err)) || (r = eo_alloc(st, st.d_radj_val, 2 * E, err)) ||
        (r = eo_alloc(st, rank, N, err)) || (r = eo_alloc(st, idx, E, err)) || (r = eo_alloc(st, idx_out, E, err)) || (r = eo_alloc(st, ord1, E, err)) ||
        (r = eo_alloc(st, idx2, 2 * E, err)) || (r = eo_alloc(st, idx2_out, 2 * E, err)) || (r = eo_alloc(st, deg_to, N + 1, err)) ||
        (r = eo_alloc(st, deg_from, N + 1, err)) || (r = eo_alloc(st, k0, E, err)) || (r = eo_alloc(st, k0o, E, err)) || (r = eo_alloc(st, k1, E, err)) ||
        (r = eo_alloc(st, k1o, E, err)) || (r = eo_alloc(st, k2, 2 * E, err)) || (r = eo_alloc(st, k2o, 2 * E, err)) ||
        (r = eo_alloc(st, off_to, N + 2, err)) || (r = eo_alloc(st, off_from, N + 2, err)) || (r = eo_alloc(st, tot, nblk + 2, err)))
        return r;
    EO_HIP(hipMemcpyAsync(rank, h_rank.data(), N * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    EO_HIP(hipMemsetAsync(deg_to, 0, (N + 1) * sizeof(uint32_t), s));
    EO_HIP(hipMemsetAsync(deg_from, 0, (N + 1) * sizeof(uint32_t), s));
    const dim3 block(256), egrid((unsigned)((std::max<size_t>(E, 1) + 255) / 256));
    if (E) hipLaunchKernelGGL(k_eo_keys, egrid, block, 0, s, d_from, d_to, (const uint32_t *)rank, E, k0, k1, k2, idx, idx2, deg_to, deg_from);
    bg_scan(deg_to, N + 1, tot, off_to, s);                       // off[x] = edges whose new node (neighbour) is below x; off[N] = E
    bg_scan(deg_from, N + 1, tot, off_from, s);
    if (E) {
        size_t tmp_bytes = 0, tb = 0;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k2, k2o, idx2, idx2_out, (int)(2 * E), 0, 64, s);
        tmp_bytes = tb;
        (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb, k0, k0o, idx, idx_out, (int)E, 0, 64, s);
        tmp_bytes = std::max(tmp_bytes, tb);
        unsigned char *tmp = nullptr;
        if ((r = eo_alloc(st, tmp, tmp_bytes, err))) return r;
        tb = tmp_bytes;
        EO_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k0, k0o, idx, idx_out, (int)E, 0, 64, s));
        tb = tmp_bytes;
        EO_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k1, k1o, idx, ord1, (int)E, 0, 64, s));
        tb = tmp_bytes;
        EO_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, k2, k2o, idx2, idx2_out, (int)(2 * E), 0, 64, s));
        hipLaunchKernelGGL(k_eo_gather, egrid, block, 0, s, d_from, d_to, d_val, (const uint32_t *)idx_out, E, st.d_from, st.d_to, st.d_val);
    }
    hipLaunchKernelGGL(k_eo_adjacency, dim3((unsigned)((N + 1 + 255) / 256)), block, 0, s, d_from, d_to, d_val, (const uint32_t *)idx_out,
                       (const uint32_t *)ord1, (const unsigned long long *)k2o, (const uint32_t *)idx2_out, (const unsigned long long *)off_to,
                       (const unsigned long long *)off_from, (uint32_t)N, st.d_adj_off, st.d_adj_id, st.d_adj_val, st.d_radj_id, st.d_radj_val);
    EO_HIP(hipStreamSynchronize(s));
    EO_HIP(hipGetLastError());
    st.tag = tag;
    st.t_total = bg_now() - t0;
    return PORRT_OK;
}

} // namespace porrt
