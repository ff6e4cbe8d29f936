// porrt_prm.hpp -- PRM* roadmap growth on the device: PRM::grow_graph / PRM::add_sample (src/prm.rs:38-109).
//
// The reference adds every sample as a node (no steering, no state check, validity id 0), asks its kd-tree for the
// nodes within heuristic_radius(n) of it -- n the graph size with the new node -- and connects, both ways, the ones
// whose transition PTOFuncs::transition_validator accepts.  The samples do not depend on the graph, so the whole
// roadmap is a function of the sample stream alone:
//     edge (j -> i), j < i   <=>   norm2(x_j, x_i) <= heuristic_radius(i + 1)  and  transition_validator(x_j -> x_i) is Some
// and every node's neighbourhood can be evaluated at once -- no batching contract, the sequential semantics exactly.
// Device: bucket all nodes into a uniform grid with cells about as wide as the final radius (count, scan, fill), then one
// wave per node scans the window of cells its own radius reaches for earlier nodes inside that radius (the exact squared-distance
// threshold the growth kernels use, rad_T2), raycasts the hits (traversed_class, neighbour -> new node as the
// reference orders the arguments); a count pass, a scan and a fill pass produce the edge list without a shared cursor.  The order inside a neighbour list (kd pre-order,
// nearest_neighbor.rs:101-117) is restored on the host when the edges are asked for, as for the belief-space graphs.
#pragma once
#include "porrt_belief.hpp"

namespace porrt {

struct PrmConst {
    uint32_t N, G;
    const double *nx, *ny;
    const double *rad_T2;             // [graph size] -> largest squared distance inside heuristic_radius
    double x0, y0, inv_cell;
    uint32_t *cell_cnt;               // [G*G] (count pass: sizes; fill pass: cursors)
    const unsigned long long *cell_off;   // [G*G + 1]
    uint32_t *cell_ids;               // [N]
    uint32_t *efrom, *eto, *ev;
    uint32_t *deg;                    // [N] number of earlier neighbours per node (count pass)
    const unsigned long long *edge_off;   // [N + 1] (fill pass)
    uint32_t *err;
};

__device__ __forceinline__ uint32_t prm_cell_coord(double v, double v0, double inv_cell, uint32_t G) {
    const double t = (v - v0) * inv_cell;
    if (!(t > 0.0)) return 0u;
    const uint32_t c = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;       // monotone in v, clamped: a disc scan over cells is exhaustive
    return c < G ? c : G - 1;
}

template <bool FILL>
__global__ __launch_bounds__(256) void k_prm_bin(PrmConst p) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.N) return;
    const uint32_t cx = prm_cell_coord(as_global(p.nx)[i], p.x0, p.inv_cell, p.G), cy = prm_cell_coord(as_global(p.ny)[i], p.y0, p.inv_cell, p.G);
    const uint32_t cell = cy * p.G + cx;
    const uint32_t at = atomicAdd(&p.cell_cnt[cell], 1u);
    if (FILL) as_global(p.cell_ids)[as_global(p.cell_off)[cell] + at] = i;
}

// One wave per node i >= 1 (node 0 is the start, PRM::init): its earlier neighbours.  FILL = false counts them, FILL =
// true (after the scan of the counts) writes them: no shared cursor, and the edge list comes out grouped by new node.
template <bool FILL>
__global__ __launch_bounds__(256) void k_prm_connect(const RunConst *__restrict__ rcp, PrmConst p) {
    const RunConst &rc = *rcp;
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / 64u + 1u, lane = threadIdx.x & 63u;
    if (i >= p.N) return;
    const double px = as_global(p.nx)[i], py = as_global(p.ny)[i];
    const double T2 = as_global(p.rad_T2)[i + 1];                           // heuristic_radius(self.graph.nodes.len()) with the new node in
    const uint32_t cx = prm_cell_coord(px, p.x0, p.inv_cell, p.G), cy = prm_cell_coord(py, p.y0, p.inv_cell, p.G);
    // cells a disc of this node's radius can touch: the cell coordinate is monotone, so |difference| <= floor(r / cell) + 1
    const double rw = T2 > 0.0 ? sqrt(T2) * p.inv_cell : 0.0;
    const uint32_t w = rw >= (double)p.G ? p.G : (uint32_t)rw + 1u;
    const uint32_t x_lo = cx > w ? cx - w : 0, x_hi = cx + w < p.G ? cx + w : p.G - 1, y_lo = cy > w ? cy - w : 0, y_hi = cy + w < p.G ? cy + w : p.G - 1;
    TableGrid grid; grid.p = rc.cls; grid.W = rc.W;
    uint32_t err = 0, total = 0;
    const unsigned long long out0 = FILL ? as_global(p.edge_off)[i] : 0ull;
    for (uint32_t yy = y_lo; yy <= y_hi; ++yy) {
        // the cells of one row are adjacent in the sorted list: one run per row
        const unsigned long long r0 = as_global(p.cell_off)[yy * p.G + x_lo], r1 = as_global(p.cell_off)[yy * p.G + x_hi + 1];
        for (unsigned long long q0 = r0; q0 < r1; q0 += 64) {
            const unsigned long long q = q0 + lane;
            bool hit = false;
            uint32_t j = 0, vid = 0;
            if (q < r1) {
                j = as_global(p.cell_ids)[q];
                if (j < i) {
                    const double ax = as_global(p.nx)[j], ay = as_global(p.ny)[j];
                    if (dist2(ax, ay, px, py) <= T2) {                       // norm2(node, new_state) <= radius
                        const int cls = traversed_class(rc, grid, ax, ay, px, py, &err);    // transition_validator(node, new_node)
                        const int v = class_to_validity(rc, cls);
                        hit = v >= 0;
                        vid = hit ? (uint32_t)v : 0u;
                    }
                }
            }
            const unsigned long long ballot = __ballot(hit);
            if (FILL && hit) {
                const unsigned long long at = out0 + total + (unsigned long long)__popcll(ballot & ((1ull << lane) - 1ull));
                as_global(p.efrom)[at] = j; as_global(p.eto)[at] = i; as_global(p.ev)[at] = vid;
            }
            total += (uint32_t)__popcll(ballot);
        }
    }
    if (!FILL && lane == 0) as_global(p.deg)[i] = total;
    if (err) atomicOr(p.err, err);
}

// ---- the roadmaps of ALL modes of a multi-modal PRM in one pass (MapShelfDomainTampPRM::grow_mm_prm, porrt_mmprm.hpp): thousands of
// small roadmaps (~130 nodes each, the first modes a few thousand), their nodes laid end to end.  A node's candidates are the
// earlier nodes of its own mode -- few enough that one wave tests them all (no grid), with the threshold of the node's position
// IN ITS MODE; count, scan, fill as k_prm_connect, then every node's neighbours are put into the reference's order (kd pre-order
// of the mode's kd-tree: the rank comes from the host).  Four launches for all modes instead of ~120 API calls per mode.
struct MmConst {
    uint32_t NT;                      // nodes of all modes
    const double *x, *y;
    const uint32_t *base;             // [NT] first node of the node's mode
    const uint32_t *rank;             // [NT] kd pre-order rank within the mode
    const double *rad_T2;
    uint32_t *deg;                    // [NT]
    const unsigned long long *edge_off;   // [NT + 1]
    uint32_t *tmp;                    // [E] neighbours (global index) in index order
    uint32_t *efrom, *eto;            // [E] neighbour -> new node, indices within the mode, the reference's order
    uint32_t *err;
};

template <bool FILL>
__global__ __launch_bounds__(256) void k_mm_connect(const RunConst *__restrict__ rcp, MmConst p) {
    const RunConst &rc = *rcp;
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / 64u, lane = threadIdx.x & 63u;
    if (i >= p.NT) return;
    const uint32_t b = as_global(p.base)[i], li = i - b;
    uint32_t err = 0, total = 0;
    if (li > 0) {
        const double px = as_global(p.x)[i], py = as_global(p.y)[i];
        const double T2 = as_global(p.rad_T2)[li + 1];                      // heuristic_radius(self.graph.nodes.len()) with the new node in
        TableGrid grid; grid.p = rc.cls; grid.W = rc.W;
        const unsigned long long out0 = FILL ? as_global(p.edge_off)[i] : 0ull;
        for (uint32_t j0 = b; j0 < i; j0 += 64u) {
            const uint32_t j = j0 + lane;
            bool hit = false;
            if (j < i) {
                const double ax = as_global(p.x)[j], ay = as_global(p.y)[j];
                if (dist2(ax, ay, px, py) <= T2) {                           // norm2(node, new_state) <= radius
                    const int cls = traversed_class(rc, grid, ax, ay, px, py, &err);    // transition_validator(node, new_node)
                    hit = class_to_validity(rc, cls) >= 0;
                }
            }
            const unsigned long long ballot = __ballot(hit);
            if (FILL && hit) as_global(p.tmp)[out0 + total + (unsigned long long)__popcll(ballot & ((1ull << lane) - 1ull))] = j;
            total += (uint32_t)__popcll(ballot);
        }
    }
    if (!FILL && lane == 0) as_global(p.deg)[i] = total;
    if (err) atomicOr(p.err, err);
}

// The pre-order rank of every node in the reference's kd-tree of its segment (KdTree::add in node order, nearest_neighbor.rs:29-46;
// the order KdTree::nearest_neighbors lists nodes in, :101-117), for many segments at once -- one workgroup per segment (a mode's
// roadmap nodes).  The tree is built a level per round: every node not yet placed bids for the empty child slot its descent has
// reached (atomicMin of the node's index: sequential insertion gives the slot to the lowest index among the nodes whose paths reach
// it, and those all reach it in the same round), the winner is placed, the others step below it.  Then every node counts itself
// into its ancestors (subtree sizes) and walks to the root once more for its rank: 1 per step down, plus the left subtree where the
// path turns right.  child / parent / aux / sz are scratch of NT entries (child: 2 NT); values written by atomics are read back with
// atomic loads (they are made in L2, past the CU's L1).
constexpr int kSegKdEmpty = 0x7FFFFFFF;
__global__ __launch_bounds__(256) void k_seg_kd_ranks(const double *__restrict__ x, const double *__restrict__ y, const uint32_t *__restrict__ seg_off,
                                                      int *__restrict__ child, int *__restrict__ parent, uint32_t *__restrict__ aux, uint32_t *__restrict__ sz,
                                                      uint32_t *__restrict__ rank) {
    const uint32_t o = seg_off[blockIdx.x], n = seg_off[blockIdx.x + 1] - o;
    if (n == 0) return;
    constexpr uint32_t kPlaced = 0x80000000u;
    auto gx = as_global(x) + o, gy = as_global(y) + o;
    int *ch = child + 2 * (size_t)o, *par = parent + o;
    uint32_t *dep = aux + o, *size = sz + o;
    for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
        ch[2 * t] = kSegKdEmpty; ch[2 * t + 1] = kSegKdEmpty;
        par[t] = t ? 0 : -1;                   // (until a node is placed: the node its descent stands at)
        dep[t] = t ? 0u : kPlaced;             // depth of that node; the root is placed
        size[t] = 1u;
    }
    __syncthreads();
    auto slot_of = [&](uint32_t t, int c, uint32_t d) {
        const bool left = (d & 1u) ? (gy[t] < gy[c]) : (gx[t] < gx[c]);          // nearest_neighbor.rs:33-35: strictly less goes left
        return 2 * c + (left ? 0 : 1);
    };
    for (;;) {
        for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
            const uint32_t d = dep[t];
            if (d & kPlaced) continue;
            atomicMin(&ch[slot_of(t, par[t], d)], (int)t);
        }
        __syncthreads();
        int more = 0;
        for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
            const uint32_t d = dep[t];
            if (d & kPlaced) continue;
            const int w = __hip_atomic_load(&ch[slot_of(t, par[t], d)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (w == (int)t) dep[t] = kPlaced | (d + 1u);
            else { par[t] = w; dep[t] = d + 1u; more = 1; }
        }
        if (!__syncthreads_or(more)) break;
    }
    // subtree sizes: every node counts itself into its ancestors
    for (uint32_t t = threadIdx.x; t < n; t += blockDim.x)
        for (int a = par[t]; a >= 0; a = par[a]) atomicAdd(&size[a], 1u);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
        uint32_t r = 0;
        int v = (int)t;
        for (int a = par[v]; a >= 0; v = a, a = par[a]) {
            r += 1u;
            const int l = __hip_atomic_load(&ch[2 * a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (l != v && l != kSegKdEmpty) r += __hip_atomic_load(&size[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // the path turns right at a: a's left subtree comes first
        }
        rank[o + t] = r;
    }
}

// a node's neighbours by ascending pre-order rank (ranks are distinct): position = number of smaller ranks in the list
__global__ __launch_bounds__(256) void k_mm_order(MmConst p) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / 64u, lane = threadIdx.x & 63u;
    if (i >= p.NT) return;
    const unsigned long long o0 = as_global(p.edge_off)[i], o1 = as_global(p.edge_off)[i + 1];
    const uint32_t d = (uint32_t)(o1 - o0), b = as_global(p.base)[i];
    for (uint32_t e = lane; e < d; e += 64u) {
        const uint32_t j = as_global(p.tmp)[o0 + e], r = as_global(p.rank)[j];
        uint32_t pos = 0;
        for (uint32_t f = 0; f < d; ++f) pos += as_global(p.rank)[as_global(p.tmp)[o0 + f]] < r ? 1u : 0u;
        as_global(p.efrom)[o0 + pos] = j - b;
        as_global(p.eto)[o0 + pos] = i - b;
    }
}

// ---- PRM::plan_path: dijkstra from the goal's node (pto_graph.rs:275-303) as sweeps over the roadmap's device adjacency
// (the same monotone relaxation as the expected costs, porrt_dp.hpp: any order ends in the same fixpoint).
__global__ __launch_bounds__(256) void k_prm_weights(uint32_t N, const unsigned long long *__restrict__ adj_off, const uint32_t *__restrict__ adj_id,
                                                     const double *__restrict__ nx, const double *__restrict__ ny, double *__restrict__ w) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double x = as_global(nx)[n], y = as_global(ny)[n];
    for (unsigned long long k = as_global(adj_off)[n]; k < as_global(adj_off)[n + 1]; ++k) {
        const uint32_t c = as_global(adj_id)[k];
        w[k] = sqrt(dist2(x, y, as_global(nx)[c], as_global(ny)[c]));          // cost_evaluator(u.state, v.state) = norm2
    }
}

__global__ __launch_bounds__(256) void k_prm_sssp_init(uint32_t N, uint32_t goal, const unsigned long long *__restrict__ adj_off,
                                                       const uint32_t *__restrict__ adj_id, double *__restrict__ dist, uint8_t *__restrict__ dirty_a,
                                                       uint8_t *__restrict__ dirty_b) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    dist[n] = n == goal ? 0.0 : __builtin_huge_val();
    if (n == goal)                                           // (both flag arrays were zeroed by the host before this launch)
        for (unsigned long long k = adj_off[n]; k < adj_off[n + 1]; ++k) dirty_b[adj_id[k]] = 1;       // evaluated by the first sweep (which reads b)
}

__global__ __launch_bounds__(256) void k_prm_sssp_sweep(uint32_t N, const unsigned long long *__restrict__ adj_off, const uint32_t *__restrict__ adj_id,
                                                        const double *__restrict__ w, double *__restrict__ dist, uint8_t *__restrict__ dirty_in,
                                                        uint8_t *__restrict__ dirty_out, uint32_t *__restrict__ flags, uint32_t slot) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N || !as_global(dirty_in)[n]) return;
    as_global(dirty_in)[n] = 0;
    const double old = as_global(dist)[n];
    if (old == 0.0) return;
    const unsigned long long a0 = as_global(adj_off)[n], a1 = as_global(adj_off)[n + 1];
    double best = old;
#pragma unroll 4
    for (unsigned long long k = a0; k < a1; ++k) {
        const double a = as_global(dist)[as_global(adj_id)[k]] + as_global(w)[k];     // dist[v] + cost(u, v)
        best = a < best ? a : best;
    }
    if (best < old) {
        as_global(dist)[n] = best;
        for (unsigned long long k = a0; k < a1; ++k) as_global(dirty_out)[as_global(adj_id)[k]] = 1;
        as_global(flags)[slot] = 1;
    }
}

struct PrmState {
    uint32_t *d_cell_cnt = nullptr;
    unsigned long long *d_cell_off = nullptr, *d_tot = nullptr, *d_edge_off = nullptr;
    uint32_t *d_cell_ids = nullptr, *d_err = nullptr, *d_deg = nullptr;
    size_t cells_cap = 0, ids_cap = 0;
    double t_device = 0, t_total = 0;
    // plan_path: edge weights, costs to the goal, sweep flags
    double *d_w = nullptr, *d_dist = nullptr;
    uint8_t *d_dirty[2] = {nullptr, nullptr};
    uint32_t *d_flags = nullptr;
    size_t w_cap = 0, dist_cap = 0;
    uint64_t w_tag = ~0ull;
    void free_device() {
        void *all[] = {d_cell_cnt, d_cell_off, d_tot, d_edge_off, d_cell_ids, d_err, d_deg, d_w, d_dist, d_dirty[0], d_dirty[1], d_flags};
        for (void *q : all) if (q) (void)hipFree(q);
        d_cell_cnt = nullptr; d_cell_off = d_tot = d_edge_off = nullptr; d_cell_ids = d_err = d_deg = nullptr;
        d_w = d_dist = nullptr; d_dirty[0] = d_dirty[1] = nullptr; d_flags = nullptr; w_cap = dist_cap = 0; w_tag = ~0ull;
        cells_cap = ids_cap = 0;
    }
    ~PrmState() { free_device(); }
};

} // namespace porrt
