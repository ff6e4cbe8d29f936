// porrt_exchange.hpp -- the one exchange of a query-sharded job, behind the C ABI.
//
// Planning queries are independent (map x seed, or the sub-queries of a TAMP search, src/map_shelves_tamp_rrt.rs:163-291):
// query q runs on rank q mod world_size and nothing is communicated while the trees grow.  When a job ends every rank
// holds, per map, its best tree (lowest path cost, RRT::get_best_solution src/rrt.rs:183-193).  The exchange:
//   1. ncclAllGather of one 16-byte entry (cost f64, rank i32, n_nodes i32) per map and rank;
//   2. per map the first minimum of (cost, rank) wins (porrt_exchange_decide, pure host code);
//   3. ncclBroadcast of the winner's node SoA (x, y, dist_root f64; parent i32 = 28 B per node) straight out of the
//      winning context's device arrays into buffers of the communicator on every rank -- device to device, RCCL over xGMI.
// The reference has no counterpart (it is one process); the host side (Rust / C++ / Python) only supplies the rendezvous:
// rank 0 makes a unique id (porrt_comm_unique_id), the caller carries its 128 bytes to the other ranks by whatever means
// it has, every rank calls porrt_comm_create.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/porrt_hip.h"

static_assert(sizeof(porrt_best_entry) == 16, "the gathered entry is 16 bytes (SURVEY 8e)");
static_assert(PORRT_UNIQUE_ID_BYTES == sizeof(ncclUniqueId), "unique id size");

struct porrt_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    porrt_best_entry *d_send = nullptr, *d_recv = nullptr;
    size_t table_cap = 0;
    struct Tree {
        double *nx = nullptr, *ny = nullptr, *dist = nullptr;
        int *parent = nullptr;
        size_t cap = 0;
        uint32_t n = 0;
    };
    std::vector<Tree> trees;           // per map: the winning tree, on this rank's device
    void set_err(const std::string &s) { err = s; }
};

#define XCHK(c, x)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) { (c)->set_err(std::string(#x) + ": " + hipGetErrorString(e_)); return PORRT_ERR_DEVICE; } \
    } while (0)
#define NCHK(c, x)                                                                                  \
    do {                                                                                            \
        ncclResult_t r_ = (x);                                                                      \
        if (r_ != ncclSuccess) { (c)->set_err(std::string(#x) + ": " + ncclGetErrorString(r_)); return PORRT_ERR_DEVICE; } \
    } while (0)

static void comm_free_tree(porrt_comm::Tree &t) {
    if (t.nx) (void)hipFree(t.nx);
    if (t.ny) (void)hipFree(t.ny);
    if (t.dist) (void)hipFree(t.dist);
    if (t.parent) (void)hipFree(t.parent);
    t = porrt_comm::Tree();
}

extern "C" {

int porrt_comm_unique_id(uint8_t id[PORRT_UNIQUE_ID_BYTES]) {
    if (!id) return PORRT_ERR_INVALID;
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return PORRT_ERR_DEVICE;
    memcpy(id, &u, sizeof u);
    return PORRT_OK;
}

porrt_comm *porrt_comm_create(int device, int rank, int world, const uint8_t id[PORRT_UNIQUE_ID_BYTES]) {
    if (world < 1 || rank < 0 || rank >= world || !id) return nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    porrt_comm *c = new porrt_comm();
    c->device = device; c->rank = rank; c->world = world;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || ncclCommInitRank(&c->comm, world, u, rank) != ncclSuccess) {
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    return c;
}

void porrt_comm_destroy(porrt_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &t : c->trees) comm_free_tree(t);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *porrt_comm_last_error(const porrt_comm *c) { return c ? c->err.c_str() : "null communicator"; }

// Winner per map among `world` gathered tables (all[r * n_maps + m]): the first minimum of (cost, rank); a map nobody
// solved (all costs +inf / NaN) has no winner (-1).  Pure host code (no GPU needed).
int porrt_exchange_decide(const porrt_best_entry *all, uint32_t world, uint32_t n_maps, int32_t *win_rank) {
    if (!all || !win_rank || world == 0) return PORRT_ERR_INVALID;
    for (uint32_t m = 0; m < n_maps; ++m) {
        int32_t w = -1;
        double best = std::numeric_limits<double>::infinity();
        for (uint32_t r = 0; r < world; ++r) {
            const porrt_best_entry &e = all[(size_t)r * n_maps + m];
            if (e.n_nodes > 0 && e.cost < best) { best = e.cost; w = (int32_t)r; }      // strict: the lowest rank keeps a tie
        }
        win_rank[m] = w;
    }
    return PORRT_OK;
}

int porrt_exchange_best(porrt_comm *c, porrt_ctx *const *ctxs, uint32_t n_ctx, const uint32_t *map_ids, uint32_t n_maps, porrt_best_entry *winners) {
    if (!c) return PORRT_ERR_INVALID;
    if ((!ctxs && n_ctx) || (!map_ids && n_ctx) || !n_maps || !winners) { c->set_err("exchange_best: arguments"); return PORRT_ERR_INVALID; }
    XCHK(c, hipSetDevice(c->device));
    // this rank's best context per map (costs evaluated on the device: one launch for the members of a batch)
    std::vector<double> costs(n_ctx, std::numeric_limits<double>::infinity());
    if (n_ctx) {
        const int r = porrt_best_cost_batch(ctxs, n_ctx, costs.data());
        if (r < 0) { c->set_err(std::string("exchange_best: ") + porrt_last_error(ctxs[0])); return r; }
    }
    std::vector<porrt_best_entry> mine(n_maps);
    std::vector<int> best_ctx(n_maps, -1);
    for (uint32_t m = 0; m < n_maps; ++m) { mine[m].cost = std::numeric_limits<double>::infinity(); mine[m].rank = c->rank; mine[m].n_nodes = 0; }
    for (uint32_t q = 0; q < n_ctx; ++q) {
        if (map_ids[q] >= n_maps) { c->set_err("exchange_best: map id out of range"); return PORRT_ERR_INVALID; }
        const uint32_t m = map_ids[q];
        if (costs[q] < mine[m].cost) {                       // first minimum in context order (rrt.rs:190 keeps the first, too)
            mine[m].cost = costs[q];
            mine[m].n_nodes = (int32_t)porrt_num_nodes(ctxs[q]);
            best_ctx[m] = (int)q;
        }
    }
    // 1. all-gather of the tables
    const size_t need = (size_t)n_maps * (size_t)c->world;
    if (c->table_cap < need) {
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr; c->table_cap = 0;
        XCHK(c, hipMalloc((void **)&c->d_send, n_maps * sizeof(porrt_best_entry)));
        XCHK(c, hipMalloc((void **)&c->d_recv, need * sizeof(porrt_best_entry)));
        c->table_cap = need;
    }
    XCHK(c, hipMemcpyAsync(c->d_send, mine.data(), n_maps * sizeof(porrt_best_entry), hipMemcpyHostToDevice, c->stream));
    NCHK(c, ncclAllGather(c->d_send, c->d_recv, n_maps * sizeof(porrt_best_entry), ncclUint8, c->comm, c->stream));
    std::vector<porrt_best_entry> all(need);
    XCHK(c, hipMemcpyAsync(all.data(), c->d_recv, need * sizeof(porrt_best_entry), hipMemcpyDeviceToHost, c->stream));
    XCHK(c, hipStreamSynchronize(c->stream));
    // 2. winners
    std::vector<int32_t> win(n_maps);
    porrt_exchange_decide(all.data(), (uint32_t)c->world, n_maps, win.data());
    // 3. the winning trees, device to device
    if (c->trees.size() < n_maps) c->trees.resize(n_maps);
    NCHK(c, ncclGroupStart());
    for (uint32_t m = 0; m < n_maps; ++m) {
        porrt_comm::Tree &t = c->trees[m];
        t.n = 0;
        if (win[m] < 0) { winners[m].cost = std::numeric_limits<double>::infinity(); winners[m].rank = -1; winners[m].n_nodes = 0; continue; }
        winners[m] = all[(size_t)win[m] * n_maps + m];
        const size_t n = (size_t)winners[m].n_nodes;
        if (t.cap < n) {
            comm_free_tree(t);
            XCHK(c, hipMalloc((void **)&t.nx, n * 8)); XCHK(c, hipMalloc((void **)&t.ny, n * 8));
            XCHK(c, hipMalloc((void **)&t.dist, n * 8)); XCHK(c, hipMalloc((void **)&t.parent, n * 4));
            t.cap = n;
        }
        t.n = (uint32_t)n;
        const void *sx = t.nx, *sy = t.ny, *sd = t.dist, *sp = t.parent;
        if (win[m] == c->rank) {
            const porrt_tree_device_view v = porrt_tree_device(ctxs[best_ctx[m]]);
            if (!v.nx || v.n_nodes != n) { (void)ncclGroupEnd(); c->set_err("exchange_best: the winning context lost its tree"); return PORRT_ERR_INVALID; }
            sx = v.nx; sy = v.ny; sd = v.dist_root; sp = v.parent;
        }
        NCHK(c, ncclBroadcast(sx, t.nx, n, ncclFloat64, win[m], c->comm, c->stream));
        NCHK(c, ncclBroadcast(sy, t.ny, n, ncclFloat64, win[m], c->comm, c->stream));
        NCHK(c, ncclBroadcast(sd, t.dist, n, ncclFloat64, win[m], c->comm, c->stream));
        NCHK(c, ncclBroadcast(sp, t.parent, n, ncclInt32, win[m], c->comm, c->stream));
    }
    NCHK(c, ncclGroupEnd());
    XCHK(c, hipStreamSynchronize(c->stream));
    return PORRT_OK;
}

uint64_t porrt_exchange_num_nodes(const porrt_comm *c, uint32_t map) { return c && map < c->trees.size() ? c->trees[map].n : 0; }

int porrt_exchange_get_tree(const porrt_comm *cc, uint32_t map, double *xy, int64_t *parent, double *dist_root) {
    porrt_comm *c = const_cast<porrt_comm *>(cc);
    if (!c || map >= c->trees.size()) return PORRT_ERR_INVALID;
    const porrt_comm::Tree &t = c->trees[map];
    const size_t n = t.n;
    if (!n) return PORRT_OK;
    XCHK(c, hipSetDevice(c->device));
    std::vector<double> hx(n), hy(n);
    std::vector<int> hp(n);
    XCHK(c, hipMemcpy(hx.data(), t.nx, n * 8, hipMemcpyDeviceToHost));
    XCHK(c, hipMemcpy(hy.data(), t.ny, n * 8, hipMemcpyDeviceToHost));
    XCHK(c, hipMemcpy(hp.data(), t.parent, n * 4, hipMemcpyDeviceToHost));
    if (xy) for (size_t i = 0; i < n; ++i) { xy[2 * i] = hx[i]; xy[2 * i + 1] = hy[i]; }
    if (parent) for (size_t i = 0; i < n; ++i) parent[i] = hp[i];
    if (dist_root) XCHK(c, hipMemcpy(dist_root, t.dist, n * 8, hipMemcpyDeviceToHost));
    return PORRT_OK;
}

} // extern "C"
