// porrt_exchange.hpp -- the one exchange of a query-sharded job, behind the C ABI.
//
// Planning queries are independent (map x seed, or the sub-queries of a TAMP search, src/map_shelves_tamp_rrt.rs:163-291):
// query q runs on rank q mod world_size and nothing is communicated while the trees grow.  When a job ends every rank
// holds, per map, its best tree (lowest path cost, RRT::get_best_solution src/rrt.rs:183-193).  The exchange:
//   0. (before each of the steps below) ncclAllGather of a status word per rank: a rank that failed locally takes part, and
//      either every rank goes on or every rank returns (porrt_exchange_agree, pure host code);
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

#include <chrono>
#include <cmath>
#include <cstring>
#include <thread>
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
    uint8_t *d_stage = nullptr;        // device staging of the all-gathers: this rank's bytes, then world x bytes (made before a sequence starts:
    size_t stage_cap = 0;              //   agreeing must not need an allocation)
    struct Tree {
        double *nx = nullptr, *ny = nullptr, *dist = nullptr;
        int *parent = nullptr;
        size_t cap = 0;
        uint32_t n = 0;
    };
    std::vector<Tree> trees;           // per map: the winning tree, on this rank's device
    const porrt_comm_ops *ops = nullptr;   // stand-in transport (porrt_comm_test_new_ops: the CPU tests' collectives over host memory) or null = RCCL
    bool broken = false;               // a collective step failed on this rank: the communicator was aborted and takes no further call
    bool test_mode = false;            // made by porrt_comm_test_new: no RCCL, no device (the failure protocol alone, for the CPU tests)
    int aborts = 0;                    // how often the communicator was aborted (0 or 1)
    int timeout_ms = 120000;           // how long a collective step may stay unfinished before this rank aborts it
    void set_err(const std::string &s) { err = s; }
};

// No rank may leave a collective sequence alone.  Before the first agreement a local failure is a status word the others see
// (comm_agree); AFTER it, a rank that fails locally -- a HIP or RCCL call returning an error, an RCCL asynchronous error, a step
// that does not finish within timeout_ms -- ABORTS the communicator before it returns: ncclCommAbort tears this rank's
// connections down, the peers' pending collectives see an asynchronous error (comm_wait polls ncclCommGetAsyncError, no
// hipStreamSynchronize that could wait for ever) and abort theirs.  The communicator is then unusable: every later call
// returns PORRT_ERR_DEVICE at once, the job makes a new one.
static int comm_fail(porrt_comm *c, int code, const std::string &msg) {
    c->set_err(msg + " -- communicator aborted, make a new one");
    if (c->comm) { (void)ncclCommAbort(c->comm); c->comm = nullptr; }
    if (c->ops && c->ops->abort) (void)c->ops->abort(c->ops->self);
    c->broken = true;
    ++c->aborts;
    return code;
}

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

#define XABORT(c, x)                                                                                \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return comm_fail((c), PORRT_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define NABORT(c, x)                                                                                \
    do {                                                                                            \
        ncclResult_t r_ = (x);                                                                      \
        if (r_ != ncclSuccess) return comm_fail((c), PORRT_ERR_DEVICE, std::string(#x) + ": " + ncclGetErrorString(r_)); \
    } while (0)

// Waits for the communicator's stream without ever waiting for good: the stream is polled together with RCCL's asynchronous
// error state, and a step that is neither done nor failed after timeout_ms is given up (a peer that died, or left).
static int comm_wait(porrt_comm *c, const char *what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) return PORRT_OK;
        if (q != hipErrorNotReady) return comm_fail(c, PORRT_ERR_DEVICE, std::string("exchange_best: ") + what + ": " + hipGetErrorString(q));
        ncclResult_t ar = ncclSuccess;
        if (c->comm && ncclCommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
            return comm_fail(c, PORRT_ERR_PEER, std::string("exchange_best: ") + what + ": RCCL reports " + ncclGetErrorString(ar) + " (a peer failed or left)");
        const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        if (ms > c->timeout_ms) return comm_fail(c, PORRT_ERR_PEER, std::string("exchange_best: ") + what + " did not finish within " + std::to_string(c->timeout_ms) + " ms");
        if (spin > 64) std::this_thread::sleep_for(std::chrono::microseconds(spin > 4096 ? 1000 : 20));
    }
}

// ---- the transport under the exchange: RCCL on the communicator's stream, or the stand-in a test handed in (host memory all the way:
// "device" buffers are whatever its alloc returns).  Every function returns PORRT_OK or has aborted the communicator (comm_fail).
static int xg_alloc(porrt_comm *c, void **p, size_t bytes) {
    *p = nullptr;
    if (c->ops) { *p = c->ops->alloc(c->ops->self, bytes); return *p ? PORRT_OK : PORRT_ERR_DEVICE; }
    return hipMalloc(p, bytes) == hipSuccess ? PORRT_OK : PORRT_ERR_DEVICE;
}
static void xg_free(porrt_comm *c, void *p) {
    if (!p) return;
    if (c->ops) c->ops->release(c->ops->self, p);
    else (void)hipFree(p);
}
// every rank's `bytes` at h_send, gathered in rank order into h_recv (world x bytes); host memory on both sides
static int xg_all_gather(porrt_comm *c, const void *h_send, void *h_recv, size_t bytes, const char *what) {
    if (c->ops) {
        const int r = c->ops->all_gather(c->ops->self, h_send, h_recv, bytes);
        return r ? comm_fail(c, PORRT_ERR_DEVICE, std::string("exchange_best: ") + what + ": the transport's all-gather failed (" + std::to_string(r) + ")") : PORRT_OK;
    }
    const size_t need = bytes * ((size_t)c->world + 1);
    if (c->stage_cap < need) {                    // (grown before the sequence starts: exchange_tables reserves the largest size it will use)
        if (c->d_stage) (void)hipFree(c->d_stage);
        c->d_stage = nullptr; c->stage_cap = 0;
        if (hipMalloc((void **)&c->d_stage, need) != hipSuccess) return comm_fail(c, PORRT_ERR_DEVICE, std::string("exchange_best: ") + what + ": hipMalloc of the gather buffer");
        c->stage_cap = need;
    }
    XABORT(c, hipMemcpyAsync(c->d_stage, h_send, bytes, hipMemcpyHostToDevice, c->stream));
    NABORT(c, ncclAllGather(c->d_stage, c->d_stage + bytes, bytes, ncclUint8, c->comm, c->stream));
    XABORT(c, hipMemcpyAsync(h_recv, c->d_stage + bytes, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
    return comm_wait(c, what);
}
struct XgBcast { const void *src; void *dst; size_t bytes; int root; };
// all broadcasts of one exchange, issued together (one RCCL group); every rank passes the same list (src only matters on the root)
static int xg_broadcasts(porrt_comm *c, const std::vector<XgBcast> &items, const char *what) {
    if (c->ops) {
        for (const XgBcast &it : items) {
            const int r = c->ops->broadcast(c->ops->self, it.src, it.dst, it.bytes, it.root);
            if (r) return comm_fail(c, PORRT_ERR_DEVICE, std::string("exchange_best: ") + what + ": the transport's broadcast failed (" + std::to_string(r) + ")");
        }
        return PORRT_OK;
    }
    ncclResult_t nerr = ncclGroupStart();
    if (nerr != ncclSuccess) return comm_fail(c, PORRT_ERR_DEVICE, std::string("ncclGroupStart: ") + ncclGetErrorString(nerr));
    for (const XgBcast &it : items) {        // an RCCL error is remembered and the group closed all the same
        ncclResult_t r;
        if (nerr == ncclSuccess && (r = ncclBroadcast(it.src, it.dst, it.bytes, ncclUint8, it.root, c->comm, c->stream)) != ncclSuccess) nerr = r;
    }
    const ncclResult_t gend = ncclGroupEnd();
    if (nerr == ncclSuccess) nerr = gend;
    if (nerr != ncclSuccess) return comm_fail(c, PORRT_ERR_DEVICE, std::string("exchange_best: ncclBroadcast: ") + ncclGetErrorString(nerr));
    return comm_wait(c, what);
}
// bytes of a winning tree's buffer to the host
static int xg_fetch(porrt_comm *c, void *h_dst, const void *src, size_t bytes) {
    if (c->ops) return c->ops->fetch(c->ops->self, h_dst, src, bytes) ? PORRT_ERR_DEVICE : PORRT_OK;
    return hipMemcpy(h_dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? PORRT_OK : PORRT_ERR_DEVICE;
}

static void comm_free_tree(porrt_comm *c, porrt_comm::Tree &t) {
    xg_free(c, t.nx); xg_free(c, t.ny); xg_free(c, t.dist); xg_free(c, t.parent);
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
    c->stage_cap = 4096 * ((size_t)world + 1);          // room for the status words and for tables of 256 maps without another allocation
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void **)&c->d_stage, c->stage_cap) != hipSuccess ||
        ncclCommInitRank(&c->comm, world, u, rank) != ncclSuccess) {
        if (c->d_stage) (void)hipFree(c->d_stage);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return nullptr;
    }
    return c;
}

void porrt_comm_destroy(porrt_comm *c) {
    if (!c) return;
    if (c->test_mode) { for (auto &t : c->trees) comm_free_tree(c, t); delete c; return; }
    (void)hipSetDevice(c->device);
    if (c->stream && !c->broken) (void)hipStreamSynchronize(c->stream);
    for (auto &t : c->trees) comm_free_tree(c, t);
    if (c->d_stage) (void)hipFree(c->d_stage);
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

// The worst status of a collective step, decided alike on every rank from the gathered status words (pure host code):
// word r = {code of rank r (0 or a negative PORRT_ERR_*), the n_maps rank r was called with}.  Returns 0 when every rank
// is fine and all agree on n_maps; otherwise the code this rank must return -- its own if it failed, PORRT_ERR_PEER if only
// others did, PORRT_ERR_INVALID on every rank when the map counts differ -- and the first failing rank in *bad_rank.
int porrt_exchange_agree(const int32_t *words /* world x 2 */, uint32_t world, uint32_t my_rank, int32_t *bad_rank) {
    if (!words || world == 0 || my_rank >= world) return PORRT_ERR_INVALID;
    int32_t bad = -1;
    for (uint32_t r = 0; r < world && bad < 0; ++r)
        if (words[2 * r] != 0) bad = (int32_t)r;
    if (bad_rank) *bad_rank = bad;
    if (bad >= 0) return words[2 * my_rank] != 0 ? words[2 * my_rank] : PORRT_ERR_PEER;
    for (uint32_t r = 1; r < world; ++r)
        if (words[2 * r + 1] != words[1]) { if (bad_rank) *bad_rank = (int32_t)r; return PORRT_ERR_INVALID; }
    return PORRT_OK;
}

} // extern "C"

// All ranks meet here before every data step: a rank that failed locally still takes part, so nobody is left waiting in a
// collective the failed rank never enters.  16 bytes per rank through buffers made with the communicator.
static int comm_agree(porrt_comm *c, int local, uint32_t n_maps, const char *what) {
    int32_t mine[2] = {(int32_t)local, (int32_t)n_maps};
    std::vector<int32_t> all(2 * (size_t)c->world);
    // (a failure in here is a failure inside a collective step: the peers are in the same all-gather)
    { const int w = xg_all_gather(c, mine, all.data(), sizeof mine, what); if (w != PORRT_OK) return w; }
    int32_t bad = -1;
    const int r = porrt_exchange_agree(all.data(), (uint32_t)c->world, (uint32_t)c->rank, &bad);
    if (r == PORRT_ERR_PEER) c->set_err(std::string("exchange_best: rank ") + std::to_string(bad) + " failed " + what + " (code " + std::to_string(all[2 * bad]) + "); no rank went on");
    else if (r != PORRT_OK && local == PORRT_OK) c->set_err("exchange_best: the ranks were called with different numbers of maps");
    return r;
}

// The collective part of the exchange on this rank's per-map entries (cost, n_nodes) and the arrays of its trees: steps 0-3 of the
// header.  `local` = what went wrong on this rank before it got here (the others learn it in the first agreement).
static int exchange_tables(porrt_comm *c, int local, const porrt_best_entry *mine, const porrt_tree_device_view *view, uint32_t n_maps, porrt_best_entry *winners) {
    auto fail = [&](int code, const std::string &msg) { if (local == PORRT_OK) { local = code; c->set_err(msg); } };
    const size_t need = (size_t)n_maps * (size_t)c->world;
    // the gather buffer for the tables, before the sequence starts (a failure here is still a status word)
    if (local == PORRT_OK && !c->ops && c->stage_cap < n_maps * sizeof(porrt_best_entry) * ((size_t)c->world + 1)) {
        if (c->d_stage) (void)hipFree(c->d_stage);
        c->d_stage = nullptr; c->stage_cap = 0;
        const size_t want = n_maps * sizeof(porrt_best_entry) * ((size_t)c->world + 1);
        if (hipMalloc((void **)&c->d_stage, want) != hipSuccess) fail(PORRT_ERR_DEVICE, "exchange_best: hipMalloc of the gathered tables");
        else c->stage_cap = want;
    }
    if (!c->ops && !c->d_stage) {                 // (no buffer to agree through: nothing collective can be said any more)
        return comm_fail(c, PORRT_ERR_DEVICE, "exchange_best: no gather buffer");
    }
    int st = comm_agree(c, local, n_maps, "before the all-gather");
    if (st != PORRT_OK) return st;
    // ---- 1. all-gather of the tables (from here on every rank is inside the sequence: a local failure aborts the communicator)
    std::vector<porrt_best_entry> all(need);
    st = xg_all_gather(c, mine, all.data(), n_maps * sizeof(porrt_best_entry), "the all-gather of the tables");
    if (st != PORRT_OK) return st;
    // ---- 2. winners (the same decision on every rank), and room for their trees -- allocated before the broadcasts are agreed on
    std::vector<int32_t> win(n_maps);
    porrt_exchange_decide(all.data(), (uint32_t)c->world, n_maps, win.data());
    if (c->trees.size() < n_maps) c->trees.resize(n_maps);
    for (uint32_t m = 0; m < n_maps; ++m) {
        porrt_comm::Tree &t = c->trees[m];
        t.n = 0;
        if (win[m] < 0 || local != PORRT_OK) continue;
        const size_t n = (size_t)all[(size_t)win[m] * n_maps + m].n_nodes;
        if (t.cap < n) {
            comm_free_tree(c, t);
            if (xg_alloc(c, (void **)&t.nx, n * 8) || xg_alloc(c, (void **)&t.ny, n * 8) || xg_alloc(c, (void **)&t.dist, n * 8) || xg_alloc(c, (void **)&t.parent, n * 4)) {
                comm_free_tree(c, t);
                fail(PORRT_ERR_DEVICE, "exchange_best: allocation of a winning tree's buffers");
                continue;
            }
            t.cap = n;
        }
        if (win[m] == c->rank && view[m].n_nodes != n) fail(PORRT_ERR_INVALID, "exchange_best: the winning context's tree changed during the exchange");
    }
    st = comm_agree(c, local, n_maps, "before the broadcasts");
    if (st != PORRT_OK) return st;
    // ---- 3. the winning trees, device to device.  Every rank issues the same calls.
    std::vector<XgBcast> items;
    for (uint32_t m = 0; m < n_maps; ++m) {
        porrt_comm::Tree &t = c->trees[m];
        if (win[m] < 0) { winners[m].cost = std::numeric_limits<double>::infinity(); winners[m].rank = -1; winners[m].n_nodes = 0; continue; }
        winners[m] = all[(size_t)win[m] * n_maps + m];
        const size_t n = (size_t)winners[m].n_nodes;
        t.n = (uint32_t)n;
        const void *sx = t.nx, *sy = t.ny, *sd = t.dist, *sp = t.parent;
        if (win[m] == c->rank) { sx = view[m].nx; sy = view[m].ny; sd = view[m].dist_root; sp = view[m].parent; }
        items.push_back({sx, t.nx, n * 8, win[m]});
        items.push_back({sy, t.ny, n * 8, win[m]});
        items.push_back({sd, t.dist, n * 8, win[m]});
        items.push_back({sp, t.parent, n * 4, win[m]});
    }
    st = xg_broadcasts(c, items, "the broadcasts of the winning trees");
    if (st != PORRT_OK) { for (auto &t : c->trees) t.n = 0; return st; }
    return PORRT_OK;
}

extern "C" {

// Collective-safe by construction: everything that can fail on one rank alone (arguments, the cost evaluation, the view of
// the winning context's arrays, allocation) happens BEFORE a collective step and its outcome is agreed on by all ranks
// (comm_agree) -- either every rank enters the step or none does; nothing returns between ncclGroupStart and ncclGroupEnd.
int porrt_exchange_best(porrt_comm *c, porrt_ctx *const *ctxs, uint32_t n_ctx, const uint32_t *map_ids, uint32_t n_maps, porrt_best_entry *winners) {
    if (!c) return PORRT_ERR_INVALID;                        // no communicator: nothing to take part with
    if (c->broken || !c->comm) { c->set_err("exchange_best: this communicator was aborted after a failed collective step (or is a test stand-in): make a new one"); return PORRT_ERR_DEVICE; }
    int local = PORRT_OK;
    auto fail = [&](int code, const std::string &msg) { if (local == PORRT_OK) { local = code; c->set_err(msg); } };
    if (hipSetDevice(c->device) != hipSuccess) fail(PORRT_ERR_DEVICE, "exchange_best: hipSetDevice");
    if ((!ctxs && n_ctx) || (!map_ids && n_ctx) || !n_maps || !winners) fail(PORRT_ERR_INVALID, "exchange_best: arguments");
    // ---- local: this rank's best context per map (costs evaluated on the device: one launch for the members of a batch)
    std::vector<porrt_best_entry> mine(n_maps);
    std::vector<porrt_tree_device_view> view(n_maps);
    for (uint32_t m = 0; m < n_maps; ++m) { mine[m].cost = std::numeric_limits<double>::infinity(); mine[m].rank = c->rank; mine[m].n_nodes = 0; view[m] = porrt_tree_device_view{}; }
    if (local == PORRT_OK && n_ctx) {
        std::vector<double> costs(n_ctx, std::numeric_limits<double>::infinity());
        std::vector<int> best_ctx(n_maps, -1);
        for (uint32_t q = 0; q < n_ctx && local == PORRT_OK; ++q) {
            if (!ctxs[q]) fail(PORRT_ERR_INVALID, "exchange_best: null context");
            else if (map_ids[q] >= n_maps) fail(PORRT_ERR_INVALID, "exchange_best: map id out of range");
        }
        if (local == PORRT_OK) {
            const int r = porrt_best_cost_batch(ctxs, n_ctx, costs.data());
            if (r < 0) fail(r, std::string("exchange_best: ") + porrt_last_error(ctxs[0]));
        }
        for (uint32_t q = 0; q < n_ctx && local == PORRT_OK; ++q) {
            const uint32_t m = map_ids[q];
            if (costs[q] < mine[m].cost) { mine[m].cost = costs[q]; best_ctx[m] = (int)q; }      // first minimum in context order (rrt.rs:190 keeps the first, too)
        }
        for (uint32_t m = 0; m < n_maps && local == PORRT_OK; ++m) {
            if (best_ctx[m] < 0) continue;
            view[m] = porrt_tree_device(ctxs[best_ctx[m]]);                                  // empty unless the context holds an RRT* tree
            if (!view[m].nx || !view[m].ny || !view[m].dist_root || !view[m].parent || view[m].n_nodes == 0 || view[m].n_nodes > 0x7fffffffull)
                fail(PORRT_ERR_INVALID, "exchange_best: a context with a solution holds no RRT* tree on the device (contexts of mode PORRT_MODE_RRT only)");
            else mine[m].n_nodes = (int32_t)view[m].n_nodes;
        }
    }
    return exchange_tables(c, local, mine.data(), view.data(), n_maps, winners);
}

int porrt_exchange_tables(porrt_comm *c, const porrt_best_entry *mine, const porrt_tree_device_view *views, uint32_t n_maps, porrt_best_entry *winners) {
    if (!c) return PORRT_ERR_INVALID;
    if (c->broken || (!c->comm && !c->ops)) { c->set_err("exchange_tables: this communicator was aborted after a failed collective step: make a new one"); return PORRT_ERR_DEVICE; }
    int local = PORRT_OK;
    if (!c->ops && hipSetDevice(c->device) != hipSuccess) { local = PORRT_ERR_DEVICE; c->set_err("exchange_tables: hipSetDevice"); }
    if (!mine || !views || !n_maps || !winners) { if (local == PORRT_OK) { local = PORRT_ERR_INVALID; c->set_err("exchange_tables: arguments"); } }
    std::vector<porrt_best_entry> m2(n_maps ? n_maps : 1);
    std::vector<porrt_tree_device_view> v2(n_maps ? n_maps : 1);
    for (uint32_t m = 0; m < n_maps; ++m) {
        m2[m].cost = std::numeric_limits<double>::infinity(); m2[m].rank = c->rank; m2[m].n_nodes = 0; v2[m] = porrt_tree_device_view{};
        if (local != PORRT_OK) continue;
        m2[m] = mine[m]; m2[m].rank = c->rank; v2[m] = views[m];
        if (m2[m].n_nodes > 0 && (!v2[m].nx || !v2[m].ny || !v2[m].dist_root || !v2[m].parent || v2[m].n_nodes != (uint64_t)m2[m].n_nodes)) {
            local = PORRT_ERR_INVALID; c->set_err("exchange_tables: an entry with nodes needs its tree's arrays (n_nodes of the view = n_nodes of the entry)");
        }
    }
    return exchange_tables(c, local, m2.data(), v2.data(), n_maps, winners);
}

// 1 while the communicator takes calls; 0 once a collective step failed on this rank (it was aborted then: make a new one)
int porrt_comm_usable(const porrt_comm *c) { return c && !c->broken && (c->comm || c->test_mode) ? 1 : 0; }

int porrt_comm_set_timeout_ms(porrt_comm *c, int ms) {
    if (!c || ms < 1) return PORRT_ERR_INVALID;
    c->timeout_ms = ms;
    return PORRT_OK;
}

// ---- for the CPU tests of the failure protocol (no RCCL, no device): a stand-in communicator and the ONE function every
// failure of the real path goes through.  stage 0: a local failure before the first agreement (a status word: the communicator
// stays usable, the code comes back); stage >= 1: a failure at or after the first agreement (inside comm_agree, in the all-gather,
// between the agreements, in the broadcast group, a timed-out wait): the communicator is aborted and refuses every later call.
porrt_comm *porrt_comm_test_new(int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) return nullptr;
    porrt_comm *c = new porrt_comm();
    c->rank = rank; c->world = world; c->test_mode = true;
    return c;
}
// a communicator over a transport the caller brings (the CPU tests: collectives over host memory between processes); porrt_exchange_tables
// and the porrt_exchange_* getters work on it, porrt_exchange_best (which evaluates contexts on a device) does not
porrt_comm *porrt_comm_test_new_ops(int rank, int world, const porrt_comm_ops *ops) {
    if (world < 1 || rank < 0 || rank >= world || !ops || !ops->all_gather || !ops->broadcast || !ops->alloc || !ops->release || !ops->fetch) return nullptr;
    porrt_comm *c = new porrt_comm();
    c->rank = rank; c->world = world; c->test_mode = true; c->ops = ops;
    return c;
}
int porrt_comm_test_fail(porrt_comm *c, int stage, int code) {
    if (!c || !c->test_mode || code >= 0) return PORRT_ERR_INVALID;
    if (c->broken) { c->set_err("exchange_best: this communicator was aborted after a failed collective step: make a new one"); return PORRT_ERR_DEVICE; }
    if (stage <= 0) { c->set_err("exchange_best: local failure before the first agreement (agreed on by all ranks, nobody is left waiting)"); return code; }
    return comm_fail(c, code, std::string("exchange_best: simulated failure at stage ") + std::to_string(stage));
}
int porrt_comm_test_aborts(const porrt_comm *c) { return c ? c->aborts : -1; }

uint64_t porrt_exchange_num_nodes(const porrt_comm *c, uint32_t map) { return c && map < c->trees.size() ? c->trees[map].n : 0; }

int porrt_exchange_get_tree(const porrt_comm *cc, uint32_t map, double *xy, int64_t *parent, double *dist_root) {
    porrt_comm *c = const_cast<porrt_comm *>(cc);
    if (!c || map >= c->trees.size()) return PORRT_ERR_INVALID;
    const porrt_comm::Tree &t = c->trees[map];
    const size_t n = t.n;
    if (!n) return PORRT_OK;
    if (!c->ops) XCHK(c, hipSetDevice(c->device));
    std::vector<double> hx(n), hy(n);
    std::vector<int> hp(n);
    if (xg_fetch(c, hx.data(), t.nx, n * 8) || xg_fetch(c, hy.data(), t.ny, n * 8) || xg_fetch(c, hp.data(), t.parent, n * 4) ||
        (dist_root && xg_fetch(c, dist_root, t.dist, n * 8))) { c->set_err("exchange_get_tree: copy from the device failed"); return PORRT_ERR_DEVICE; }
    if (xy) for (size_t i = 0; i < n; ++i) { xy[2 * i] = hx[i]; xy[2 * i + 1] = hy[i]; }
    if (parent) for (size_t i = 0; i < n; ++i) parent[i] = hp[i];
    return PORRT_OK;
}

} // extern "C"
