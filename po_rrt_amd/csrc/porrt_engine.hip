// porrt_engine.hip -- host side of libporrt_hip.so: context, domain preprocessing, the batched growth
// loop (launch schedule of the kernels in porrt_device.hpp) and the C ABI of include/porrt_hip.h.
//
// The host mirrors what the reference does OUTSIDE its per-iteration work:
//   - MapShelfDomain::build/add_zones (src/map_shelves_io.rs:88-148), Map::build/add_zones
//     (src/map_io.rs:90-161): ppm, zone centroids, worlds, world validities;
//   - the loop condition and batching of RRT::grow_tree (src/rrt.rs:109) / PTO::grow_graph (src/pto.rs:67);
//   - heuristic_radius (src/common.rs:357-369) with the platform libm, tabulated per tree size;
//   - the sampler state that persists across plans (src/sample_space.rs, tamp_rrt.rs:196-232);
//   - get_best_solution / get_path_to / get_path_cost (src/rrt.rs:183-193, 48-61, 223-227).
// No CPU fallback exists: without a HIP device porrt_create() fails.
#include "../../include/porrt_hip.h"
#include "porrt_device.hpp"
#include "porrt_group.hpp"
#include "porrt_belief.hpp"
#include "porrt_dp.hpp"
#include "porrt_prm.hpp"
#include "porrt_edges.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <system_error>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <limits>
#include <memory>
#include <vector>

// Nothing is thrown across the C boundary: an entry point whose body can allocate runs inside abi_guard.
template <class F> static inline int abi_guard(F &&f) noexcept {
    try { return f(); } catch (const std::bad_alloc &) { return PORRT_ERR_NOMEM; } catch (...) { return PORRT_ERR_INVALID; }
}

using namespace porrt;
typedef unsigned __int128 u128;

namespace {

#define HIPCHK_CTX(ctx, expr)                                                                      \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->set_err(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
            return PORRT_ERR_DEVICE;                                                               \
        }                                                                                          \
    } while (0)

#define HIPCHK(expr)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_err(std::string(#expr) + ": " + hipGetErrorString(e_));                            \
            return PORRT_ERR_DEVICE;                                                               \
        }                                                                                          \
    } while (0)

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- rand_pcg 0.3 Pcg64 (Lcg128Xsl64) + rand_core 0.6 seed_from_u64 + rand 0.8 gen_range on the host.
// The device generates the continuous stream (k_gen_samples); the host keeps the authoritative state,
// draws the rejection-sampled world indices and is the exact fallback when a float draw would retry.
const u128 PCG_MULT = (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL;
struct Pcg64 {
    u128 state, inc;
    void from_state_incr(u128 s, u128 i) {
        state = s; inc = i;
        state += inc;
        step();
    }
    void step() { state = state * PCG_MULT + inc; }
    void seed_from_u64(uint64_t s) {
        const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
        uint32_t w[8];
        for (int c = 0; c < 8; ++c) {
            s = s * MUL + INC;
            uint32_t xs = (uint32_t)(((s >> 18) ^ s) >> 27), rot = (uint32_t)(s >> 59);
            w[c] = (xs >> rot) | (xs << ((32 - rot) & 31));
        }
        uint64_t q[4];
        for (int i = 0; i < 4; ++i) q[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        from_state_incr((u128)q[0] | ((u128)q[1] << 64), ((u128)q[2] | ((u128)q[3] << 64)) | 1);
    }
    uint64_t next_u64() {
        step();
        uint32_t rot = (uint32_t)(state >> 122);
        uint64_t xsl = (uint64_t)(state >> 64) ^ (uint64_t)state;
        return (xsl >> rot) | (xsl << ((64 - rot) & 63));
    }
    void advance(u128 delta) {
        u128 am = 1, ap = 0, cm = PCG_MULT, cp = inc;
        while (delta > 0) {
            if (delta & 1) { am *= cm; ap = ap * cm + cp; }
            cp = (cm + 1) * cp;
            cm *= cm;
            delta >>= 1;
        }
        state = am * state + ap;
    }
    double gen_range_f64(double low, double high) {
        double scale = high - low;
        for (;;) {
            uint64_t bits = (next_u64() >> 12) | 0x3FF0000000000000ULL;
            double v12;
            memcpy(&v12, &bits, 8);
            volatile double prod = (v12 - 1.0) * scale;
            double res = prod + low;
            if (res < high) return res;
        }
    }
    uint64_t gen_range_usize(uint64_t n) {
        if (n == 0) return next_u64();
        uint64_t zone = (n << __builtin_clzll(n)) - 1;
        for (;;) {
            u128 m = (u128)next_u64() * (u128)n;
            if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
        }
    }
};

// All device buffers live in ONE allocation (2 MiB granules): few large pages instead of dozens of small
// mappings keeps the dependent, scattered loads of the connect / kd kernels out of page-table walks.
struct DevBufBase {
    void *vp = nullptr;
    size_t n = 0, want = 0, elem = 1;
};
template <class T>
struct DevBuf : DevBufBase {
    T *p = nullptr;
    DevBuf() { elem = sizeof(T); }
    hipError_t reserve(size_t w) { if (w > want) want = w; return hipSuccess; }
    void release() {}
};
struct Arena {
    void *base = nullptr;
    size_t cap = 0;
};

uint64_t ones(int n) { return n >= 64 ? ~0ULL : ((1ULL << n) - 1); }

} // namespace

#include "porrt_mmprm.hpp"

// Device scratch that outlives a call: numbered slots that only grow (a caller that builds roadmaps again and again pays for its
// buffers once: a dozen hipMalloc / hipFree pairs cost more than the kernels they serve).
struct GrowScratch {
    std::vector<std::pair<void *, size_t>> slots;
    template <class T> hipError_t get(size_t slot, T *&p, size_t n) {
        if (slots.size() <= slot) slots.resize(slot + 1, {nullptr, 0});
        const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        if (slots[slot].second < bytes) {
            if (slots[slot].first) (void)hipFree(slots[slot].first);
            slots[slot] = {nullptr, 0};
            void *q = nullptr;
            const hipError_t e = hipMalloc(&q, bytes + bytes / 8);
            if (e != hipSuccess) return e;
            slots[slot] = {q, bytes + bytes / 8};
        }
        p = (T *)slots[slot].first;
        return hipSuccess;
    }
    void free_all() { for (auto &sl : slots) if (sl.first) (void)hipFree(sl.first); slots.clear(); }
};

struct porrt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    void set_err(const std::string &s) { err = s; }

    // ---- domain (host copies)
    std::vector<uint8_t> occ, zones, cls;
    uint32_t W = 0, H = 0;
    double low[2] = {0, 0}, ppm = 0;
    int domain = 0;
    bool has_grid = false, cls_dirty = true;
    double visibility = 0;
    int n_zones = 0, n_worlds = 1, n_validities = 1;
    double zone_pos[64][2];
    uint64_t validities[65];
    // ---- samplers
    double s_low[2] = {-1, -1}, s_up[2] = {1, 1};
    Pcg64 crng, drng;
    std::vector<double> inj_xy;
    size_t inj_pos = 0;
    bool has_inj = false, inj_dirty = false;
    std::vector<uint32_t> inj_worlds;
    size_t inj_wpos = 0;
    bool has_inj_worlds = false;
    // ---- goal
    int goal_kind = 0;
    uint32_t G = 0;
    double gcx[64], gcy[64];
    uint64_t gmask[64];
    double g_l1 = 0;
    double w2g[64][2];
    uint32_t obs_zone = 0;
    // ---- options
    bool opt_profile = false;
    bool opt_graph = true;
    uint32_t opt_kd_group = 0;     // steps per kd insertion (0 = choose by K)
    // "kd_after": 1 = the kd structure (tie order) is not built beside the steps: every equal-cost parent is deferred and the
    // structure is built after the last step, group by group with the GPU to itself, then the ties are settled (measured
    // against the default).  0 (default).
    bool opt_kd_after = false;
    int opt_kd_inline = 0;                 // "kd_inline": 1 = the kd groups run on the main stream between the steps (no side stream, no events): a sub-batch
                                           // needs one hardware queue instead of two, so four sub-batches fit the four queues
    int opt_kd_lazy = 1;                   // "kd_lazy": 1 (default; 2 means the same) = with the group kernels and in a single query's one-kernel-per-step form only the goal path of the kd order is kept beside the steps
                                           // (g_track_step) and the whole structure is built after them if a tie needs it; 0 = built beside the steps
    int kd_built_after = 0;
    bool kd_lazy = false, kd_build_now = false;    // in force for the running grow; the build after the steps is being launched
    int kd_full_build(const std::vector<std::pair<uint32_t, uint32_t>> &segs);
    uint32_t opt_claim_threads = 0;        // "kd_claim_threads": 0 = the engine's choice (256 beside a batch's step kernels), else 256 / 512 / 1024
    int opt_host_ranks = 1;                // "host_ranks": 1 (default) = the kd pre-order ranks of a graph's nodes (edge order) from the host kd-tree; 0 = made on the
                                           //   device (k_kd1_*): measured slower for one deep tree (a level per launch), and PRM::plan_path needs the host tree anyway
    int opt_compact = 1;                   // "compact_rows": a batch whose rows end at different steps launches its later steps on the rows that still have work
    uint32_t n_compactions = 0;            //   how often the last such batch led by this context gathered them ("compactions")
    int opt_kd_ride = 0;                   // "kd_ride": 1 = also for several contexts the hints and deferred ties ride in the next group's locate kernel
    uint32_t kd_after_K = 0;               // batch_K of the running launch sequence (the deferred groups need it)
    // "group_lanes": lanes per sample of the RRT* step kernels.  16 / 32 / 64: k_nn2 + k_conn2 (several samples per wave: fewer
    // waves, hits kept in LDS -- throughput); 0: k_near + k_connect_rrt (one wave per sample: the shortest dependent chain per
    // step -- latency).  -1 (default): by the number of queries advanced together, 16 from 8 queries on, else 0.
    int opt_group_req = -1;
    uint32_t opt_group = 0;        // the choice in force for the running launch sequence
    // "early_wave_steps": the group kernels take one wave per sample (64 lanes, 320 hits in LDS) for this many first steps of a run,
    // where the tree is a dense blob and a sample has hundreds of neighbours, and 16 lanes per sample afterwards
    uint32_t opt_early_wave = 0;
    // "batch_streams": porrt_grow_batch advances its contexts as this many sub-batches side by side, each a launch sequence
    // (hipGraph) of its own on its own streams, so that one sub-batch's kernel tails and its kd side chain are filled by the
    // other's kernels.  0 (default): 2 from 32 contexts on, else 1.
    uint32_t opt_batch_streams = 0;
    // "pipeline": RRT* steps of the one-wave-per-sample kernels as k_step_rrt / k_file_commit (step b + 1 is searched while step
    // b is connected: the chain of dependent kernels of a single query is max(search, connect) + file per step instead of
    // search + connect: 5.9 against 6.65 ms on the bench's query).  1 / 0.  4 (default) below.  2: the same phases as ONE persistent launch
    // (k_coop_rrt: barriers over the grid instead of kernel boundaries, a cooperative launch; for batch_K <= 1024, else as 1).
    // 4: ONE kernel per step (k_step1_rrt: the filing of a step's nodes and its rewire commit run beside the next step; for
    // batch_K <= 1024, else as 1).
    int opt_pipeline = 4;
    bool pipe_on = false;                  // the choice in force for the running launch sequence (set with opt_group)
    bool lag_on = false;                   // pipeline = 4: one kernel per step (k_step1_rrt); the data layout is the pipelined one (pipe_on)
    uint32_t lag_near_done = 0xFFFFFFFFu;  //   the step whose search is already launched
    uint32_t pipe_near_done = 0xFFFFFFFFu; // pipelined: the step whose search and filing are already launched
    // porrt_get_trees (first context of the call): pinned staging slots and copy streams, one per worker thread
    std::vector<void *> dl_pin;
    std::vector<hipStream_t> dl_streams;
    std::vector<size_t> dl_pin_cap;                 // bytes of each pinned slot (slots made by different calls differ)
    hipStream_t dl_out_stream = nullptr;            // porrt_get_trees into pinned arrays: its stream (highest priority),
    TreeOut *d_tree_out = nullptr;                  //   the trees' descriptors on the device
    size_t d_tree_out_cap = 0;
    uint32_t opt_tree_out_blocks = 4;               // "tree_out_blocks": workgroups per tree of that kernel
    uint32_t opt_fetch_workers = 8;                 // "fetch_workers": host threads (each with a copy stream and a staging slot) of the staged porrt_get_trees
    bool sub_eager = false;                        // leader of a sub-batch on measured streams: launch step by step (see porrt_grow_batch)
    int last_launch_mode = 0;                      // how the last porrt_grow_batch led by this context ran: 0 one sequence (hipGraph / eager), G >= 2 sequences
                                                   // on measured streams, -G sequences on the contexts' own streams (the stream probe found no set: e.g. under a profiler)
    uint32_t sub_streams_tried = 0;                 // a probe for this many streams already failed: not repeated call after call
    std::vector<hipStream_t> sub_streams;          // first context of such a call: the sub-batches' main streams (see porrt_grow_batch)
    bool opt_gtrack_side = false;          // "gtrack_side": 1 = a single query's goal-path workgroup as a kernel of its own on the side stream, beside the step kernel (measured slower: 4.59 against 4.24 ms -- a cross-stream dependency per step); 0 (default) = a workgroup of the step kernel
    bool gt_pend = false;                  //   a k_gtrack is in flight on the side stream (its event: ev_kd[gt_par])
    uint32_t gt_par = 0;
    bool opt_box_table = true;             // "box_table": a segment whose end pixels' bounding box is all free is not walked (summed-area table; 0 = always walk)
    bool opt_dp_sweeps = false;            // "dp_sweeps": expected costs by whole-graph sweeps instead of layer by layer
    uint32_t opt_cand_cap = 2048;
    uint64_t edge_per_node = 256, tie_pool_mult = 16;      // pool sizes: grown and the run replayed when one overflows (as the neighbour lists)
    // ---- device buffers
    DevBuf<double> d_nx, d_ny, d_distA, d_distB, d_sx, d_sy, d_qx, d_qy, d_pgxy, d_pgd, d_candxy, d_candval, d_radT2, d_inj, d_ssx, d_ssy, d_bqx, d_bqy, d_t2at;
    DevBuf<int> d_parent, d_qnn, d_qvid, d_pgid, d_candid, d_gid, d_kdup;
    DevBuf<KdRec> d_kdrec;
    DevBuf<KdBox> d_kdbox, d_locbox, d_gndbox;
    DevBuf<KdMove> d_kdlosers;
    DevBuf<int> d_bcscratch;
    DevBuf<BestCost> d_bcout;
    DevBuf<uint32_t> d_bccursor;
    DevBuf<int> d_loccur;
    DevBuf<uint32_t> d_locdcur, d_locgex, d_locflags, d_kdsurv;
    DevBuf<double> d_gx, d_gy, d_gndx, d_gndy, d_kqx, d_kqy;
    DevBuf<int> d_kqvid;
    DevBuf<uint32_t> d_rgcnt, d_rgdir, d_gsnap, d_pendoff, d_pendn, d_pendcur, d_pendstate;
    DevBuf<int> d_pendnew, d_pendpool;
    DevBuf<int> d_rep;
    DevBuf<uint32_t> d_kddepth, d_kdgexit;
    DevBuf<unsigned long long> d_reachA, d_reachB, d_finalmask, d_validmask, d_kdhint, d_rgocc;
    DevBuf<uint8_t> d_vid, d_finalflag, d_cls;
    DevBuf<uint32_t> d_nat, d_sworld, d_candcnt, d_efrom, d_eto, d_etv;
    DevBuf<uint16_t> d_perm, d_bqk;
    DevBuf<uint32_t> d_slotof;
    DevBuf<uint32_t> d_sched_i0, d_sched_nb;       // a batch row's own step plan (RunConst::sched_*)
    DevBuf<Counters> d_cnt;
    DevBuf<RunConst> d_rc;
    DevBuf<PcgJump> d_jump;
    Arena arena;
    std::vector<DevBufBase *> all_bufs;
    int layout_buffers();
    // radius table cache
    std::vector<double> radT2;
    double rad_max_step = -1, rad_search_radius = -1;
    int rad_mode = -1;
    size_t rad_uploaded = 0;
    bool jump_valid = false;               // the sampler's jump table on the device belongs to jump_inc
    PcgJump jump_host;
    u128 jump_inc = 0;
    // ---- run state / results
    RunConst rc;
    int mode = 0;
    uint64_t n_iter = 0, n_nodes = 0, n_steps = 0;
    Counters counters;
    bool complete = false;
    bool have_results = false;
    std::vector<double> h_nx, h_ny, h_dist;
    std::vector<int> h_parent;
    std::vector<unsigned long long> h_reach, h_finalmask;
    std::vector<uint8_t> h_vid, h_finalflag;
    std::vector<uint64_t> h_final_ids;
    std::vector<uint32_t> h_efrom, h_eto, h_etv;
    porrt_metrics metrics;
    std::vector<hipEvent_t> ev_pool;

    int build_cls();
    int ensure_radius_table(double max_step, double search_radius, size_t n_needed);
    int grow(const double start[2], double max_step, double search_radius, uint64_t n_iter_min, uint64_t n_iter_max,
             uint32_t K, int mode);
    int grow_once(const double start[2], double max_step, double search_radius, uint64_t n_iter_min, uint64_t n_iter_max,
                  uint32_t K, int mode, bool host_samples, int stage = 0);
    int finish_batch_member(uint64_t n_iter_done, uint32_t steps, uint32_t own_steps, float device_ms);
    Counters batch_hc;
    Pcg64 batch_drng0;                     // PTO member of a batch: the discrete sampler before the plan's worlds were drawn
    size_t batch_wpos0 = 0;
    uint64_t batch_world_draws = 0;
    uint32_t batch_nodes = 0;
    BatchOut *d_batch_out = nullptr;       // leader of a batch: gathered counters of the members
    uint32_t *d_active = nullptr, *h_active = nullptr;     // leader of a batch with step plans: rows still running per step (device / pinned host)
    RunConst *d_rcarr_c[2] = {nullptr, nullptr};           //   and the rows that still have work, compacted (two buffers in turn), with their indices
    uint32_t *d_live_idx = nullptr;
    size_t rcarr_c_cap = 0;
    size_t active_cap = 0;
    std::vector<RunConst> rc_staging;      // leader of a batch: the members' RunConst, uploaded in one copy
    std::vector<uint32_t> worlds_staging;  // sampled worlds of the last upload (PTO)
    size_t batch_out_cap = 0;
    size_t run_lds_bytes = 0;
    int best_cost_device(double *cost, uint64_t *final_id);
    BeliefGraphState bg;                   // porrt_build_belief_graph: result of the last build (device CSR)
    int build_belief_graph(const double *start_belief, uint32_t n_worlds_in);
    DpState dp;                            // porrt_bg_compute_expected_costs: dist per belief node (device)
    int compute_expected_costs();
    int extract_policy();
    PrmState prm;                          // porrt_grow_prm: grid scratch
    EdgeOrderState eo;                     // adjacency order of the last PTO graph / roadmap (device)
    std::shared_ptr<void> host_kd;         // the kd-tree of the node coordinates on the host (pre-order ranks, nearest nodes)
    uint64_t host_kd_tag = ~0ull;
    int ensure_edge_order();
    int grow_prm(const double start[2], double max_step, double search_radius, uint64_t n_iter);
    MmState mm;                            // porrt_grow_mm_prm: the mode tree and the modes' roadmaps
    GrowScratch mm_scratch;                //   and the device buffers of roadmaps_of_modes, kept across calls
    int grow_mm_prm(const double start[2], const double *initial_belief, uint32_t n_worlds_in, double max_step, double search_radius, uint64_t n_iter_per_belief);
    int roadmaps_of_modes(double max_step, double search_radius);
    int roadmap_of_points(const std::vector<double> &xy, double max_step, double search_radius, std::vector<uint32_t> &efrom, std::vector<uint32_t> &eto, double &dev_s);
    int64_t prm_plan_path(const double start[2], const double goal[2], double *path_xy, uint64_t cap);
    int read_best_cost(double *cost, uint64_t *final_id);
    porrt_ctx *batch_leader = nullptr;     // set by porrt_grow_batch: the context whose RunConst array holds this one
    uint64_t batch_gen = 0, batch_gen_counter = 0;     // which of the leader's batches this context belongs to / the leader's count
    std::vector<porrt_ctx *> batch_members;            // leader: the members of its last batch (their back pointers are cleared when it goes)
    uint64_t bg_graph_tag = 0;             // results_tag the belief graph (and the costs on it) were built from
    uint32_t batch_slot = 0, batch_size = 0;
    RunConst *d_rcarr = nullptr;      // leader of a porrt_grow_batch: the members' RunConst, one per grid row
    size_t rcarr_cap = 0;
    enum : unsigned { DL_TREE = 1, DL_DIST = 2, DL_MASKS = 4, DL_EDGES = 8 };
    int download(unsigned want);
    unsigned got = 0;
    uint64_t results_tag = 0, downloaded_tag = ~0ull;
    void launch_step(uint32_t b, uint32_t i0, uint32_t nb, uint32_t vwords, size_t lds_bytes, bool prof, size_t &ev_used,
                     uint32_t nxt2_i0, uint32_t nxt2_nb);
    void flush_commit();
    const RunConst *launch_rcp = nullptr;     // RunConst array the step kernels read (one row of the grid per entry)
    uint32_t launch_Q = 1;
    uint32_t commit_pend_b = 0xFFFFFFFFu, commit_pend_nb = 0;     // RRT*: step whose rewire phase 2 rides in the next k_near
    hipEvent_t ev_join = nullptr;
    void join_side();
    void launch_kd_group();
    int launch_coop(uint32_t n_steps, uint32_t K, uint64_t n_iter, uint32_t vwords, size_t lds_bytes);
    int coop_blocks = 0;                      // grid of the persistent step loop on this device (0: not looked up yet)
    int ensure_side_stream() {
        if (stream2) return PORRT_OK;
        if (hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking) != hipSuccess) { stream2 = nullptr; set_err("hipStreamCreate (side stream)"); return PORRT_ERR_DEVICE; }
        return PORRT_OK;
    }
    uint32_t kd_b0 = 0, kd_last_b = 0, kd_last_nb = 0, kd_group = 1, kd_gidx = 0;
    uint32_t kd_hint_b0 = 0, kd_hint_ns = 0;   // single query: the group whose hints and deferred ties ride in the next group's locate kernel
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_step_done = nullptr, ev_kd[3] = {nullptr, nullptr, nullptr}, ev_steered = nullptr;
    bool kd_pend[3] = {false, false, false}, side_active = false;
    // cached hipGraph of the steps up to n_iter_min
    hipGraphExec_t graph_exec = nullptr;
    uint64_t graph_key[6] = {0, 0, 0, 0, 0, 0};
};

// ---------------------------------------------------------------------------------------------------------
// Pre-classified raster: one byte per pixel holding what the raycast needs (map_shelves_io.rs:150-156;
// map_io.rs:165-174,190-196).
// (Re)lay out every buffer in the arena when one of them has to grow.  Contents are lost on a re-layout;
// every grow re-initialises what it uses, and the cached uploads are marked stale.
int porrt_ctx::layout_buffers() {
    if (all_bufs.empty()) {
        // the first five are zeroed together at the start of every grow (one memset over the span)
        DevBufBase *list[] = {&d_cnt, &d_rgcnt, &d_validmask, &d_kdhint, &d_pendstate, &d_nx, &d_ny, &d_distA, &d_distB, &d_sx, &d_sy, &d_qx, &d_qy,
                              &d_pgxy, &d_candxy, &d_candval, &d_radT2, &d_inj, &d_parent, &d_qnn, &d_qvid, &d_pgid, &d_candid, &d_gid, &d_kdup,
                              &d_kdrec, &d_gx, &d_gy, &d_rgdir, &d_rep, &d_kdbox, &d_locbox, &d_kdlosers, &d_gsnap, &d_pendoff, &d_pendn, &d_pendcur,
                              &d_pendnew, &d_pendpool, &d_kddepth, &d_kdgexit, &d_reachA, &d_reachB, &d_finalmask, &d_vid, &d_finalflag, &d_cls,
                              &d_nat, &d_sworld, &d_candcnt, &d_efrom, &d_eto, &d_etv, &d_rc, &d_jump, &d_loccur, &d_locdcur, &d_locgex, &d_locflags,
                              &d_kdsurv, &d_gndx, &d_gndy, &d_gndbox, &d_kqx, &d_kqy, &d_kqvid, &d_bcscratch, &d_bcout, &d_bccursor, &d_perm, &d_pgd, &d_slotof, &d_ssx, &d_ssy, &d_bqx, &d_bqy, &d_t2at, &d_bqk, &d_rgocc, &d_sched_i0, &d_sched_nb};
        for (DevBufBase *b2 : list) all_bufs.push_back(b2);
    }
    bool grow_needed = false;
    for (DevBufBase *b2 : all_bufs) if (b2->want > b2->n) grow_needed = true;
    if (!grow_needed) return PORRT_OK;
    size_t total = 0;
    for (DevBufBase *b2 : all_bufs) {
        if (b2->want > b2->n) b2->n = b2->want + b2->want / 8;     // a little headroom
        total += (b2->n * b2->elem + 4095) & ~(size_t)4095;
    }
    total = (total + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    if (total > arena.cap) {
        (void)hipStreamSynchronize(stream);
        if (arena.base) (void)hipFree(arena.base);
        arena.base = nullptr; arena.cap = 0;
        HIPCHK(hipMalloc(&arena.base, total));
        arena.cap = total;
    }
    size_t off = 0;
    for (DevBufBase *b2 : all_bufs) { b2->vp = (char *)arena.base + off; off += (b2->n * b2->elem + 4095) & ~(size_t)4095; }
    d_nx.p = (double *)d_nx.vp; d_ny.p = (double *)d_ny.vp; d_distA.p = (double *)d_distA.vp; d_distB.p = (double *)d_distB.vp;
    d_sx.p = (double *)d_sx.vp; d_sy.p = (double *)d_sy.vp; d_qx.p = (double *)d_qx.vp; d_qy.p = (double *)d_qy.vp;
    d_pgxy.p = (double *)d_pgxy.vp; d_candxy.p = (double *)d_candxy.vp; d_candval.p = (double *)d_candval.vp; d_radT2.p = (double *)d_radT2.vp; d_inj.p = (double *)d_inj.vp;
    d_parent.p = (int *)d_parent.vp; d_qnn.p = (int *)d_qnn.vp; d_qvid.p = (int *)d_qvid.vp; d_pgid.p = (int *)d_pgid.vp;
    d_candid.p = (int *)d_candid.vp; d_gid.p = (int *)d_gid.vp; d_kdup.p = (int *)d_kdup.vp; d_kdrec.p = (KdRec *)d_kdrec.vp;
    d_gx.p = (double *)d_gx.vp; d_gy.p = (double *)d_gy.vp; d_rgcnt.p = (uint32_t *)d_rgcnt.vp; d_rgdir.p = (uint32_t *)d_rgdir.vp;
    d_rep.p = (int *)d_rep.vp;
    d_kdbox.p = (KdBox *)d_kdbox.vp; d_locbox.p = (KdBox *)d_locbox.vp; d_gndbox.p = (KdBox *)d_gndbox.vp; d_kdlosers.p = (KdMove *)d_kdlosers.vp; d_bcscratch.p = (int *)d_bcscratch.vp; d_bcout.p = (BestCost *)d_bcout.vp; d_bccursor.p = (uint32_t *)d_bccursor.vp; d_kdhint.p = (unsigned long long *)d_kdhint.vp;
    d_gsnap.p = (uint32_t *)d_gsnap.vp; d_pendoff.p = (uint32_t *)d_pendoff.vp; d_pendn.p = (uint32_t *)d_pendn.vp; d_pendcur.p = (uint32_t *)d_pendcur.vp;
    d_pendstate.p = (uint32_t *)d_pendstate.vp; d_pendnew.p = (int *)d_pendnew.vp; d_pendpool.p = (int *)d_pendpool.vp; d_kddepth.p = (uint32_t *)d_kddepth.vp; d_kdgexit.p = (uint32_t *)d_kdgexit.vp;
    d_reachA.p = (unsigned long long *)d_reachA.vp; d_reachB.p = (unsigned long long *)d_reachB.vp;
    d_finalmask.p = (unsigned long long *)d_finalmask.vp; d_validmask.p = (unsigned long long *)d_validmask.vp;
    d_vid.p = (uint8_t *)d_vid.vp; d_finalflag.p = (uint8_t *)d_finalflag.vp; d_cls.p = (uint8_t *)d_cls.vp;
    d_nat.p = (uint32_t *)d_nat.vp; d_sworld.p = (uint32_t *)d_sworld.vp; d_candcnt.p = (uint32_t *)d_candcnt.vp;
    d_efrom.p = (uint32_t *)d_efrom.vp; d_eto.p = (uint32_t *)d_eto.vp; d_etv.p = (uint32_t *)d_etv.vp;
    d_loccur.p = (int *)d_loccur.vp; d_locdcur.p = (uint32_t *)d_locdcur.vp; d_locgex.p = (uint32_t *)d_locgex.vp;
    d_locflags.p = (uint32_t *)d_locflags.vp; d_kdsurv.p = (uint32_t *)d_kdsurv.vp;
    d_gndx.p = (double *)d_gndx.vp; d_gndy.p = (double *)d_gndy.vp;
    d_kqx.p = (double *)d_kqx.vp; d_kqy.p = (double *)d_kqy.vp; d_kqvid.p = (int *)d_kqvid.vp;
    d_sched_i0.p = (uint32_t *)d_sched_i0.vp; d_sched_nb.p = (uint32_t *)d_sched_nb.vp;
    d_cnt.p = (Counters *)d_cnt.vp; d_rc.p = (RunConst *)d_rc.vp; d_jump.p = (PcgJump *)d_jump.vp; d_perm.p = (uint16_t *)d_perm.vp; d_pgd.p = (double *)d_pgd.vp; d_slotof.p = (uint32_t *)d_slotof.vp; d_ssx.p = (double *)d_ssx.vp; d_ssy.p = (double *)d_ssy.vp; d_bqx.p = (double *)d_bqx.vp; d_bqy.p = (double *)d_bqy.vp; d_t2at.p = (double *)d_t2at.vp; d_bqk.p = (uint16_t *)d_bqk.vp; d_rgocc.p = (unsigned long long *)d_rgocc.vp;
    // cached uploads are gone
    rad_uploaded = 0;
    cls_dirty = true;
    inj_dirty = true;
    jump_valid = false;
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    return PORRT_OK;
}

// bytes of the raster planes on the device: classes, clearance, (aligned) summed-area table
static inline size_t cls_sat_offset(size_t W, size_t H) { return (2 * W * H + 15) & ~(size_t)15; }
static inline size_t cls_bytes(size_t W, size_t H) { return cls_sat_offset(W, H) + (H + 1) * (W + 1) * sizeof(uint32_t) + 16; }

int porrt_ctx::build_cls() {
    if (!has_grid) return PORRT_OK;
    if (!cls_dirty) return PORRT_OK;
    size_t n = (size_t)W * H;
    cls.resize(n);
    for (size_t p = 0; p < n; ++p) {
        uint8_t v = occ[p];
        uint8_t c;
        if (domain == PORRT_DOMAIN_SHELF) {
            c = v == 255 ? CLS_FREE : (v >= 127 ? CLS_LOW : (v == 0 ? CLS_HIGH0 : CLS_HIGH));
        } else {
            if (v == 255) c = CLS_FREE;
            else if (v == 0) c = CLS_HIGH0;
            else if (zones.empty() || zones[p] == 255 || zones[p] >= 64) c = CLS_BAD;
            else c = (uint8_t)(CLS_ZONE + zones[p]);
        }
        cls[p] = c;
    }
    // clearance plane behind the classes: Chebyshev distance to the nearest pixel that is not free or lies outside the
    // raster (two chamfer passes are exact for that metric), capped at 255
    cls.resize(2 * n);
    {
        uint8_t *d = cls.data() + n;
        auto at = [&](long i, long j) -> int { return (i < 0 || j < 0 || i >= (long)H || j >= (long)W) ? 0 : d[(size_t)i * W + j]; };
        for (long i = 0; i < (long)H; ++i)
            for (long j = 0; j < (long)W; ++j) {
                int v = cls[(size_t)i * W + j] == CLS_FREE ? 255 : 0;
                if (v) v = std::min(v, 1 + std::min(std::min(at(i - 1, j - 1), at(i - 1, j)), std::min(at(i - 1, j + 1), at(i, j - 1))));
                d[(size_t)i * W + j] = (uint8_t)v;
            }
        for (long i = (long)H - 1; i >= 0; --i)
            for (long j = (long)W - 1; j >= 0; --j) {
                int v = d[(size_t)i * W + j];
                if (v) v = std::min(v, 1 + std::min(std::min(at(i + 1, j + 1), at(i + 1, j)), std::min(at(i + 1, j - 1), at(i, j + 1))));
                d[(size_t)i * W + j] = (uint8_t)v;
            }
    }
    // ... and behind that (16-byte aligned) the summed-area table of the pixels that are not free (segment_box_free)
    const size_t sat_at = cls_sat_offset(W, H), Ws = (size_t)W + 1;
    cls.resize(sat_at + (H + 1) * Ws * sizeof(uint32_t));
    {
        std::vector<uint32_t> sat((H + 1) * Ws, 0u);
        for (size_t i = 0; i < H; ++i) {
            uint32_t row = 0;
            for (size_t j = 0; j < W; ++j) {
                row += cls[i * W + j] == CLS_FREE ? 0u : 1u;
                sat[(i + 1) * Ws + j + 1] = sat[i * Ws + j + 1] + row;
            }
        }
        memcpy(cls.data() + sat_at, sat.data(), sat.size() * sizeof(uint32_t));
    }
    HIPCHK(hipMemcpyAsync(d_cls.p, cls.data(), cls.size(), hipMemcpyHostToDevice, stream));
    cls_dirty = false;
    return PORRT_OK;
}

// heuristic_radius(n) (src/common.rs:357-369, platform libm as Rust's f64::ln/powf) turned into the exact
// threshold on the squared distance: norm2 <= radius  <=>  d2 <= T2 with T2 = max{t : sqrt(t) <= radius}.
int porrt_ctx::ensure_radius_table(double max_step, double search_radius, size_t n_needed) {
    if (max_step != rad_max_step || search_radius != rad_search_radius) {
        radT2.clear();
        rad_uploaded = 0;
        rad_max_step = max_step;
        rad_search_radius = search_radius;
    }
    size_t have = radT2.size();
    if (have < n_needed) {
        radT2.resize(n_needed);
        for (size_t n = have; n < n_needed; ++n) {
            double nn = (double)n;
            double s = search_radius * pow(log(nn) / nn, 1.0 / 2.0);
            double r = s < max_step ? s : max_step;
            double t;
            if (!(r >= 0.0)) {
                t = -1.0;   // n = 0 is never used
            } else {
                t = r * r;
                while (sqrt(nextafter(t, INFINITY)) <= r) t = nextafter(t, INFINITY);
                while (t > 0.0 && sqrt(t) > r) t = nextafter(t, -INFINITY);
            }
            radT2[n] = t;
        }
    }
    if (d_radT2.n < n_needed) { set_err("radius table capacity"); return PORRT_ERR_INVALID; }
    if (rad_uploaded < n_needed) {
        HIPCHK(hipMemcpyAsync(d_radT2.p + rad_uploaded, radT2.data() + rad_uploaded, (n_needed - rad_uploaded) * sizeof(double),
                              hipMemcpyHostToDevice, stream));
        rad_uploaded = n_needed;
    }
    return PORRT_OK;
}

// IEEE check of the two non-trivial f64 operations the path relies on (sqrt, divide) against the host
__global__ void k_selftest(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ out_sqrt,
                           double *__restrict__ out_div, unsigned long long n) {
    unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    out_sqrt[t] = sqrt(a[t]);
    out_div[t] = a[t] / b[t];
}

// Members of a porrt_grow_batch are prepared together: one grid row per context, everything taken from the run constants.
// k_batch_prep clears what a grow starts from (the host's two memsets) and derives the sampler's jump table from its increment
// (entry i advances the LCG 2^i steps: 64 squarings, one thread); k_init_root / k_gen_samples / k_sort_samples follow with the
// same rows.  Seven small operations per member become four launches per batch.
__global__ __launch_bounds__(256) void k_batch_prep(const RunConst *__restrict__ rcp) {
    const RunConst &rc = rcp[blockIdx.y];
    const unsigned long long tid = (unsigned long long)blockIdx.x * 256u + threadIdx.x, nth = (unsigned long long)gridDim.x * 256u;
    {
        const unsigned long long n4 = rc.zero_words / 4ull;            // zero0 is 16-byte aligned (arena granules)
        uint4 *z = reinterpret_cast<uint4 *>(rc.zero0);
        for (unsigned long long i = tid; i < n4; i += nth) z[i] = make_uint4(0u, 0u, 0u, 0u);
        for (unsigned long long i = 4ull * n4 + tid; i < rc.zero_words; i += nth) rc.zero0[i] = 0u;
    }
    for (unsigned long long i = tid; i < (unsigned long long)kRepInts; i += nth) rc.rep[i] = -1;        // (ids -1; the positions behind them NaN)
    {   // final flags of the nodes this grow may create (connect_rrt_sample only writes the set ones)
        unsigned long long *f8 = reinterpret_cast<unsigned long long *>(rc.final_flag);
        const unsigned long long n8 = ((unsigned long long)rc.sched_max_nodes + 7ull) / 8ull;
        for (unsigned long long i = tid; i < n8; i += nth) f8[i] = 0ull;
    }
    if (blockIdx.x == 1u || gridDim.x == 1u) {          // the context's own copy of its run constants (porrt_best_cost and the like)
        static_assert(sizeof(RunConst) % 8 == 0, "RunConst is copied in 8-byte words");
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&rc);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(rc.self);
        for (uint32_t i = threadIdx.x; i < sizeof(RunConst) / 8; i += 256u) dst[i] = src[i];
    }
    if (tid == 0) {
        u128 cm = mk128(0x4385DF649FCCF645ull, 0x2360ED051FC65DA4ull), cp = mk128(rc.rng_inc_lo, rc.rng_inc_hi);
        for (int b = 0; b < 64; ++b) {
            rc.jump->mult_lo[b] = (unsigned long long)cm; rc.jump->mult_hi[b] = (unsigned long long)(cm >> 64);
            rc.jump->plus_lo[b] = (unsigned long long)cp; rc.jump->plus_hi[b] = (unsigned long long)(cp >> 64);
            cp = (cm + 1) * cp;
            cm *= cm;
        }
    }
}

// from_rc: start and root validity from the run constants of the row (batch) instead of the arguments
__global__ void k_init_root(const RunConst *__restrict__ rcp, double x, double y, unsigned long long reach, int vid, int from_rc) {
    const RunConst &rc = rcp[blockIdx.y];
    if (from_rc) { x = rc.start_x; y = rc.start_y; reach = rc.root_reach; vid = rc.root_vid; }
    rc.nx[0] = x;
    rc.ny[0] = y;
    rc.parent[0] = -1;
    rc.distA[0] = 0.0;
    rc.distB[0] = 0.0;
    rc.reachA[0] = reach;
    rc.reachB[0] = reach;
    rc.vid[0] = (uint8_t)vid;
    rc.final_flag[0] = 0;
    rc.final_mask[0] = 0;
    rc.n_at[0] = 1;
    rc.t2_at[0] = 0.0;          // heuristic_radius(1) = 0 (common.rs:357-369: ln 1 = 0), whatever the parameters
    rep_insert(rc, x, y, 0);
    {   // region pages
        const uint32_t r = region_of(rc, x, y);
        rc.pg_xy[2 * ((size_t)r * kPage)] = x; rc.pg_xy[2 * ((size_t)r * kPage) + 1] = y;
        rc.pg_id[(size_t)r * kPage] = 0;
        rc.pg_d[(size_t)r * kPage] = 0.0;
        rc.slot_of[0] = r * kPage;
        rc.rg_cnt[r] = 1;
        for (uint32_t w = 0; w < kOccWords; ++w) rc.rg_occ[w] = w == r / 64u ? 1ull << (r % 64u) : 0ull;
    }
    rc.g_id[0] = 0;          // the root is on every kd descent path
    rc.cnt->g_len = 1;
    rc.cnt->g_first_dup[0] = 0xFFFFFFFFu;
    rc.cnt->g_first_dup[1] = 0xFFFFFFFFu;
    if (x == rc.gp_x && y == rc.gp_y) { rc.cnt->g_first_dup[0] = 0; rc.cnt->g_nd_len = 0; }
    else { rc.g_nd[0] = 0; rc.g_nd_x[0] = x; rc.g_nd_y[0] = y; rc.g_nd_box[0] = g_box_after(g_box_all(), x, y, 0u, rc.gp_x, rc.gp_y); rc.cnt->g_nd_len = 1; }
    KdRec rec;
    rec.x = x; rec.y = y; rec.child[0] = kEmpty; rec.child[1] = kEmpty;
    rc.kd_rec[0] = rec;
    rc.g_x[0] = x;
    rc.g_y[0] = y;
    rc.kd_up[0] = -1;
    {
        const double INF = __longlong_as_double(0x7FF0000000000000ll);
        KdBox bx;
        bx.lox = -INF; bx.hix = INF; bx.loy = -INF; bx.hiy = INF;
        rc.kd_box[0] = bx;
    }
    rc.kd_depth[0] = 0;
    rc.kd_gexit[0] = kOnG;
    rc.cnt->kd_done = 1;
    rc.cnt->kd_snap = 0;
    rc.g_snap[0] = 1; rc.g_snap[1] = rc.cnt->g_nd_len; rc.g_snap[2] = rc.cnt->g_first_dup[0]; rc.g_snap[3] = rc.cnt->g_first_dup[1];
}

__global__ void k_wait_us(uint32_t us) {
    const unsigned long long t0 = wall_clock64();          // 100 MHz
    while (wall_clock64() - t0 < 100ull * us) __builtin_amdgcn_s_sleep(8);
}
// One step.  Main stream: near (NN + steer + radius search), connect, commit (which also bounds the next step's
// samples).  Side stream (RRT*): order-exact kd insertion of this step's nodes, started as soon as their positions
// are final and needed only by the NEXT step's connect -- two cross-stream edges per step.
void porrt_ctx::launch_step(uint32_t b, uint32_t i0, uint32_t nb, uint32_t vwords, size_t lds_bytes, bool prof, size_t &ev_used,
                            uint32_t nxt_i0, uint32_t nxt_nb) {
    const uint32_t wave_blocks = (nb * 64 + 255) / 256;
    const RunConst *rcp = launch_rcp;
    const uint32_t Q = launch_Q;
    auto ev = [&](void) {
        if (prof && ev_used < ev_pool.size()) (void)hipEventRecord(ev_pool[ev_used++], stream);
    };
    const bool rrt = mode == PORRT_MODE_RRT;
    if (rrt && lag_on && opt_group == 0) {
        // one kernel per step: [k_near(b) unless its search rode in the step before]  X(b) = connect(b) | search(b + 1) | file(b) | commit(b - 1)
        ev();
        if (lag_near_done != b) {
            // (everything before step b is filed -- the start of a run, or after a join: the plain search over the pages)
            hipLaunchKernelGGL(k_near<false>, dim3(wave_blocks, Q), dim3(256), 0, stream, rcp, b, i0, nb, vwords, 0xFFFFFFFFu, 0u);
        }
        ev();
        (void)hipEventRecord(ev_steered, stream);          // positions and ids of step b are final
        ev();
        const uint32_t cb4 = (nb + kConnectWaves - 1) / kConnectWaves;
        const uint32_t cnb = commit_pend_b != 0xFFFFFFFFu ? commit_pend_nb : 0u;
        const bool gt_side = kd_lazy && opt_gtrack_side && !prof;
        if (gt_side) {
            // the goal path of step b's new nodes beside X(b): it needs what the kernel before X(b) left (recorded just above), and X(b) --
            // whose connect pass reads the exit levels of the nodes before step b -- needs the goal path of step b - 1
            if (gt_pend) (void)hipStreamWaitEvent(stream, ev_kd[gt_par], 0);
            (void)hipStreamWaitEvent(stream2, ev_steered, 0);
            hipLaunchKernelGGL(k_gtrack, dim3(1, Q), dim3(256), 0, stream2, rcp, b, nb, vwords);
            gt_par ^= 1u;
            (void)hipEventRecord(ev_kd[gt_par], stream2);
            gt_pend = true;
        }
        const uint32_t lazy = (kd_lazy && !gt_side) ? 1u : 0u;
        const dim3 xg(cb4 + (nxt_nb + 3) / 4 + 1 + lazy + (cnb + 3) / 4, Q);
        if (lds_bytes) hipLaunchKernelGGL(k_step1_rrt<true>, xg, dim3(256), lds_bytes, stream, rcp, b, nb, nxt_i0, nxt_nb, vwords, commit_pend_b, cnb, lazy);
        else hipLaunchKernelGGL(k_step1_rrt<false>, xg, dim3(256), 0, stream, rcp, b, nb, nxt_i0, nxt_nb, vwords, commit_pend_b, cnb, lazy);
        ev();
        lag_near_done = nxt_nb ? b + 1 : 0xFFFFFFFFu;
        commit_pend_b = b; commit_pend_nb = nb;
        kd_last_b = b; kd_last_nb = nb;
        side_active = true;
        if (!opt_kd_after && !kd_lazy && b + 1 - kd_b0 >= kd_group) launch_kd_group();
        return;
    }
    if (rrt && pipe_on && opt_group == 0) {
        // pipelined steps (k_step_rrt): [k_near(b), F(b) unless launched ahead]  S(b) = connect(b) + search(b + 1)  F(b + 1)
        const uint32_t cb4 = (nb + kConnectWaves - 1) / kConnectWaves;
        ev();
        if (pipe_near_done != b) {
            hipLaunchKernelGGL(k_near<false>, dim3(wave_blocks, Q), dim3(256), 0, stream, rcp, b, i0, nb, vwords, 0xFFFFFFFFu, 0u);
            const uint32_t cnb = commit_pend_b != 0xFFFFFFFFu ? commit_pend_nb : 0u;
            hipLaunchKernelGGL(k_file_commit, dim3(1 + (cnb + 15) / 16, Q), dim3(1024), 0, stream, rcp, b, nb, commit_pend_b, cnb, vwords);
            commit_pend_b = 0xFFFFFFFFu;
        }
        ev();
        (void)hipEventRecord(ev_steered, stream);          // positions and ids of step b are final
        ev();
        const dim3 sg(cb4 + (nxt_nb + 3) / 4, Q);
        if (lds_bytes) hipLaunchKernelGGL(k_step_rrt<true>, sg, dim3(256), lds_bytes, stream, rcp, b, nb, nxt_i0, nxt_nb, vwords);
        else hipLaunchKernelGGL(k_step_rrt<false>, sg, dim3(256), 0, stream, rcp, b, nb, nxt_i0, nxt_nb, vwords);
        ev();
        if (nxt_nb) {
            hipLaunchKernelGGL(k_file_commit, dim3(1 + (nb + 15) / 16, Q), dim3(1024), 0, stream, rcp, b + 1, nxt_nb, b, nb, vwords);
            pipe_near_done = b + 1;
        } else {
            commit_pend_b = b; commit_pend_nb = nb;
            pipe_near_done = 0xFFFFFFFFu;
        }
        kd_last_b = b; kd_last_nb = nb;
        side_active = true;
        if (!opt_kd_after && b + 1 - kd_b0 >= kd_group) launch_kd_group();
        return;
    }
    ev();
    const uint32_t GLn = rrt ? opt_group : 0u;
    // the connect pass of the first steps with one wave per sample (slots are slots whatever the group size)
    const uint32_t GLc = (GLn == 16u && b < opt_early_wave) ? 64u : GLn;
    if (mode == PORRT_MODE_PTO) hipLaunchKernelGGL(k_near<true>, dim3(wave_blocks, Q), dim3(256), 0, stream, rcp, b, i0, nb, vwords, 0xFFFFFFFFu, 0u);
    else if (GLn) {
        // GL lanes per sample; the previous step's rewire phase 2 rides along in extra workgroups
        const uint32_t spb = 256u / GLn, sblocks = (nb + spb - 1) / spb;
        const uint32_t cblocks = commit_pend_b != 0xFFFFFFFFu ? (commit_pend_nb + spb - 1) / spb : 0;
        const dim3 g(sblocks + cblocks, Q);
        const uint32_t cnb = cblocks ? commit_pend_nb : 0u;
        if (GLn == 16) hipLaunchKernelGGL(k_nn2<16>, g, dim3(256), 0, stream, rcp, b, i0, nb, vwords, commit_pend_b, cnb);
        else if (GLn == 32) hipLaunchKernelGGL(k_nn2<32>, g, dim3(256), 0, stream, rcp, b, i0, nb, vwords, commit_pend_b, cnb);
        else hipLaunchKernelGGL(k_nn2<64>, g, dim3(256), 0, stream, rcp, b, i0, nb, vwords, commit_pend_b, cnb);
        commit_pend_b = 0xFFFFFFFFu;
    } else {
        // the previous step's rewire phase 2 rides along in extra workgroups
        const uint32_t cblocks = commit_pend_b != 0xFFFFFFFFu ? (commit_pend_nb + 3) / 4 : 0;
        hipLaunchKernelGGL(k_near<false>, dim3(wave_blocks + cblocks, Q), dim3(256), 0, stream, rcp, b, i0, nb, vwords, commit_pend_b, cblocks ? commit_pend_nb : 0u);
        commit_pend_b = 0xFFFFFFFFu;
    }
    ev();
    // + 1: the workgroup that files the new nodes into the region pages
    const dim3 cgrid((nb + kConnectWaves - 1) / kConnectWaves + 1, Q), cblock(kConnectWaves * 64);
    if (!rrt) {
        ev();
        if (lds_bytes) hipLaunchKernelGGL(k_connect_pto<true>, cgrid, cblock, lds_bytes, stream, rcp, b, nb, vwords);
        else hipLaunchKernelGGL(k_connect_pto<false>, cgrid, cblock, 0, stream, rcp, b, nb, vwords);
        ev();
        hipLaunchKernelGGL(k_commit_pto, dim3(wave_blocks, Q), dim3(256), 0, stream, rcp, b, nb, vwords);
        return;
    }
    // RRT*: the kd structure that orders equal-cost parents is built beside the steps on a second stream, several
    // steps' nodes at a time, and is never waited for -- a tie that needs nodes it does not hold yet is deferred
    // (k_tie_fix).
    (void)hipEventRecord(ev_steered, stream);
    ev();
    if (GLn) {
        const uint32_t spb = 256u / GLc;
        const uint32_t lazy = kd_lazy ? 1u : 0u;
        const dim3 g2((nb + spb - 1) / spb + 2 + lazy, Q);       // + the clone workgroup + the page-filing workgroup (+ the goal path's)
        const size_t dyn = conn2_lds_bytes(GLc);
        if (GLc == 16) hipLaunchKernelGGL(k_conn2<16>, g2, dim3(256), dyn, stream, rcp, b, nb, vwords, lazy);
        else if (GLc == 32) hipLaunchKernelGGL(k_conn2<32>, g2, dim3(256), dyn, stream, rcp, b, nb, vwords, lazy);
        else hipLaunchKernelGGL(k_conn2<64>, g2, dim3(256), dyn, stream, rcp, b, nb, vwords, lazy);
    } else if (lds_bytes) hipLaunchKernelGGL(k_connect_rrt<true>, cgrid, cblock, lds_bytes, stream, rcp, b, nb, vwords);
    else hipLaunchKernelGGL(k_connect_rrt<false>, cgrid, cblock, 0, stream, rcp, b, nb, vwords);
    ev();
    commit_pend_b = b; commit_pend_nb = nb;
    kd_last_b = b; kd_last_nb = nb;
    side_active = true;
    if (!opt_kd_after && !kd_lazy && b + 1 - kd_b0 >= kd_group) launch_kd_group();
}

// k_kd_claim of one group: the form by the group's size, the row count and (developer option) kd_claim_threads
static void launch_kd_claim(hipStream_t st, const RunConst *rcp, uint32_t Q, uint32_t b0, uint32_t ns, uint32_t K, uint32_t vwords, uint32_t threads) {
    const uint64_t nodes = (uint64_t)ns * K;
    if (Q == 1 && threads == 0) {
        if (nodes <= 2048u) hipLaunchKernelGGL(k_kd_claim<2048>, dim3(1, 1), dim3(1024), 0, st, rcp, b0, ns, vwords);
        else hipLaunchKernelGGL(k_kd_claim<kClaimMax>, dim3(1, 1), dim3(1024), 0, st, rcp, b0, ns, vwords);
        return;
    }
    if (threads == 0) threads = 256u;
    if (nodes <= 1024u) {
        if (threads <= 256u) hipLaunchKernelGGL((k_kd_claim<1024, false, 256>), dim3(1, Q), dim3(256), 0, st, rcp, b0, ns, vwords);
        else if (threads <= 512u) hipLaunchKernelGGL((k_kd_claim<1024, false, 512>), dim3(1, Q), dim3(512), 0, st, rcp, b0, ns, vwords);
        else hipLaunchKernelGGL((k_kd_claim<1024, false, 1024>), dim3(1, Q), dim3(1024), 0, st, rcp, b0, ns, vwords);
    } else if (nodes <= 2048u) {
        if (threads <= 256u) hipLaunchKernelGGL((k_kd_claim<2048, false, 256>), dim3(1, Q), dim3(256), 0, st, rcp, b0, ns, vwords);
        else if (threads <= 512u) hipLaunchKernelGGL((k_kd_claim<2048, false, 512>), dim3(1, Q), dim3(512), 0, st, rcp, b0, ns, vwords);
        else hipLaunchKernelGGL((k_kd_claim<2048, false, 1024>), dim3(1, Q), dim3(1024), 0, st, rcp, b0, ns, vwords);
    } else {
        if (threads <= 512u) hipLaunchKernelGGL((k_kd_claim<kClaimMax, false, 512>), dim3(1, Q), dim3(512), 0, st, rcp, b0, ns, vwords);
        else hipLaunchKernelGGL((k_kd_claim<kClaimMax, false, 1024>), dim3(1, Q), dim3(1024), 0, st, rcp, b0, ns, vwords);
    }
}

// kd insertion of steps [kd_b0, kd_last_b] on the side stream (positions are final since the last k_near; the
// deferred ties are looked at again once the last commit is through)
void porrt_ctx::launch_kd_group() {
    if (kd_b0 > kd_last_b) return;
    const RunConst *rcp = launch_rcp;
    const uint32_t Q = launch_Q;
    const uint32_t nsteps = kd_last_b - kd_b0 + 1, K = rc.cand_K, vwords = (K + 63) / 64;
    if (opt_kd_inline && Q > 1) {
        // on the main stream, between two steps: stream order is all the synchronisation there is
        if ((uint64_t)nsteps * K * Q >= 4096u) hipLaunchKernelGGL(k_kd_locate<1>, dim3((nsteps * K + 255) / 256, Q), dim3(256), 0, stream, rcp, kd_b0, nsteps, K, kd_last_nb, vwords, 0u, 0u, 0u);
        else hipLaunchKernelGGL(k_kd_locate<64>, dim3((nsteps * K * 64 + 255) / 256, Q), dim3(256), 0, stream, rcp, kd_b0, nsteps, K, kd_last_nb, vwords, 0u, 0u, 0u);
        hipLaunchKernelGGL(k_kd_link, dim3((nsteps * K + 255) / 256, Q), dim3(256), 0, stream, rcp, kd_b0, nsteps, vwords, 0u);
        launch_kd_claim(stream, rcp, Q, kd_b0, nsteps, K, vwords, opt_claim_threads);
        hipLaunchKernelGGL(k_kd_hint_fix, dim3((nsteps * K + 255) / 256 + kTieParts, Q), dim3(256), 0, stream, rcp, kd_b0, nsteps, vwords);
        ++kd_gidx;
        kd_b0 = kd_last_b + 1;
        return;
    }
    (void)hipStreamWaitEvent(stream2, ev_steered, 0);
    // few nodes: one wave per node (latency); many (several contexts at once): one thread per node (wave slots)
    // the hints of the group before and the deferred ties ride in this group's locate kernel (extra workgroups per row)
    const bool ride = opt_kd_ride != 0 || Q == 1;
    const uint32_t extra = (ride && kd_hint_ns) ? (kd_hint_ns * K + 255) / 256 + kTieParts : 0;
    const uint32_t hb0 = extra ? kd_hint_b0 : 0u, hns = extra ? kd_hint_ns : 0u;
    // (a thread per node from 4096 nodes on: a wave per node is no faster there and takes the step kernels' wave slots)
    if ((uint64_t)nsteps * K * Q >= 4096u) hipLaunchKernelGGL(k_kd_locate<1>, dim3((nsteps * K + 255) / 256 + extra, Q), dim3(256), 0, stream2, rcp, kd_b0, nsteps, K, kd_last_nb, vwords, 0u, hb0, hns);
    else hipLaunchKernelGGL(k_kd_locate<64>, dim3((nsteps * K * 64 + 255) / 256 + extra, Q), dim3(256), 0, stream2, rcp, kd_b0, nsteps, K, kd_last_nb, vwords, 0u, hb0, hns);
    hipLaunchKernelGGL(k_kd_link, dim3((nsteps * K + 255) / 256, Q), dim3(256), 0, stream2, rcp, kd_b0, nsteps, vwords, 0u);
    if (Q == 1) hipLaunchKernelGGL((k_kd_claim<kClaimMax, true>), dim3(1, 1), dim3(1024), 0, stream2, rcp, kd_b0, nsteps, vwords);
    else launch_kd_claim(stream2, rcp, Q, kd_b0, nsteps, K, vwords, opt_claim_threads);
    if (ride) {
        // (a record's placeholder parent is in place before the record can be seen: nothing to wait for on the main stream);
        // join_side launches the last group's
        kd_hint_b0 = kd_b0; kd_hint_ns = nsteps;
    } else {
        hipLaunchKernelGGL(k_kd_hint, dim3((nsteps * K + 255) / 256, Q), dim3(256), 0, stream2, rcp, kd_b0, nsteps, vwords);
        (void)hipEventRecord(ev_step_done, stream);
        (void)hipStreamWaitEvent(stream2, ev_step_done, 0);
        hipLaunchKernelGGL(k_tie_fix<256>, dim3(1, Q), dim3(256), 0, stream2, rcp);
    }
    // A lagging join: the main stream waits for the group BEFORE this one, which had a whole group of steps to finish.
    // It bounds how far the kd structure may fall behind and keeps a replayed hipGraph from running the side branch last.
    // (a single query waits for the group before THAT: its first groups -- thousands of nodes into a tree of a few -- take several
    // times as long as four steps, and the ties they leave deferred are few)
    const uint32_t ring = Q == 1 ? 3u : 2u;
    const uint32_t par = kd_gidx % ring, oldest = (kd_gidx + 1u) % ring;
    (void)hipEventRecord(ev_kd[par], stream2);
    kd_pend[par] = true;
    if (kd_pend[oldest]) { (void)hipStreamWaitEvent(stream, ev_kd[oldest], 0); kd_pend[oldest] = false; }
    ++kd_gidx;
    kd_b0 = kd_last_b + 1;
}

// steps whose new nodes are inserted into the kd structure together: the kd kernels' run time is set by the deepest
// descent, not by the node count, so a single context takes as many steps as the claim kernel holds (kClaimMax
// nodes: 7 % faster than half of that); with several contexts per launch the kernels are throughput bound and smaller
// groups keep the structure (and the deferred ties) closer behind the steps.
static uint32_t kd_group_for(uint32_t K, uint32_t opt, uint32_t Q) {
    const uint32_t cap = std::max<uint32_t>(1u, std::min<uint32_t>(8u, kClaimMax / K));
    uint32_t g = std::max<uint32_t>(1u, std::min<uint32_t>(8u, (Q > 1 ? 2048u : kClaimMax) / K));
    if (opt) g = std::min(opt, cap);
    return g;
}

// join the kd streams back into the main stream (end of a launch sequence / of a capture) and settle what is
// left of the deferred ties
// the last launched step's rewire phase 2, stand-alone
void porrt_ctx::flush_commit() {
    if (commit_pend_b == 0xFFFFFFFFu) return;
    const uint32_t vwords = (rc.cand_K + 63) / 64;
    if (opt_group == 16) hipLaunchKernelGGL(k_commit2<16>, dim3((commit_pend_nb * 16 + 255) / 256, launch_Q), dim3(256), 0, stream, launch_rcp, commit_pend_b, commit_pend_nb, vwords);
    else if (opt_group == 32) hipLaunchKernelGGL(k_commit2<32>, dim3((commit_pend_nb * 32 + 255) / 256, launch_Q), dim3(256), 0, stream, launch_rcp, commit_pend_b, commit_pend_nb, vwords);
    else hipLaunchKernelGGL(k_commit_rrt, dim3((commit_pend_nb * 64 + 255) / 256, launch_Q), dim3(256), 0, stream, launch_rcp, commit_pend_b, commit_pend_nb, vwords, lag_on ? 1u : 0u);
    commit_pend_b = 0xFFFFFFFFu;
}

// The persistent step loop (option pipeline = 2): every step of a single query's grow in one cooperative launch on the main
// stream, and -- queued behind it on the side stream, each group's first kernel waiting for the steps it inserts (coop_filed) --
// the kd groups.  Returns PORRT_OK, or a code without having launched anything (the caller then takes the launches step by step).
int porrt_ctx::launch_coop(uint32_t n_steps, uint32_t K, uint64_t n_iter, uint32_t vwords, size_t lds_bytes) {
    const void *fn = lds_bytes ? reinterpret_cast<const void *>(&k_coop_rrt<true>) : reinterpret_cast<const void *>(&k_coop_rrt<false>);
    if (!coop_blocks) {
        int occ = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, kConnectWaves * 64, lds_bytes) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || occ < 1 || cus < 1) { (void)hipGetLastError(); return PORRT_ERR_DEVICE; }
        coop_blocks = cus * std::min(occ, 2);         // two workgroups per CU hold a step's 512 work items of K = 1024 in one round
    }
    // the side stream may not start before this grow's counters are cleared (it polls one of them)
    (void)hipEventRecord(ev_steered, stream);
    (void)hipStreamWaitEvent(stream2, ev_steered, 0);
    const RunConst *rcp = launch_rcp;
    uint32_t n_iter32 = (uint32_t)n_iter;
    void *args[] = {(void *)&rcp, (void *)&n_steps, (void *)&K, (void *)&n_iter32, (void *)&vwords};
    // 2: a cooperative launch (the runtime refuses a grid that is not resident at once); 3: the same grid as an ordinary launch
    // (developer switch: the grid is sized from the occupancy query either way, and the barrier gives up rather than hang)
    const hipError_t le = opt_pipeline == 3 ? hipLaunchKernel(fn, dim3((unsigned)coop_blocks), dim3(kConnectWaves * 64), args, lds_bytes, stream)
                                            : hipLaunchCooperativeKernel(fn, dim3((unsigned)coop_blocks), dim3(kConnectWaves * 64), args, (unsigned)lds_bytes, stream);
    if (le != hipSuccess) {
        (void)hipGetLastError();
        return PORRT_ERR_DEVICE;
    }
    for (uint32_t b0 = 0; b0 < n_steps; b0 += kd_group) {
        const uint32_t ns = std::min<uint32_t>(kd_group, n_steps - b0), last = b0 + ns - 1u;
        const uint32_t nb_last = (uint32_t)std::min<uint64_t>(K, n_iter - (uint64_t)last * K);
        if ((uint64_t)ns * K >= 4096u) hipLaunchKernelGGL(k_kd_locate<1>, dim3((ns * K + 255) / 256, 1), dim3(256), 0, stream2, rcp, b0, ns, K, nb_last, vwords, kWaitFiled);
        else hipLaunchKernelGGL(k_kd_locate<64>, dim3((ns * K * 64 + 255) / 256, 1), dim3(256), 0, stream2, rcp, b0, ns, K, nb_last, vwords, kWaitFiled);
        hipLaunchKernelGGL(k_kd_link, dim3((ns * K + 255) / 256, 1), dim3(256), 0, stream2, rcp, b0, ns, vwords, 0u);
        if ((uint64_t)ns * K <= 2048u) hipLaunchKernelGGL(k_kd_claim<2048>, dim3(1, 1), dim3(1024), 0, stream2, rcp, b0, ns, vwords);
        else hipLaunchKernelGGL(k_kd_claim<kClaimMax>, dim3(1, 1), dim3(1024), 0, stream2, rcp, b0, ns, vwords);
        hipLaunchKernelGGL(k_kd_hint, dim3((ns * K + 255) / 256, 1), dim3(256), 0, stream2, rcp, b0, ns, vwords);
        hipLaunchKernelGGL(k_tie_fix<1024>, dim3(1, 1), dim3(1024), 0, stream2, rcp);
    }
    (void)hipEventRecord(ev_join, stream2);
    (void)hipStreamWaitEvent(stream, ev_join, 0);
    hipLaunchKernelGGL(k_tie_fix<1024>, dim3(1, 1), dim3(1024), 0, stream, rcp);
    kd_pend[0] = kd_pend[1] = kd_pend[2] = false;
    side_active = false;
    commit_pend_b = 0xFFFFFFFFu;
    pipe_near_done = 0xFFFFFFFFu;
    return PORRT_OK;
}

void porrt_ctx::join_side() {
    flush_commit();
    if (!side_active) return;
    if (kd_lazy && !kd_build_now) {        // (the goal path rode in the step kernels, or beside them in kernels of its own: the last of those is waited for)
        if (gt_pend) { (void)hipStreamWaitEvent(stream, ev_kd[gt_par], 0); gt_pend = false; }
        side_active = false;
        return;
    }
    if (opt_kd_after || kd_build_now) {
        // the whole structure now, on the main stream: groups of as many steps as the claim kernel holds, in id order
        const RunConst *rcp = launch_rcp;
        const uint32_t Q = launch_Q, K = rc.cand_K, vwords = (K + 63) / 64;
        const uint32_t g = std::max<uint32_t>(1u, std::min<uint32_t>(8u, kClaimMax / K));
        while (kd_b0 <= kd_last_b) {
            const uint32_t ns = std::min<uint32_t>(g, kd_last_b - kd_b0 + 1), last = kd_b0 + ns - 1;
            const uint32_t nbl = last == kd_last_b ? kd_last_nb : K;
            if ((uint64_t)ns * K * Q >= 4096u) hipLaunchKernelGGL(k_kd_locate<1>, dim3((ns * K + 255) / 256, Q), dim3(256), 0, stream, rcp, kd_b0, ns, K, nbl, vwords, 0u);
            else hipLaunchKernelGGL(k_kd_locate<64>, dim3((ns * K * 64 + 255) / 256, Q), dim3(256), 0, stream, rcp, kd_b0, ns, K, nbl, vwords, 0u);
            hipLaunchKernelGGL(k_kd_link, dim3((ns * K + 255) / 256, Q), dim3(256), 0, stream, rcp, kd_b0, ns, vwords, 0u);
            launch_kd_claim(stream, rcp, Q, kd_b0, ns, K, vwords, opt_claim_threads);
            hipLaunchKernelGGL(k_kd_hint, dim3((ns * K + 255) / 256, Q), dim3(256), 0, stream, rcp, kd_b0, ns, vwords);
            kd_b0 = last + 1;
        }
        if (Q > 1) hipLaunchKernelGGL(k_tie_fix<256>, dim3(1, Q), dim3(256), 0, stream, rcp);
        else hipLaunchKernelGGL(k_tie_fix<1024>, dim3(1, Q), dim3(1024), 0, stream, rcp);
        side_active = false;
        return;
    }
    launch_kd_group();
    if (opt_kd_inline && launch_Q > 1) {
        hipLaunchKernelGGL(k_tie_fix<256>, dim3(1, launch_Q), dim3(256), 0, stream, launch_rcp);
        kd_pend[0] = kd_pend[1] = kd_pend[2] = false;
        side_active = false;
        return;
    }
    if (kd_hint_ns) {
        hipLaunchKernelGGL(k_kd_hint_fix, dim3((kd_hint_ns * rc.cand_K + 255) / 256 + kTieParts, launch_Q), dim3(256), 0, stream2, launch_rcp, kd_hint_b0, kd_hint_ns, (rc.cand_K + 63) / 64);
        kd_hint_ns = 0;
    }
    (void)hipEventRecord(ev_join, stream2);
    (void)hipStreamWaitEvent(stream, ev_join, 0);
    kd_pend[0] = kd_pend[1] = kd_pend[2] = false;
    if (launch_Q > 1) hipLaunchKernelGGL(k_tie_fix<256>, dim3(1, launch_Q), dim3(256), 0, stream, launch_rcp);
    else hipLaunchKernelGGL(k_tie_fix<1024>, dim3(1, launch_Q), dim3(1024), 0, stream, launch_rcp);
    side_active = false;
}

// kd_lazy, after the steps: a tie needed more than the goal path (Counters::n_lca).  The kd state goes back to the grow's start and
// the whole structure is built in id order on the main stream (what option kd_after does after every run); k_tie_fix then settles
// the records that waited for it.
int porrt_ctx::kd_full_build(const std::vector<std::pair<uint32_t, uint32_t>> &segs) {
    // segs: (last step, its sample count) of every run of steps in which only the last one may be short (the steps up to
    // n_iter_min, then the steps of the loop's tail)
    hipLaunchKernelGGL(k_kd_reset, dim3(64, launch_Q), dim3(256), 0, stream, launch_rcp);
    kd_b0 = 0; kd_gidx = 0; kd_hint_ns = 0;
    commit_pend_b = 0xFFFFFFFFu;
    kd_build_now = true;
    for (const auto &sg : segs) {
        if (sg.first + 1u <= kd_b0) continue;
        kd_last_b = sg.first; kd_last_nb = sg.second;
        side_active = true;
        join_side();
    }
    kd_build_now = false;
    return hipGetLastError() == hipSuccess ? PORRT_OK : PORRT_ERR_DEVICE;
}

int porrt_ctx::grow(const double start[2], double max_step, double search_radius, uint64_t n_iter_min, uint64_t n_iter_max,
                    uint32_t K, int mode_) {
    if (K == 0 || K > 4096) { set_err("batch_K must be in 1..4096"); return PORRT_ERR_INVALID; }
    if (mode_ != PORRT_MODE_RRT && mode_ != PORRT_MODE_PTO) { set_err("bad mode"); return PORRT_ERR_INVALID; }
    if (mode_ == PORRT_MODE_PTO && !has_grid) { set_err("PTO mode needs a grid"); return PORRT_ERR_INVALID; }
    if (n_iter_max < n_iter_min) n_iter_max = n_iter_min;
    if (n_iter_max + 2 >= 0x7FFFFFF0ull) { set_err("n_iter_max too large"); return PORRT_ERR_INVALID; }
    if (!(max_step > 0.0) || !(search_radius >= 0.0)) { set_err("max_step / search_radius"); return PORRT_ERR_INVALID; }
    have_results = false;
    batch_leader = nullptr;
    bg.release();
    dp.release();
    ++results_tag;
    // the sampler state must survive a capacity retry
    const Pcg64 c0 = crng, d0 = drng;
    const size_t ip0 = inj_pos, iw0 = inj_wpos;
    bool host_samples = false;
    for (int attempt = 0; attempt < 12; ++attempt) {
        int rc_ = grow_once(start, max_step, search_radius, n_iter_min, n_iter_max, K, mode_, host_samples);
        if (rc_ == -100) {          // neighbour list overflow: regrow the lists and replay
            opt_cand_cap = (uint32_t)std::min<uint64_t>((uint64_t)opt_cand_cap * 4, n_iter_max + 2);
        } else if (rc_ == -101) {   // a float draw would have retried: replay with the exact host stream
            host_samples = true;
        } else if (rc_ == -102) {   // edge pool (PTO: the reference's parameters reach ~240 neighbours per node): regrow and replay
            edge_per_node *= 4;
        } else if (rc_ == -103) {   // deferred-tie pool
            tie_pool_mult *= 4;
        } else {
            return rc_;
        }
        crng = c0; drng = d0; inj_pos = ip0; inj_wpos = iw0;
    }
    set_err("neighbour list capacity");
    return PORRT_ERR_CAPACITY;
}

// stage 0: the whole call.  stage 1 (member of a porrt_grow_batch): everything up to the growth loop -- buffers,
// run constants, root, sample stream -- and return; the batch leader runs the steps for all members.
int porrt_ctx::grow_once(const double start[2], double max_step, double search_radius, uint64_t n_iter_min, uint64_t n_iter_max,
                         uint32_t K, int mode_, bool host_samples, int stage) {
    const double t_begin = now_s();
    double t_setup = 0.0;
    HIPCHK(hipSetDevice(device));
    mode = mode_;
    memset(&metrics, 0, sizeof metrics);

    // ---- capacities
    const uint64_t Nmax = n_iter_max + 2;
    const uint64_t steps_max = n_iter_max / K + 4;
    const uint32_t vwords = (K + 63) / 64;
    const uint32_t Kpad = vwords * 64;       // sample stride of the per-chunk arrays (multiple of the wave size)
    const uint32_t cand_cap = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(opt_cand_cap, 64), Nmax);
    // deferred equal-cost parents: at most one record per iteration; the pooled ids are bounded by experience
    const uint64_t pend_cap = n_iter_max + 2, pool_cap = std::max<uint64_t>(1u << 20, tie_pool_mult * n_iter_max);
    {
        double t0 = now_s();
        HIPCHK(d_nx.reserve(Nmax)); HIPCHK(d_ny.reserve(Nmax)); HIPCHK(d_distA.reserve(Nmax)); HIPCHK(d_distB.reserve(Nmax));
        HIPCHK(d_parent.reserve(Nmax)); HIPCHK(d_reachA.reserve(Nmax)); HIPCHK(d_reachB.reserve(Nmax));
        HIPCHK(d_vid.reserve(Nmax)); HIPCHK(d_finalflag.reserve(Nmax + 8)); HIPCHK(d_finalmask.reserve(Nmax));
        HIPCHK(d_nat.reserve(steps_max + 2)); HIPCHK(d_validmask.reserve((steps_max + 2) * vwords));
        HIPCHK(d_sx.reserve(n_iter_max + 1)); HIPCHK(d_sy.reserve(n_iter_max + 1)); HIPCHK(d_sworld.reserve(n_iter_max + 1));
        HIPCHK(d_qx.reserve(2 * Kpad)); HIPCHK(d_qy.reserve(2 * Kpad)); HIPCHK(d_qnn.reserve(2 * Kpad)); HIPCHK(d_qvid.reserve(2 * Kpad));      // two halves: pipelined steps (q_stride)
        // region pages: one static page per region + a pool that cannot run out (sum of ceil(n_r / 64) <= N / 64 + regions)
        const uint64_t rg_maxp = Nmax / kPage + 2, pg_cap = 2ull * kRegions + Nmax / kPage + 8;
        HIPCHK(d_rgcnt.reserve(2 * kRegions)); HIPCHK(d_rgocc.reserve(2 * kOccWords)); HIPCHK(d_rgdir.reserve((size_t)kRegions * rg_maxp));
        HIPCHK(d_pgxy.reserve(2 * (size_t)pg_cap * kPage)); HIPCHK(d_pgid.reserve((size_t)pg_cap * kPage)); HIPCHK(d_pgd.reserve((size_t)pg_cap * kPage)); HIPCHK(d_slotof.reserve(Nmax));
        HIPCHK(d_candcnt.reserve(3 * (size_t)K)); HIPCHK(d_kdbox.reserve(Nmax)); HIPCHK(d_kdlosers.reserve(kClaimMax)); HIPCHK(d_bcscratch.reserve(8 * Nmax + 4096)); HIPCHK(d_bcout.reserve(1)); HIPCHK(d_bccursor.reserve(1)); HIPCHK(d_locbox.reserve(2 * (8 * (size_t)K + 4096))); HIPCHK(d_kdhint.reserve((size_t)kHG * kHG));
        HIPCHK(d_loccur.reserve(2 * (8 * (size_t)K + 4096))); HIPCHK(d_locdcur.reserve(2 * (8 * (size_t)K + 4096))); HIPCHK(d_locgex.reserve(2 * (8 * (size_t)K + 4096))); HIPCHK(d_locflags.reserve(2 * (8 * (size_t)K + 4096)));
        HIPCHK(d_gsnap.reserve((steps_max + 4) * 4)); HIPCHK(d_pendoff.reserve(pend_cap)); HIPCHK(d_pendn.reserve(pend_cap)); HIPCHK(d_pendcur.reserve(pend_cap));
        HIPCHK(d_pendstate.reserve(pend_cap)); HIPCHK(d_pendnew.reserve(pend_cap)); HIPCHK(d_pendpool.reserve(pool_cap)); HIPCHK(d_kdsurv.reserve(Nmax)); HIPCHK(d_gndx.reserve(Nmax)); HIPCHK(d_gndy.reserve(Nmax)); HIPCHK(d_gndbox.reserve(Nmax));
        HIPCHK(d_kqx.reserve((steps_max + 2) * Kpad)); HIPCHK(d_kqy.reserve((steps_max + 2) * Kpad)); HIPCHK(d_kqvid.reserve((steps_max + 2) * Kpad)); { const size_t np = (opt_pipeline == 4 && stage != 1 && K <= 1024) ? 3u : 2u; HIPCHK(d_candid.reserve(np * (size_t)K * cand_cap)); HIPCHK(d_candxy.reserve(2 * np * (size_t)K * cand_cap)); HIPCHK(d_candval.reserve((np == 3u ? 3u : 1u) * (size_t)K * cand_cap)); }
        HIPCHK(d_gid.reserve(Nmax));
        HIPCHK(d_perm.reserve((steps_max + 2) * Kpad)); HIPCHK(d_ssx.reserve((steps_max + 2) * Kpad)); HIPCHK(d_ssy.reserve((steps_max + 2) * Kpad));
        HIPCHK(d_bqx.reserve(Kpad)); HIPCHK(d_bqy.reserve(Kpad)); HIPCHK(d_bqk.reserve(Kpad)); HIPCHK(d_t2at.reserve(steps_max + 4));
        HIPCHK(d_rep.reserve(kRepInts));
        HIPCHK(d_sched_i0.reserve(steps_max + 4)); HIPCHK(d_sched_nb.reserve(steps_max + 4));
        HIPCHK(d_kdrec.reserve(Nmax)); HIPCHK(d_gx.reserve(Nmax + 16)); HIPCHK(d_gy.reserve(Nmax + 16)); HIPCHK(d_kdup.reserve(Nmax)); HIPCHK(d_kddepth.reserve(Nmax)); HIPCHK(d_kdgexit.reserve(Nmax));
        HIPCHK(d_radT2.reserve(Nmax + 8));
        HIPCHK(d_cnt.reserve(1)); HIPCHK(d_rc.reserve(1)); HIPCHK(d_jump.reserve(1));
        if (mode == PORRT_MODE_PTO) {
            const uint64_t ecap = std::min<uint64_t>(Nmax * edge_per_node + 4096, 1ull << 31);
            HIPCHK(d_efrom.reserve(ecap)); HIPCHK(d_eto.reserve(ecap)); HIPCHK(d_etv.reserve(ecap));
        }
        if (has_grid) HIPCHK(d_cls.reserve(cls_bytes(W, H)));
        if (has_inj) HIPCHK(d_inj.reserve(inj_xy.size() + 2));
        int r = layout_buffers();
        if (r) return r;
        r = build_cls();
        if (r) return r;
        if (has_inj && inj_dirty) {
            HIPCHK(hipMemcpyAsync(d_inj.p, inj_xy.data(), inj_xy.size() * sizeof(double), hipMemcpyHostToDevice, stream));
            inj_dirty = false;
        }
        t_setup += now_s() - t0;
    }

    // ---- run constants
    RunConst &c = rc;
    memset(&c, 0, sizeof c);
    c.nx = d_nx.p; c.ny = d_ny.p; c.distA = d_distA.p; c.distB = d_distB.p; c.parent = d_parent.p;
    c.reachA = d_reachA.p; c.reachB = d_reachB.p; c.vid = d_vid.p; c.final_flag = d_finalflag.p; c.final_mask = d_finalmask.p;
    c.n_at = d_nat.p; c.valid_mask = d_validmask.p; c.cnt = d_cnt.p;
    c.sx = d_sx.p; c.sy = d_sy.p; c.sworld = d_sworld.p;
    c.inj_xy = (has_inj && !host_samples) ? d_inj.p : nullptr;
    c.inj_base = inj_pos; c.inj_n = inj_xy.size() / 2;
    c.q_x = d_qx.p; c.q_y = d_qy.p; c.q_nn = d_qnn.p; c.q_vid = d_qvid.p;
    c.rg_cnt = d_rgcnt.p; c.rg_occ = d_rgocc.p; c.rg_dir = d_rgdir.p; c.pg_xy = d_pgxy.p; c.pg_id = d_pgid.p; c.pg_d = d_pgd.p; c.slot_of = d_slotof.p;
    c.rg_maxp = (uint32_t)(Nmax / kPage + 2); c.pg_cap = (uint32_t)(2ull * kRegions + Nmax / kPage + 8);
    c.loc_cur = d_loccur.p; c.loc_dcur = d_locdcur.p; c.loc_gex = d_locgex.p; c.loc_flags = d_locflags.p; c.g_nd = d_kdsurv.p; c.g_nd_x = d_gndx.p; c.g_nd_y = d_gndy.p; c.g_nd_box = d_gndbox.p; c.kq_x = d_kqx.p; c.kq_y = d_kqy.p; c.kq_vid = d_kqvid.p; c.kd_box = d_kdbox.p; c.loc_box = d_locbox.p; c.kd_losers = d_kdlosers.p; c.bc_scratch = d_bcscratch.p; c.bc_cap = (uint32_t)std::min<size_t>(d_bcscratch.n, 0xFFFFFFFFu); c.bc_cursor = d_bccursor.p; c.bc_out = d_bcout.p; c.kd_hint = d_kdhint.p; c.g_snap = d_gsnap.p; c.loc_stride = 8 * K + 4096;
    c.pend_new = d_pendnew.p; c.pend_pool = d_pendpool.p; c.pend_off = d_pendoff.p; c.pend_n = d_pendn.p; c.pend_cur = d_pendcur.p; c.pend_state = d_pendstate.p;
    c.pend_cap = (uint32_t)std::min<uint64_t>(pend_cap, 0xFFFFFFFFull); c.pool_cap = (uint32_t)std::min<uint64_t>(pool_cap, 0xFFFFFFFFull);
    c.cand_K = K; c.cand_cnt = d_candcnt.p; c.cand_id = d_candid.p; c.cand_xy = d_candxy.p; c.cand_val = d_candval.p; c.cand_cap = cand_cap;
    c.rad_T2 = d_radT2.p;
    c.e_from = d_efrom.p; c.e_to = d_eto.p; c.e_tv = d_etv.p; c.e_cap = (uint32_t)d_efrom.n;
    c.rep = d_rep.p; c.rep_f = reinterpret_cast<flt2 *>(d_rep.p + kRepTotal);
    {
        // box of the bound pyramid and of the region grid (points outside it fall into the border cells)
        double x0 = std::min(s_low[0], start[0]), x1 = std::max(s_up[0], start[0]);
        double y0 = std::min(s_low[1], start[1]), y1 = std::max(s_up[1], start[1]);
        if (has_inj) for (size_t t = 0; t + 1 < inj_xy.size(); t += 2) { x0 = std::min(x0, inj_xy[t]); x1 = std::max(x1, inj_xy[t]); y0 = std::min(y0, inj_xy[t + 1]); y1 = std::max(y1, inj_xy[t + 1]); }
        for (uint32_t g = 0; g < G; ++g) { x0 = std::min(x0, gcx[g]); x1 = std::max(x1, gcx[g]); y0 = std::min(y0, gcy[g]); y1 = std::max(y1, gcy[g]); }
        for (int z = 0; z < n_zones; ++z) { x0 = std::min(x0, zone_pos[z][0]); x1 = std::max(x1, zone_pos[z][0]); y0 = std::min(y0, zone_pos[z][1]); y1 = std::max(y1, zone_pos[z][1]); }
        c.bx0 = x0; c.by0 = y0; c.binv_w = 1.0 / (x1 - x0); c.binv_h = 1.0 / (y1 - y0);
    }
    c.kd_rec = d_kdrec.p; c.g_x = d_gx.p; c.g_y = d_gy.p; c.kd_up = d_kdup.p; c.kd_depth = d_kddepth.p; c.kd_gexit = d_kdgexit.p;
    c.g_id = d_gid.p; c.g_cap = (uint32_t)std::min<uint64_t>(d_gid.n, 0xFFFFFFFFull);
    c.cls = d_cls.p; c.clr = d_cls.p + (size_t)W * H; c.sat = opt_box_table ? (const uint32_t *)(d_cls.p + cls_sat_offset(W, H)) : nullptr; c.W = W; c.H = H; c.low0 = low[0]; c.low1 = low[1]; c.ppm = ppm; c.domain = domain; c.has_grid = has_grid;
    c.n_validities = n_validities;
    for (int i = 0; i < n_validities; ++i) c.validities[i] = validities[i];
    c.all_worlds = ones(n_worlds);
    c.goal_kind = goal_kind; c.G = G; c.g_l1 = g_l1;
    for (uint32_t g = 0; g < G; ++g) { c.gcx[g] = gcx[g]; c.gcy[g] = gcy[g]; c.gmask[g] = gmask[g]; }
    for (int w = 0; w < 64; ++w) { c.w2g_x[w] = w2g[w][0]; c.w2g_y[w] = w2g[w][1]; }
    if (goal_kind == 2) { c.zone_x = zone_pos[obs_zone][0]; c.zone_y = zone_pos[obs_zone][1]; }
    c.visibility = visibility;
    // the point every 100th RRT* iteration re-adds (rrt.rs:176-181: goal_example(0))
    if (goal_kind == 1) { c.gp_x = w2g[0][0]; c.gp_y = w2g[0][1]; }
    else if (goal_kind == 2) { c.gp_x = c.zone_x; c.gp_y = c.zone_y; }
    c.s_low0 = s_low[0]; c.s_low1 = s_low[1]; c.s_up0 = s_up[0]; c.s_up1 = s_up[1];
    c.max_step = max_step; c.mode = mode;
    c.part_stride = Kpad;
    c.sched_max_nodes = (uint32_t)Nmax;
    // pipelined steps only with the one-wave-per-sample kernels (the group kernels keep q_* in one half); a batch member's
    // row is set by the leader, who knows which kernels will run
    pipe_on = stage != 1 && mode == PORRT_MODE_RRT && opt_pipeline != 0 && (opt_group_req < 0 ? 0 : opt_group_req) == 0;
    lag_on = pipe_on && opt_pipeline == 4 && K <= 1024;
    c.q_stride = pipe_on ? (uint32_t)Kpad : 0u;
    c.cand_par3 = lag_on ? 1u : 0u;
    c.hint_max = 0xFFFFFFFFu;              // (porrt_grow_batch lowers it for its rows)
    lag_near_done = 0xFFFFFFFFu;
    c.perm = d_perm.p; c.ssx = d_ssx.p; c.ssy = d_ssy.p; c.bq_x = d_bqx.p; c.bq_y = d_bqy.p; c.bq_k = d_bqk.p; c.t2_at = d_t2at.p;

    // ---- root (rrt.rs:105-106 / pto.rs:61-64)
    uint64_t root_reach = 0;
    int root_vid = 0;
    if (mode == PORRT_MODE_PTO) {
        // state_validity(start) on the host raster (same arithmetic as the device)
        volatile double ty = (start[1] - low[1]) * ppm;
        double fi = (double)(H - 1) - ty;
        volatile double fj = (start[0] - low[0]) * ppm;
        auto as_u32 = [](double v) -> uint64_t { if (!(v == v) || v <= 0.0) return 0; if (v >= 4294967295.0) return 4294967295ull; return (uint64_t)v; };
        uint64_t i = as_u32(fi), j = as_u32(fj);
        if (i >= H || j >= W) { set_err("start outside the map"); return PORRT_ERR_RASTER; }
        uint8_t cc = cls[i * W + j];
        if (cc == CLS_BAD) { set_err("door pixel without zone id at the start"); return PORRT_ERR_RASTER; }
        if (cc == CLS_FREE) root_vid = n_validities - 1;
        else if (cc >= CLS_ZONE) root_vid = cc - CLS_ZONE;
        else { set_err("Start from a valid state!"); return PORRT_ERR_INVALID_START; }
        root_reach = validities[root_vid];
    }

    // LDS tile per wave for the raycasts: every neighbour lies within radius <= max_step of the new node
    size_t lds_bytes = 0;
    {
        double rpx = max_step * ppm;
        if (has_grid && rpx < (double)(kTileRMax - 2)) {
            c.tile_R = (uint32_t)ceil(rpx) + 2;
            const uint32_t TW = 2 * c.tile_R + 1;
            lds_bytes = (size_t)kConnectWaves * ((TW * TW + 15u) & ~15u);
        }
    }

    // ---- sampler jump tables for this grow
    PcgJump jt;
    {
        u128 cm = PCG_MULT, cp = crng.inc;
        for (int b = 0; b < 64; ++b) {
            jt.mult_lo[b] = (uint64_t)cm; jt.mult_hi[b] = (uint64_t)(cm >> 64);
            jt.plus_lo[b] = (uint64_t)cp; jt.plus_hi[b] = (uint64_t)(cp >> 64);
            cp = (cm + 1) * cp;
            cm *= cm;
        }
    }
    const Pcg64 crng0 = crng;
    // a member of a porrt_grow_batch (RRT*): the leader prepares all members at once on the device (k_batch_prep ...)
    const bool batch_prep = stage == 1 && mode == PORRT_MODE_RRT && !host_samples;
    c.start_x = start[0]; c.start_y = start[1]; c.root_reach = root_reach; c.root_vid = root_vid;
    c.rng_st_lo = (uint64_t)crng0.state; c.rng_st_hi = (uint64_t)(crng0.state >> 64); c.rng_inc_lo = (uint64_t)crng0.inc; c.rng_inc_hi = (uint64_t)(crng0.inc >> 64);
    c.jump = d_jump.p;
    c.self = d_rc.p;
    c.zero0 = (uint32_t *)d_cnt.p;
    c.zero_words = (uint64_t)(((char *)d_pendstate.p + pend_cap * sizeof(uint32_t)) - (char *)d_cnt.p) / 4;
    if (batch_prep) {
        char *z0 = (char *)d_cnt.p;
        if (!((char *)d_rgcnt.p > z0 && (char *)d_validmask.p > (char *)d_rgcnt.p && (char *)d_kdhint.p > (char *)d_validmask.p &&
              (char *)d_pendstate.p > (char *)d_kdhint.p) || ((uintptr_t)z0 & 15u)) {
            set_err("arena layout"); return PORRT_ERR_DEVICE;
        }
        jump_valid = true;           // the table the device derives is the one of this increment
        jump_inc = crng.inc;
    } else {
        double t0 = now_s();
        HIPCHK(hipMemcpyAsync(d_rc.p, &c, sizeof c, hipMemcpyHostToDevice, stream));
        if (!jump_valid || jump_inc != crng.inc) {            // the table depends on the stream's increment only: once per seed
            jump_host = jt;
            HIPCHK(hipMemcpyAsync(d_jump.p, &jump_host, sizeof jump_host, hipMemcpyHostToDevice, stream));
            jump_valid = true;
            jump_inc = crng.inc;
        }
        // counters, region counts, valid masks, kd hints (0 = the root: depth 0, id 0) and deferred-tie states lie next to
        // each other in the arena: one memset
        {
            char *z0 = (char *)d_cnt.p, *z1 = (char *)d_pendstate.p + pend_cap * sizeof(uint32_t);
            if (!((char *)d_rgcnt.p > z0 && (char *)d_validmask.p > (char *)d_rgcnt.p && (char *)d_kdhint.p > (char *)d_validmask.p &&
                  (char *)d_pendstate.p > (char *)d_kdhint.p)) {
                set_err("arena layout"); return PORRT_ERR_DEVICE;
            }
            HIPCHK(hipMemsetAsync(z0, 0, (size_t)(z1 - z0), stream));
        }
        HIPCHK(hipMemsetAsync(d_rep.p, 0xFF, kRepInts * sizeof(int), stream));
        HIPCHK(hipMemsetAsync(d_finalflag.p, 0, (size_t)((Nmax + 7) & ~7ull), stream));      // connect_rrt_sample only writes the set flags
        t_setup += now_s() - t0;
    }
    if (!batch_prep) hipLaunchKernelGGL(k_init_root, dim3(1), dim3(1), 0, stream, d_rc.p, start[0], start[1], (unsigned long long)root_reach, root_vid, 0);


    // profiling events
    const bool prof = opt_profile;
    size_t ev_used = 0;
    if (prof) {
        size_t want = (size_t)(steps_max + 2) * 4 + 4;
        while (ev_pool.size() < want) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            ev_pool.push_back(e);
        }
    }
    ScopedEvents<2> run_evs;                                // timing of a stand-alone grow (a batch member only prepares)
    if (stage != 1) HIPCHK(run_evs.create());
    hipEvent_t ev_first = run_evs.e[0], ev_last = run_evs.e[1];

    std::vector<uint32_t> &worlds = worlds_staging;      // outlives the asynchronous upload (a batch member returns without a sync)
    std::vector<double> hs_x, hs_y;
    Pcg64 hs_rng = crng0;   // exact host stream (fallback)
    // produce the samples of iterations [it0, it0+n) on the device buffers
    auto make_samples = [&](uint64_t it0, uint64_t n, uint32_t b0) -> int {
        double t0 = now_s();
        if (mode == PORRT_MODE_PTO) {
            worlds.resize(n);
            for (uint64_t t = 0; t < n; ++t) {
                if (has_inj_worlds) {
                    if (inj_wpos >= inj_worlds.size()) { set_err("injected world stream exhausted"); return PORRT_ERR_INVALID; }
                    worlds[t] = inj_worlds[inj_wpos++];
                } else {
                    worlds[t] = (uint32_t)drng.gen_range_usize((uint64_t)n_worlds);   // pto.rs:142
                }
                if (worlds[t] >= (uint32_t)n_worlds) { set_err("world index out of range"); return PORRT_ERR_INVALID; }
            }
            HIPCHK(hipMemcpyAsync(d_sworld.p + it0, worlds.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            if (!opt_graph) HIPCHK(hipStreamSynchronize(stream));   // `worlds` is reused by the next call
        }
        if (host_samples) {
            hs_x.resize(n); hs_y.resize(n);
            for (uint64_t t = 0; t < n; ++t) {
                uint64_t it = it0 + t + 1;
                if (it % 100 == 0) {
                    uint32_t w = mode == PORRT_MODE_PTO ? worlds[t] : 0;
                    if (goal_kind == 1) { hs_x[t] = w2g[w & 63][0]; hs_y[t] = w2g[w & 63][1]; }
                    else if (goal_kind == 2) { hs_x[t] = zone_pos[obs_zone][0]; hs_y[t] = zone_pos[obs_zone][1]; }
                    else { hs_x[t] = 0; hs_y[t] = 0; }
                } else if (has_inj) {
                    if (inj_pos >= inj_xy.size() / 2) { set_err("injected sample stream exhausted"); return PORRT_ERR_INVALID; }
                    hs_x[t] = inj_xy[2 * inj_pos]; hs_y[t] = inj_xy[2 * inj_pos + 1];
                    ++inj_pos;
                } else {
                    hs_x[t] = hs_rng.gen_range_f64(s_low[0], s_up[0]);
                    hs_y[t] = hs_rng.gen_range_f64(s_low[1], s_up[1]);
                }
            }
            HIPCHK(hipMemcpyAsync(d_sx.p + it0, hs_x.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(d_sy.p + it0, hs_y.data(), n * sizeof(double), hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
        } else if (!batch_prep) {
            hipLaunchKernelGGL(k_gen_samples, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const RunConst *)d_rc.p,
                               (const PcgJump *)d_jump.p, (unsigned long long)it0, (unsigned long long)n,
                               (unsigned long long)(uint64_t)crng0.state, (unsigned long long)(uint64_t)(crng0.state >> 64),
                               (unsigned long long)(uint64_t)crng0.inc, (unsigned long long)(uint64_t)(crng0.inc >> 64), 0ull);
        }
        if (mode == PORRT_MODE_RRT && n > 0 && !batch_prep)       // the step kernels take a step's samples in spatial order
            hipLaunchKernelGGL(k_sort_samples, dim3((unsigned)((n + K - 1) / K)), dim3(256), 0, stream, (const RunConst *)d_rc.p, b0,
                               (unsigned long long)it0, (unsigned long long)n, K);
        t_setup += now_s() - t0;
        return PORRT_OK;
    };

    // ---- growth loop (rrt.rs:109 / pto.rs:67)
    uint64_t i = 0;
    uint32_t b = 0;
    int rcode = PORRT_OK;
    if (stage == 1) { batch_drng0 = drng; batch_wpos0 = inj_wpos; batch_world_draws = n_iter_min; }
    if (n_iter_min > 0) {
        double t0 = now_s();
        int r = ensure_radius_table(max_step, search_radius, n_iter_min + 4);
        if (r) return r;
        t_setup += now_s() - t0;
        r = make_samples(0, n_iter_min, 0);
        if (r) return r;
    }
    run_lds_bytes = lds_bytes;
    if (stage == 1) return PORRT_OK;
    { int r = ensure_side_stream(); if (r) return r; }
    launch_rcp = d_rc.p;
    launch_Q = 1;
    opt_group = opt_group_req < 0 ? 0u : (uint32_t)opt_group_req;
    // (a single query's one-kernel-per-step form carries the goal path's workgroup too: since the workgroup finds a node's exit level by
    // bisection it is no longer than the step's other workgroups, and the side chain's kernels are not launched at all -- 4.23 against
    // 4.59 ms per query)
    kd_lazy = opt_kd_lazy && (opt_group != 0 || lag_on) && mode == PORRT_MODE_RRT && !opt_kd_after;
    kd_built_after = 0;
    gt_pend = false; gt_par = 0;
    if ((c.kd_lazy != 0u) != kd_lazy) {          // (the run constants were uploaded above)
        c.kd_lazy = kd_lazy ? 1u : 0u;
        HIPCHK(hipMemcpyAsync(d_rc.p, &c, sizeof c, hipMemcpyHostToDevice, stream));
    }
    HIPCHK(hipEventRecord(ev_first, stream));
    side_active = false;
    pipe_near_done = 0xFFFFFFFFu;
    commit_pend_b = 0xFFFFFFFFu;
    kd_b0 = 0; kd_last_b = 0; kd_last_nb = 0; kd_gidx = 0; kd_hint_ns = 0;
    kd_pend[0] = kd_pend[1] = kd_pend[2] = false;
    kd_group = kd_group_for(K, opt_kd_group, 1);
    bool coop_done = false;
    if (pipe_on && (opt_pipeline == 2 || opt_pipeline == 3) && K <= 1024 && !prof && n_iter_min > 0) {
        uint32_t nst = (uint32_t)((n_iter_min + K - 1) / K);
        if (launch_coop(nst, K, n_iter_min, vwords, lds_bytes) == PORRT_OK) {
            coop_done = true;
            i = n_iter_min; b = nst;
            kd_b0 = b; kd_last_b = b ? b - 1 : 0;
        }
    }
    if (coop_done) {
    } else if (opt_graph && !prof && n_iter_min > 0) {
        // all steps up to n_iter_min as one hipGraph (two branches: main pipeline + kd insertion); the graph
        // only depends on the launch geometry, so it is instantiated once and replayed by later grows
        const uint64_t key[6] = {(uint64_t)mode, K, n_iter_min, lds_bytes, (uint64_t)(uintptr_t)launch_rcp, kd_group | ((uint64_t)opt_early_wave << 8) | ((uint64_t)launch_Q << 32) | ((uint64_t)opt_group << 48) | ((uint64_t)pipe_on << 56) | ((uint64_t)lag_on << 57) | ((uint64_t)kd_lazy << 58) | ((uint64_t)(kd_lazy && opt_gtrack_side) << 59)};
        if (!graph_exec || memcmp(key, graph_key, sizeof key)) {
            double t0 = now_s();
            if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
            hipGraph_t g = nullptr;
            // stream capture: the side stream only ever waits on events of the capturing stream, and the capturing
            // stream joins it once at the end (HIP's capture bookkeeping does not take side streams that wait on each other)
            HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            uint64_t ci = 0;
            uint32_t cb = 0;
            while (ci < n_iter_min) {
                uint32_t nb = (uint32_t)std::min<uint64_t>(K, n_iter_min - ci);
                uint64_t i2 = ci + nb;                                                       // start of step cb+1
                uint32_t nb2 = i2 < n_iter_min ? (uint32_t)std::min<uint64_t>(K, n_iter_min - i2) : 0;
                launch_step(cb, (uint32_t)ci, nb, vwords, lds_bytes, false, ev_used, (uint32_t)i2, nb2);
                ci += nb;
                ++cb;
            }
            join_side();
            HIPCHK(hipStreamEndCapture(stream, &g));
            HIPCHK(hipGraphInstantiate(&graph_exec, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            memcpy(graph_key, key, sizeof key);
            t_setup += now_s() - t0;
        }
        HIPCHK(hipGraphLaunch(graph_exec, stream));
        while (i < n_iter_min) { i += std::min<uint64_t>(K, n_iter_min - i); ++b; }
        kd_b0 = b; kd_last_b = b ? b - 1 : 0;      // the graph inserted every step it ran into the kd structure and joined
    } else {
        while (i < n_iter_min) {
            uint32_t nb = (uint32_t)std::min<uint64_t>(K, n_iter_min - i);
            uint64_t i2 = i + nb;
            uint32_t nb2 = i2 < n_iter_min ? (uint32_t)std::min<uint64_t>(K, n_iter_min - i2) : 0;
            launch_step(b, (uint32_t)i, nb, vwords, lds_bytes, prof, ev_used, (uint32_t)i2, nb2);
            i += nb;
            ++b;
        }
        join_side();
    }
    Counters hc;
    auto read_counters = [&]() -> int {
        HIPCHK(hipMemcpyAsync(&hc, d_cnt.p, sizeof hc, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return PORRT_OK;
    };
    auto is_done = [&]() {
        if (mode == PORRT_MODE_RRT) return hc.n_final > 0;                                       // rrt.rs:109
        return hc.n_final > 0 && (hc.finality & c.all_worlds) == c.all_worlds;                    // pto_reachability.rs:81-90
    };
    {
        int r = read_counters();
        if (r) return r;
    }
    while (!is_done() && i < n_iter_max && !(hc.err & (ERR_CAND_OVERFLOW | ERR_RNG_RETRY))) {
        uint32_t nb = (uint32_t)std::min<uint64_t>(K, n_iter_max - i);
        double t0 = now_s();
        int r = ensure_radius_table(max_step, search_radius, i + nb + 4);
        if (r) return r;
        t_setup += now_s() - t0;
        r = make_samples(i, nb, b);
        if (r) return r;
        launch_step(b, (uint32_t)i, nb, vwords, lds_bytes, prof, ev_used, 0, 0);
        join_side();
        i += nb;
        ++b;
        r = read_counters();
        if (r) return r;
    }
    if (kd_lazy && hc.n_lca && !(hc.err & (ERR_CAND_OVERFLOW | ERR_RNG_RETRY))) {
        // a tie between two nodes off the goal path: the whole kd structure after all, then the records that waited for it
        // (for the steps up to the last such tie: its record names older nodes only)
        std::vector<std::pair<uint32_t, uint32_t>> segs;
        const uint32_t s1 = (uint32_t)((n_iter_min + K - 1) / K), S = std::min<uint32_t>(hc.lca_next, b);
        if (S <= s1) segs.emplace_back(S - 1u, S == s1 ? (uint32_t)(n_iter_min - (uint64_t)(s1 - 1u) * K) : K);
        else {
            segs.emplace_back(s1 - 1u, (uint32_t)(n_iter_min - (uint64_t)(s1 - 1u) * K));
            segs.emplace_back(S - 1u, S == b ? (uint32_t)(i - n_iter_min - (uint64_t)(b - 1u - s1) * K) : K);
        }
        int r = kd_full_build(segs);
        if (r) return r;
        kd_built_after = 1;
        r = read_counters();
        if (r) return r;
    }
    HIPCHK(hipEventRecord(ev_last, stream));
    uint32_t n_final_nodes = 0;
    HIPCHK(hipMemcpyAsync(&n_final_nodes, d_nat.p + b, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_err(std::string("kernel launch: ") + hipGetErrorString(e)); return PORRT_ERR_DEVICE; }
    }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev_first, ev_last));

    if (hc.coop_abort) { set_err("the persistent step loop gave up waiting (a barrier or a kd kernel beside it): results discarded"); return PORRT_ERR_DEVICE; }
    if (hc.err & ERR_CAND_OVERFLOW) return -100;
    if (hc.err & ERR_RNG_RETRY) {
        if (has_inj) { set_err("injected sample stream exhausted"); return PORRT_ERR_INVALID; }
        return -101;
    }
    if ((hc.err & ERR_EDGE_OVERFLOW) && (Nmax * edge_per_node + 4096 < (1ull << 31))) return -102;
    if ((hc.err & ERR_TIE_POOL) && tie_pool_mult < (1ull << 12)) return -103;
    counters = hc;
    if (getenv("PORRT_DEBUG")) {
        fprintf(stderr, "[porrt] tie fallbacks %u g_len %u; samples served through the lists in memory %u\n", hc.tie_fallbacks, hc.g_len, hc.n_heavy);
        fprintf(stderr, "[porrt] deferred ties: records %u pooled ids %u settled %u; goal path: %u levels, %u of them not copies of the goal point\n", hc.pend_cnt, hc.pool_n, hc.n_deferred, hc.g_len, hc.g_nd_len);
        for (int t = 0; t < 8; ++t)
            if (hc.tim[t + 8]) fprintf(stderr, "[porrt] phase %d: %.2f us per wave over %llu waves (total %.1f wave-ms)\n", t, 1e-2 * (double)hc.tim[t] / (double)hc.tim[t + 8],
                                       (unsigned long long)hc.tim[t + 8], 1e-5 * (double)hc.tim[t]);
    }
    n_iter = i;
    n_steps = b;
    n_nodes = n_final_nodes;
    complete = mode == PORRT_MODE_PTO ? is_done() : hc.n_final > 0;
    have_results = true;
    ++results_tag;

    // advance the persistent sampler state by what this grow consumed
    const uint64_t calls = i - i / 100;
    if (host_samples) { if (!has_inj) crng = hs_rng; }
    else if (has_inj) inj_pos += calls;
    else crng.advance((u128)2 * calls);

    metrics.n_iter = n_iter;
    metrics.n_nodes = n_nodes;
    metrics.n_steps = n_steps;
    metrics.n_tie_fallbacks = hc.tie_fallbacks + ((hc.err & ERR_GPATH_OVERFLOW) ? 1 : 0);
    metrics.device_s = ms * 1e-3;
    if (prof) {
        // events were recorded around k_near [0,1] and the connect kernel [2,3] of every step
        std::vector<uint32_t> nat(b + 1);
        HIPCHK(hipMemcpy(nat.data(), d_nat.p, (b + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        double scan = 0, conn = 0, pairs = 0, bytes = 0;
        uint64_t launches = 0;
        uint64_t it = 0;
        for (uint32_t s = 0; s < b && (size_t)(4 * s + 3) < ev_used; ++s) {
            float a = 0, r2 = 0;
            (void)hipEventElapsedTime(&a, ev_pool[4 * s + 0], ev_pool[4 * s + 1]);
            (void)hipEventElapsedTime(&r2, ev_pool[4 * s + 2], ev_pool[4 * s + 3]);
            scan += a * 1e-3;
            conn += r2 * 1e-3;
            launches += 1;
            if (getenv("PORRT_DEBUG_STEPS")) fprintf(stderr, "[porrt] step %u N %u near %.1f us connect %.1f us\n", s, nat[s], a * 1e3, r2 * 1e3);
            uint64_t nbq = std::min<uint64_t>(K, (it < n_iter_min ? n_iter_min : n_iter_max) - it);
            it += nbq;
            // what the two searches of the step answer: every (sample, node) pair of the NN and of the radius query
            pairs += 2.0 * (double)nbq * (double)nat[s];
            // algorithmic bytes of the searches: node x,y, the samples, nn / state writes
            bytes += 16.0 * (double)nat[s] + (16.0 + 20.0) * (double)nbq;
        }
        metrics.scan_s = scan;
        metrics.connect_s = conn;
        metrics.scan_launches = launches;
        metrics.scan_pairs = pairs;
        metrics.scan_bytes = bytes;
    }
    metrics.setup_s = t_setup;
    metrics.total_s = now_s() - t_begin;

    if (hc.err & ERR_RASTER) { set_err("raster access outside the map, door pixel without zone id, or two zones on one segment (the reference panics here)"); return PORRT_ERR_RASTER; }
    if (hc.err & ERR_EDGE_OVERFLOW) { set_err("edge pool overflow"); return PORRT_ERR_CAPACITY; }
    if (hc.err & (ERR_TIE_POOL | ERR_PAGE_OVERFLOW)) { set_err("deferred-tie pool / region page pool overflow"); return PORRT_ERR_CAPACITY; }
    if (mode == PORRT_MODE_PTO && !complete) { set_err("final nodes are not reached for each world"); rcode = PORRT_INCOMPLETE; }
    return rcode;
}

// Results are fetched lazily and in parts: a caller that only asks for the best path (porrt_best_solution) pays for
// coordinates, parents and final flags; dist_root, masks, validity ids and edges come when a getter asks for them.
int porrt_ctx::download(unsigned want) {
    if (!have_results) { set_err("no results: call porrt_grow first"); return PORRT_ERR_INVALID; }
    if (downloaded_tag != results_tag) { got = 0; downloaded_tag = results_tag; }
    const unsigned need = want & ~got;
    if (!need) return PORRT_OK;
    HIPCHK(hipSetDevice(device));
    const size_t N = n_nodes;
    if (need & DL_TREE) {
        h_nx.resize(N); h_ny.resize(N); h_parent.resize(N); h_finalflag.resize(N);
        HIPCHK(hipMemcpy(h_nx.data(), d_nx.p, N * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_ny.data(), d_ny.p, N * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_parent.data(), d_parent.p, N * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_finalflag.data(), d_finalflag.p, N, hipMemcpyDeviceToHost));
        h_final_ids.clear();
        for (size_t j = 0; j < N; ++j)
            if (h_finalflag[j]) h_final_ids.push_back(j);   // ascending id == push order of the reference
    }
    if (need & DL_DIST) {
        h_dist.resize(N);
        HIPCHK(hipMemcpy(h_dist.data(), d_distA.p, N * 8, hipMemcpyDeviceToHost));
    }
    if (need & DL_MASKS) {
        h_reach.resize(N); h_finalmask.resize(N); h_vid.resize(N);
        HIPCHK(hipMemcpy(h_reach.data(), d_reachA.p, N * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_finalmask.data(), d_finalmask.p, N * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(h_vid.data(), d_vid.p, N, hipMemcpyDeviceToHost));
    }
    if (!(need & DL_EDGES)) { got |= need; return PORRT_OK; }
    if (mode == PORRT_MODE_PTO || mode == PORRT_MODE_PRM) {
        // Edges come back in the order PTOGraph's adjacency lists are filled (pto.rs:103-120): new nodes ascending, and
        // for one new node its neighbours in the order KdTree::nearest_neighbors lists them -- kd pre-order
        // (nearest_neighbor.rs:101-117).  The growth kernels only kept the edge set; the order is restored on the device
        // (porrt_edges.hpp) and copied out here.
        int r = ensure_edge_order();
        if (r) return r;
        const size_t E = counters.n_edges;
        h_efrom.resize(E); h_eto.resize(E); h_etv.resize(E);
        if (E) {
            HIPCHK(hipMemcpy(h_efrom.data(), eo.d_from, E * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(h_eto.data(), eo.d_to, E * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(h_etv.data(), eo.d_val, E * 4, hipMemcpyDeviceToHost));
        }
    } else {
        h_efrom.clear(); h_eto.clear(); h_etv.clear();
    }
    got |= need;
    return PORRT_OK;
}

// ========================================================================================== C ABI
// KdTree::nearest_neighbor (nearest_neighbor.rs:48-91) on the kd-tree of the nodes in id order: strict improvement, the
// side of the query first -- the literal traversal, so that equal distances resolve as in the reference.
namespace {
struct HostKd {
    const std::vector<double> &x, &y;
    std::vector<int> left, right;
    HostKd(const std::vector<double> &xs, const std::vector<double> &ys) : x(xs), y(ys), left(xs.size(), -1), right(xs.size(), -1) {
        for (size_t id = 1; id < x.size(); ++id) {                 // KdTree::add (nearest_neighbor.rs:29-46)
            size_t cur = 0;
            for (uint32_t d = 0;; ++d) {
                const bool l = (d & 1u) ? (y[id] < y[cur]) : (x[id] < x[cur]);
                int &c = l ? left[cur] : right[cur];
                if (c < 0) { c = (int)id; break; }
                cur = (size_t)c;
            }
        }
    }
    static double norm2(double ax, double ay, double bx, double by) {
        double d2 = 0.0;
        const double dx = bx - ax, dy = by - ay;
        d2 += dx * dx;
        d2 += dy * dy;
        return std::sqrt(d2);
    }
    size_t nearest(double qx, double qy) const {
        double dmin = std::numeric_limits<double>::infinity();
        size_t best = 0;
        struct Frame { int node; uint32_t axis; int stage; };
        std::vector<Frame> st{{0, 0, 0}};
        while (!st.empty()) {                                       // the recursion of `inner`, unrolled
            Frame &f = st.back();
            const int n = f.node;
            const double s = f.axis ? y[n] : x[n], q = f.axis ? qy : qx;
            const bool left_first = q < s;
            if (f.stage == 0) {
                const double d = norm2(x[n], y[n], qx, qy);
                if (d < dmin) { dmin = d; best = (size_t)n; }
            }
            if (f.stage >= 2) { st.pop_back(); continue; }
            const int stage = f.stage++;
            const bool go_left = (stage == 0) == left_first;        // first the query's side, then the other
            const uint32_t next_axis = (f.axis + 1) % 2;
            if (go_left) { if (q - dmin < s && left[n] >= 0) st.push_back({left[n], next_axis, 0}); }
            else { if (q + dmin >= s && right[n] >= 0) st.push_back({right[n], next_axis, 0}); }
        }
        return best;
    }
};
static HostKd *host_kd_of(porrt_ctx *c) {                            // built once per graph
    if (c->host_kd_tag != c->results_tag || !c->host_kd) {
        c->host_kd = std::shared_ptr<void>(new HostKd(c->h_nx, c->h_ny), [](void *p) { delete (HostKd *)p; });
        c->host_kd_tag = c->results_tag;
    }
    return (HostKd *)c->host_kd.get();
}
} // namespace

// The pre-order rank of every node in the kd-tree of the coordinates (sequential KdTree::add in id order,
// nearest_neighbor.rs:29-46) -- from the host's kd-tree, or (option host_ranks = 0) made on the device --, then the edge order and
// the adjacency lists on the device (porrt_edges.hpp).
int porrt_ctx::ensure_edge_order() {
    if (eo.tag == results_tag) return PORRT_OK;
    HIPCHK(hipSetDevice(device));
    const size_t N = n_nodes;
    std::string e;
    int r;
    if (opt_host_ranks) {                    // (the default; option "host_ranks" = 0 makes them on the device, below)
        if ((r = download(DL_TREE))) return r;
        const HostKd *kd = host_kd_of(this);
        const std::vector<int> &ch0 = kd->left, &ch1 = kd->right;
        std::vector<uint32_t> rank(N, 0);
        std::vector<int> stack;
        uint32_t next = 0;
        if (N) stack.push_back(0);
        while (!stack.empty()) {
            const int n = stack.back();
            stack.pop_back();
            rank[n] = next++;
            if (ch1[n] >= 0) stack.push_back(ch1[n]);      // left subtree first (popped first)
            if (ch0[n] >= 0) stack.push_back(ch0[n]);
        }
        r = edge_order_build(eo, results_tag, N, (size_t)counters.n_edges, d_efrom.p, d_eto.p, d_etv.p, &rank, nullptr, nullptr, stream, e);
    } else {
        r = edge_order_build(eo, results_tag, N, (size_t)counters.n_edges, d_efrom.p, d_eto.p, d_etv.p, nullptr, d_nx.p, d_ny.p, stream, e);
    }
    if (r) set_err(e);
    return r;
}

// PTO::build_belief_graph (pto.rs:185-259) on the graph of the last grow; see porrt_belief.hpp.
int porrt_ctx::build_belief_graph(const double *start_belief, uint32_t n_worlds_in) {
    if (!have_results || mode != PORRT_MODE_PTO) { set_err("build_belief_graph: grow a PTO graph first (porrt_grow, mode PORRT_MODE_PTO)"); return PORRT_ERR_INVALID; }
    if (!start_belief || (int)n_worlds_in != n_worlds) { set_err("build_belief_graph: the start belief needs one probability per world"); return PORRT_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    const double te0 = now_s();
    int r = download(DL_TREE | DL_MASKS);
    if (r) return r;
    if ((r = ensure_edge_order())) return r;
    const double t_edges = now_s() - te0;
    BeliefInputs in{};
    in.domain = domain; in.n_zones = n_zones; in.n_worlds = n_worlds; in.n_validities = n_validities;
    in.validities = validities; in.zone_pos = zone_pos; in.visibility = visibility;
    in.d_rc = d_rc.p;
    in.N = (size_t)n_nodes; in.E = (size_t)counters.n_edges;
    in.d_nx = d_nx.p; in.d_ny = d_ny.p; in.d_vid = d_vid.p; in.h_vid = h_vid.data();
    in.d_adj_off = eo.d_adj_off; in.d_adj_id = eo.d_adj_id; in.d_adj_val = eo.d_adj_val; in.d_radj_id = eo.d_radj_id; in.d_radj_val = eo.d_radj_val;
    in.graph_tag = results_tag;
    bg_graph_tag = results_tag;
    in.stream = stream;
    std::string e;
    r = belief_graph_build(bg, in, start_belief, e);
    if (r) { bg.release(); set_err(e); }
    else { bg.t_edges = t_edges; bg.t_total += t_edges; }
    return r;
}

// PTO::compute_expected_costs_to_goals (pto.rs:261-275): the final belief nodes, then conditional_dijkstra on the device.
int porrt_ctx::compute_expected_costs() {
    if (!bg.valid || bg_graph_tag != results_tag) { set_err("compute_expected_costs: build the belief graph first (porrt_build_belief_graph)"); return PORRT_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    int r = download(DL_TREE | DL_MASKS);
    if (r) return r;
    const BeliefSpace &bs = bg.cache.space;
    const size_t B = bg.B;
    std::vector<unsigned long long> finals;
    std::vector<uint64_t> possible(B, 0);             // per belief: the worlds it gives a probability > 0 (at most 64 worlds: the finality masks are words)
    for (size_t b = 0; b < B; ++b)
        for (uint32_t w = 0; w < bs.nw; ++w) possible[b] |= bs.at(b)[w] > 0.0 ? 1ull << w : 0ull;
    for (uint64_t id : h_final_ids) {                 // final_nodes_with_validities(): push order = ascending id here
        const uint64_t finality = h_finalmask[id];
        for (size_t b = 0; b < B; ++b) {
            if (!((bg.cache.compat[b] >> h_vid[id]) & 1ull)) continue;          // node_to_belief_nodes[final_id][b] is None
            if (!(possible[b] & ~finality)) finals.push_back((unsigned long long)(id * B + b));     // is_compatible(belief_state, validity)
        }
    }
    DpConst c{};
    c.n = (unsigned long long)(bg.N * bg.B); c.B = (uint32_t)bg.B; c.nw = bg.nw;
    c.nx = d_nx.p; c.ny = d_ny.p; c.bvec = nullptr; c.beliefs = bg.d_beliefs; c.types = bg.d_types;
    c.child_off = bg.d_child_off; c.par_off = bg.d_par_off; c.child_id = bg.d_child_id; c.par_id = bg.d_par_id;
    std::string e;
    // layers solved one after the other when observations always shrink the set of possible worlds (always, in the
    // reference's domains; checked by the build), the general sweeps otherwise or on request (option "dp_sweeps")
    if (bg.support_shrinks && !opt_dp_sweeps) r = dp_run_layered(dp, bg, c, finals, stream, e);
    else r = dp_run(dp, c, true, finals, stream, e);
    if (r) set_err(e);
    return r;
}

// PTO::extract_policy (pto.rs:277-283, belief_graph.rs:177-263) on the expected costs of the last compute_expected_costs.
int porrt_ctx::extract_policy() {
    if (!bg.valid || !dp.valid || bg_graph_tag != results_tag) { set_err("extract_policy: compute the expected costs first (porrt_bg_compute_expected_costs)"); return PORRT_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    const BeliefSpace &bs = bg.cache.space;
    const size_t B = bg.B;
    auto belief_of = [B](uint64_t i) { return (uint32_t)(i % B); };
    auto p_of = [&bs, B](uint64_t parent, uint64_t child) {          // transition_probability (common.rs:187-190)
        const double *pb = bs.at(parent % B), *cb = bs.at(child % B);
        double s = 0.0;
        for (uint32_t w = 0; w < bs.nw; ++w) s = s + (cb[w] > 0.0 ? pb[w] : 0.0);
        return s;
    };
    std::string e;
    int r = dp_extract_policy(dp, true, belief_of, p_of, stream, e);
    if (r) set_err(e);
    return r;
}

// PRM::init + PRM::grow_graph (prm.rs:33-109); see porrt_prm.hpp.
int porrt_ctx::grow_prm(const double start[2], double max_step, double search_radius, uint64_t n_iter_) {
    const uint64_t n_iter = n_iter_;
    if (!has_grid) { set_err("PRM growth needs a grid-backed domain (porrt_set_grid)"); return PORRT_ERR_INVALID; }
    if (!(max_step > 0.0) || !(search_radius > 0.0) || !start) { set_err("bad PRM parameters"); return PORRT_ERR_INVALID; }
    if (n_iter + 1 >= 0x7FFFFFFFull) { set_err("too many PRM samples"); return PORRT_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    const double t0 = now_s();
    have_results = false;
    batch_leader = nullptr;
    bg.release();
    dp.release();
    ++results_tag;
    const size_t N = (size_t)n_iter + 1;
    // the nodes: start, then one sample per iteration (ContinuousSampler::sample: x, then y)
    std::vector<double> hx(N), hy(N);
    hx[0] = start[0]; hy[0] = start[1];
    const bool has_inj = !inj_xy.empty();
    for (size_t i = 1; i < N; ++i) {
        if (has_inj) {
            if (inj_pos >= inj_xy.size() / 2) { set_err("injected sample stream exhausted"); return PORRT_ERR_INVALID; }
            hx[i] = inj_xy[2 * inj_pos]; hy[i] = inj_xy[2 * inj_pos + 1];
            ++inj_pos;
        } else {
            hx[i] = crng.gen_range_f64(s_low[0], s_up[0]);
            hy[i] = crng.gen_range_f64(s_low[1], s_up[1]);
        }
    }
    double x0 = hx[0], x1 = hx[0], y0 = hy[0], y1 = hy[0];
    for (size_t i = 1; i < N; ++i) { x0 = std::min(x0, hx[i]); x1 = std::max(x1, hx[i]); y0 = std::min(y0, hy[i]); y1 = std::max(y1, hy[i]); }
    const uint64_t ecap = std::min<uint64_t>((uint64_t)N * 256 + 4096, 1ull << 30);
    HIPCHK(d_nx.reserve(N)); HIPCHK(d_ny.reserve(N)); HIPCHK(d_distA.reserve(N)); HIPCHK(d_parent.reserve(N)); HIPCHK(d_reachA.reserve(N));
    HIPCHK(d_vid.reserve(N)); HIPCHK(d_finalflag.reserve(N)); HIPCHK(d_finalmask.reserve(N)); HIPCHK(d_radT2.reserve(N + 8));
    HIPCHK(d_cnt.reserve(1)); HIPCHK(d_rc.reserve(1)); HIPCHK(d_cls.reserve(cls_bytes(W, H)));       // classes + clearance plane (build_cls)
    HIPCHK(d_efrom.reserve(ecap)); HIPCHK(d_eto.reserve(ecap)); HIPCHK(d_etv.reserve(ecap));
    int r = layout_buffers();
    if (r) return r;
    cls_dirty = true;                                   // a re-layout loses the raster
    if ((r = build_cls())) return r;
    rad_uploaded = 0;
    if ((r = ensure_radius_table(max_step, search_radius, N + 2))) return r;
    HIPCHK(hipMemcpyAsync(d_nx.p, hx.data(), N * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(d_ny.p, hy.data(), N * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemsetAsync(d_parent.p, 0xFF, N * sizeof(int), stream));
    HIPCHK(hipMemsetAsync(d_distA.p, 0, N * 8, stream));
    HIPCHK(hipMemsetAsync(d_reachA.p, 0, N * 8, stream));
    HIPCHK(hipMemsetAsync(d_finalmask.p, 0, N * 8, stream));
    HIPCHK(hipMemsetAsync(d_vid.p, 0, N, stream));
    HIPCHK(hipMemsetAsync(d_finalflag.p, 0, N, stream));
    memset(&rc, 0, sizeof rc);
    rc.nx = d_nx.p; rc.ny = d_ny.p; rc.vid = d_vid.p;
    rc.cls = d_cls.p; rc.clr = d_cls.p + (size_t)W * H; rc.sat = opt_box_table ? (const uint32_t *)(d_cls.p + cls_sat_offset(W, H)) : nullptr; rc.W = W; rc.H = H; rc.low0 = low[0]; rc.low1 = low[1]; rc.ppm = ppm; rc.domain = domain; rc.has_grid = has_grid;
    rc.n_validities = n_validities;
    for (int i = 0; i < n_validities; ++i) rc.validities[i] = validities[i];
    rc.all_worlds = ones(n_worlds);
    rc.visibility = visibility;
    rc.rad_T2 = d_radT2.p;
    HIPCHK(hipMemcpyAsync(d_rc.p, &rc, sizeof rc, hipMemcpyHostToDevice, stream));
    // the grid: cells about as wide as the radius of the last nodes (the smallest; most nodes are near it); a node scans
    // the window of cells its own radius reaches
    const double ext = std::max(std::max(x1 - x0, y1 - y0), max_step);
    const double r_last = std::sqrt(std::max(radT2[N], 0.0));
    uint32_t G = (uint32_t)std::min<double>(1024.0, std::max(1.0, std::floor(ext / std::max(r_last, ext / 1024.0))));
    const double cell = ext / (double)G;
    const size_t cells = (size_t)G * G;
    const size_t nblk = (std::max(cells, N) + kScanTile - 1) / kScanTile;
    if (prm.cells_cap < cells) {
        void *drop[] = {prm.d_cell_cnt, prm.d_cell_off};
        for (void *q : drop) if (q) (void)hipFree(q);
        prm.d_cell_cnt = nullptr; prm.d_cell_off = nullptr; prm.cells_cap = 0;
        HIPCHK(hipMalloc((void **)&prm.d_cell_cnt, cells * sizeof(uint32_t)));
        HIPCHK(hipMalloc((void **)&prm.d_cell_off, (cells + 1) * sizeof(unsigned long long)));
        prm.cells_cap = cells;
    }
    if (prm.ids_cap < N) {
        void *drop[] = {prm.d_cell_ids, prm.d_deg, prm.d_edge_off};
        for (void *q : drop) if (q) (void)hipFree(q);
        prm.d_cell_ids = prm.d_deg = nullptr; prm.d_edge_off = nullptr; prm.ids_cap = 0;
        const size_t cap = N + N / 8 + 1;
        HIPCHK(hipMalloc((void **)&prm.d_cell_ids, cap * sizeof(uint32_t)));
        HIPCHK(hipMalloc((void **)&prm.d_deg, cap * sizeof(uint32_t)));
        HIPCHK(hipMalloc((void **)&prm.d_edge_off, (cap + 1) * sizeof(unsigned long long)));
        prm.ids_cap = cap - 1;
    }
    if (prm.d_tot) (void)hipFree(prm.d_tot);
    prm.d_tot = nullptr;
    HIPCHK(hipMalloc((void **)&prm.d_tot, (nblk + 2) * sizeof(unsigned long long)));
    if (!prm.d_err) HIPCHK(hipMalloc((void **)&prm.d_err, sizeof(uint32_t)));
    PrmConst p{};
    p.N = (uint32_t)N; p.G = G; p.nx = d_nx.p; p.ny = d_ny.p; p.rad_T2 = d_radT2.p;
    p.x0 = x0; p.y0 = y0; p.inv_cell = 1.0 / cell;
    p.cell_cnt = prm.d_cell_cnt; p.cell_off = prm.d_cell_off; p.cell_ids = prm.d_cell_ids;
    p.efrom = d_efrom.p; p.eto = d_eto.p; p.ev = d_etv.p; p.deg = prm.d_deg; p.edge_off = prm.d_edge_off; p.err = prm.d_err;
    ScopedEvents<2> evs;
    HIPCHK(evs.create());
    hipEvent_t ev0 = evs.e[0], ev1 = evs.e[1];
    HIPCHK(hipEventRecord(ev0, stream));
    HIPCHK(hipMemsetAsync(prm.d_cell_cnt, 0, cells * sizeof(uint32_t), stream));
    HIPCHK(hipMemsetAsync(prm.d_deg, 0, N * sizeof(uint32_t), stream));
    HIPCHK(hipMemsetAsync(prm.d_err, 0, sizeof(uint32_t), stream));
    const dim3 ngrid((unsigned)((N + 255) / 256)), block(256), wgrid((unsigned)((N - 1 + 3) / 4));
    hipLaunchKernelGGL(k_prm_bin<false>, ngrid, block, 0, stream, p);
    bg_scan(prm.d_cell_cnt, cells, prm.d_tot, prm.d_cell_off, stream);
    HIPCHK(hipMemsetAsync(prm.d_cell_cnt, 0, cells * sizeof(uint32_t), stream));
    hipLaunchKernelGGL(k_prm_bin<true>, ngrid, block, 0, stream, p);
    unsigned long long n_edges = 0;
    uint32_t h_err = 0;
    if (N > 1) {
        hipLaunchKernelGGL(k_prm_connect<false>, wgrid, block, 0, stream, (const RunConst *)d_rc.p, p);
        bg_scan(prm.d_deg, N, prm.d_tot, prm.d_edge_off, stream);
        HIPCHK(hipMemcpyAsync(&n_edges, prm.d_edge_off + N, sizeof n_edges, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (n_edges > ecap) { set_err("PRM edge list outgrew its capacity (256 per node)"); return PORRT_ERR_CAPACITY; }
        hipLaunchKernelGGL(k_prm_connect<true>, wgrid, block, 0, stream, (const RunConst *)d_rc.p, p);
    }
    HIPCHK(hipEventRecord(ev1, stream));
    HIPCHK(hipMemcpyAsync(&h_err, prm.d_err, sizeof h_err, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    if (h_err & ERR_RASTER) { set_err("raster access the reference would panic on (image::get_pixel out of range, door pixel without zone, two zones on one segment)"); return PORRT_ERR_RASTER; }
    memset(&counters, 0, sizeof counters);
    counters.n_edges = (uint32_t)n_edges;
    mode = PORRT_MODE_PRM;
    n_nodes = N; this->n_iter = n_iter_; n_steps = 0;
    complete = false;
    prm.t_device = 1e-3 * (double)ms;
    prm.t_total = now_s() - t0;
    memset(&metrics, 0, sizeof metrics);
    metrics.n_iter = n_iter_; metrics.n_nodes = N; metrics.total_s = prm.t_total; metrics.device_s = prm.t_device;
    have_results = true;
    return PORRT_OK;
}

// PRM::plan_path (prm.rs:111-123): dijkstra from the goal's nearest node (pto_graph.rs:275-303 == conditional_dijkstra
// without observation nodes: the device sweeps), extract_path on the host (pto_graph.rs:305-326).

// One mode's roadmap from its ordered nodes: the kernels of porrt_grow_prm on an injected point list (node 0 = the first
// point, prm.rs:54-58), forward edges back in the reference's adjacency order.  The context's own sampler state and
// injected stream are put back afterwards.
int porrt_ctx::roadmap_of_points(const std::vector<double> &xy, double max_step, double search_radius, std::vector<uint32_t> &efrom, std::vector<uint32_t> &eto,
                                 double &dev_s) {
    efrom.clear(); eto.clear();
    const size_t n = xy.size() / 2;
    if (n < 2) return PORRT_OK;
    std::vector<double> keep_inj;
    keep_inj.swap(inj_xy);
    const size_t keep_pos = inj_pos;
    const bool keep_has = has_inj;
    const Pcg64 keep_rng = crng;
    inj_xy.assign(xy.begin() + 2, xy.end());
    inj_pos = 0; has_inj = true;
    const double start[2] = {xy[0], xy[1]};
    int r = grow_prm(start, max_step, search_radius, n - 1);
    if (!r) r = download(DL_EDGES);
    if (!r) { efrom = h_efrom; eto = h_eto; dev_s += prm.t_device; }
    inj_xy.swap(keep_inj); inj_pos = keep_pos; has_inj = keep_has; inj_dirty = true; crng = keep_rng;
    return r;
}

struct DeviceScratch {
    std::vector<void *> ps;
    ~DeviceScratch() { for (void *q : ps) if (q) (void)hipFree(q); }
    template <class T> hipError_t get(T *&p, size_t n) {
        void *q = nullptr;
        const hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) { ps.push_back(q); p = (T *)q; }
        return e;
    }
};

// All modes' roadmaps at once (k_mm_connect / k_mm_order, porrt_prm.hpp): nodes end to end, the kd pre-order rank of every node
// from the mode's kd-tree built on the device (k_seg_kd_ranks: a workgroup per mode), one count / scan / fill / order sequence, one
// download.
int porrt_ctx::roadmaps_of_modes(double max_step, double search_radius) {
    HIPCHK(hipSetDevice(device));
    size_t NT = 0, max_n = 0;
    for (const MmMode &m : mm.modes) { NT += m.xy.size() / 2; max_n = std::max(max_n, m.xy.size() / 2); }
    for (MmMode &m : mm.modes) { m.efrom.clear(); m.eto.clear(); }
    if (NT == 0) return PORRT_OK;
    if (NT + 1 >= 0x7FFFFFFFull) { set_err("too many roadmap nodes"); return PORRT_ERR_INVALID; }
    std::vector<double> hx(NT), hy(NT);
    std::vector<uint32_t> base(NT), seg(mm.modes.size() + 1, 0);
    std::vector<size_t> off(mm.modes.size() + 1, 0);
    {
        size_t at = 0;
        for (size_t mi = 0; mi < mm.modes.size(); ++mi) {
            const MmMode &m = mm.modes[mi];
            const size_t n = m.xy.size() / 2;
            off[mi] = at; seg[mi] = (uint32_t)at;
            for (size_t t = 0; t < n; ++t) { hx[at + t] = m.xy[2 * t]; hy[at + t] = m.xy[2 * t + 1]; base[at + t] = (uint32_t)at; }
            at += n;
        }
        off[mm.modes.size()] = at; seg[mm.modes.size()] = (uint32_t)at;
    }
    HIPCHK(d_radT2.reserve(max_n + 8)); HIPCHK(d_cnt.reserve(1)); HIPCHK(d_rc.reserve(1)); HIPCHK(d_cls.reserve(cls_bytes(W, H)));
    int r = layout_buffers();
    if (r) return r;
    cls_dirty = true;                                   // a re-layout loses the raster
    if ((r = build_cls())) return r;
    rad_uploaded = 0;
    if ((r = ensure_radius_table(max_step, search_radius, max_n + 2))) return r;
    memset(&rc, 0, sizeof rc);
    rc.cls = d_cls.p; rc.clr = d_cls.p + (size_t)W * H; rc.sat = opt_box_table ? (const uint32_t *)(d_cls.p + cls_sat_offset(W, H)) : nullptr; rc.W = W; rc.H = H; rc.low0 = low[0]; rc.low1 = low[1]; rc.ppm = ppm; rc.domain = domain; rc.has_grid = has_grid;
    rc.n_validities = n_validities;
    for (int i = 0; i < n_validities; ++i) rc.validities[i] = validities[i];
    rc.all_worlds = ones(n_worlds);
    rc.visibility = visibility;
    rc.rad_T2 = d_radT2.p;
    HIPCHK(hipMemcpyAsync(d_rc.p, &rc, sizeof rc, hipMemcpyHostToDevice, stream));
    GrowScratch &sc = mm_scratch;                        // kept by the context: the next call finds its buffers
    double *dx = nullptr, *dy = nullptr;
    uint32_t *dbase = nullptr, *drank = nullptr, *ddeg = nullptr, *derr = nullptr, *dtmp = nullptr, *dfrom = nullptr, *dto = nullptr, *dseg = nullptr, *daux = nullptr, *dsz = nullptr;
    int *dchild = nullptr, *dpar = nullptr;
    unsigned long long *doff = nullptr, *dtot = nullptr;
    HIPCHK(sc.get(0, dx, NT)); HIPCHK(sc.get(1, dy, NT)); HIPCHK(sc.get(2, dbase, NT)); HIPCHK(sc.get(3, drank, NT)); HIPCHK(sc.get(4, ddeg, NT)); HIPCHK(sc.get(5, derr, 1));
    HIPCHK(sc.get(6, doff, NT + 1)); HIPCHK(sc.get(7, dtot, (NT + kScanTile - 1) / kScanTile + 2));
    HIPCHK(sc.get(11, dseg, seg.size())); HIPCHK(sc.get(12, dchild, 2 * NT)); HIPCHK(sc.get(13, dpar, NT)); HIPCHK(sc.get(14, daux, NT)); HIPCHK(sc.get(15, dsz, NT));
    HIPCHK(hipMemcpyAsync(dx, hx.data(), NT * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dy, hy.data(), NT * 8, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dbase, base.data(), NT * 4, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemcpyAsync(dseg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice, stream));
    HIPCHK(hipMemsetAsync(derr, 0, sizeof(uint32_t), stream));
    MmConst p{};
    p.NT = (uint32_t)NT; p.x = dx; p.y = dy; p.base = dbase; p.rank = drank; p.rad_T2 = d_radT2.p; p.deg = ddeg; p.edge_off = doff; p.err = derr;
    ScopedEvents<2> evs;
    HIPCHK(evs.create());
    HIPCHK(hipEventRecord(evs.e[0], stream));
    const dim3 wgrid((unsigned)((NT + 3) / 4)), block(256);
    // every mode's kd-tree and the pre-order ranks of its nodes (what orders a node's neighbours), a workgroup per mode
    hipLaunchKernelGGL(k_seg_kd_ranks, dim3((unsigned)mm.modes.size()), block, 0, stream, (const double *)dx, (const double *)dy, (const uint32_t *)dseg, dchild, dpar, daux, dsz, drank);
    hipLaunchKernelGGL(k_mm_connect<false>, wgrid, block, 0, stream, (const RunConst *)d_rc.p, p);
    bg_scan(ddeg, NT, dtot, doff, stream);
    unsigned long long E = 0;
    HIPCHK(hipMemcpyAsync(&E, doff + NT, sizeof E, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    if (E >= 0xFFFFFFFFull) { set_err("multi-modal PRM: edge list too long"); return PORRT_ERR_CAPACITY; }
    HIPCHK(sc.get(8, dtmp, E)); HIPCHK(sc.get(9, dfrom, E)); HIPCHK(sc.get(10, dto, E));
    p.tmp = dtmp; p.efrom = dfrom; p.eto = dto;
    hipLaunchKernelGGL(k_mm_connect<true>, wgrid, block, 0, stream, (const RunConst *)d_rc.p, p);
    hipLaunchKernelGGL(k_mm_order, wgrid, block, 0, stream, p);
    HIPCHK(hipEventRecord(evs.e[1], stream));
    std::vector<uint32_t> hfrom(E), hto(E);
    std::vector<unsigned long long> hoff(NT + 1);
    uint32_t h_err = 0;
    if (E) {
        HIPCHK(hipMemcpyAsync(hfrom.data(), dfrom, E * 4, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(hto.data(), dto, E * 4, hipMemcpyDeviceToHost, stream));
    }
    HIPCHK(hipMemcpyAsync(hoff.data(), doff, (NT + 1) * 8, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipMemcpyAsync(&h_err, derr, sizeof h_err, hipMemcpyDeviceToHost, stream));
    HIPCHK(hipStreamSynchronize(stream));
    HIPCHK(hipGetLastError());
    if (h_err & ERR_RASTER) { set_err("raster access the reference would panic on (image::get_pixel out of range, door pixel without zone, two zones on one segment)"); return PORRT_ERR_RASTER; }
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, evs.e[0], evs.e[1]));
    mm.device_s += 1e-3 * (double)ms;
    for (size_t mi = 0; mi < mm.modes.size(); ++mi) {
        const size_t e0 = (size_t)hoff[off[mi]], e1 = (size_t)hoff[off[mi + 1]];
        mm.modes[mi].efrom.assign(hfrom.begin() + e0, hfrom.begin() + e1);
        mm.modes[mi].eto.assign(hto.begin() + e0, hto.begin() + e1);
    }
    return PORRT_OK;
}

int porrt_ctx::grow_mm_prm(const double start[2], const double *initial_belief, uint32_t n_worlds_in, double max_step, double search_radius,
                           uint64_t n_iter_per_belief) {
    using namespace mmprm;
    mm.clear();
    if (!has_grid || domain != PORRT_DOMAIN_SHELF || n_zones == 0) { set_err("multi-modal PRM: a shelf domain with zones (porrt_set_grid + porrt_set_zones)"); return PORRT_ERR_INVALID; }
    if (!start || !initial_belief || n_worlds_in != (uint32_t)n_worlds) { set_err("multi-modal PRM: start / belief of n_worlds entries"); return PORRT_ERR_INVALID; }
    if (!(max_step > 0.0) || !(search_radius > 0.0)) { set_err("bad PRM parameters"); return PORRT_ERR_INVALID; }
    {
        double sum = 0.0;
        for (uint32_t w = 0; w < n_worlds_in; ++w) sum += initial_belief[w];
        if (std::fabs(sum - 1.0) > 1e-6) { set_err("belief state does not sum to 1 (check_belief_state)"); return PORRT_ERR_INVALID; }
    }
    const double t0 = now_s();
    const uint32_t nw = n_worlds_in;
    mm.nw = nw;
    // reachable_belief_states (:331): only their number enters the growth (the sample budget)
    {
        BeliefSpace bs;
        bs.domain = domain; bs.nz = n_zones; bs.nw = nw; bs.validities = validities;
        std::string e;
        const int r = bs.reach_from(initial_belief, e);
        if (r) { set_err(e); return r; }
        mm.n_beliefs = bs.size();
    }
    const Pcg64 planner_sampler = crng;          // never advanced by the planner: every mode clones this state
    Pcg64 zone_sampler;                          // ContinuousSampler::new([0, 0], [visibility, 2 pi]) (:302)
    zone_sampler.seed_from_u64(0);
    std::unordered_map<uint64_t, size_t> mode_hash_map;          // hash -> the last mode inserted with it (:160)
    auto add_mode = [&](const std::vector<int> &remaining, double reach_p, const std::vector<double> &belief) {      // :135-164
        MmMode m;
        mode_hash_map[BeliefSpace::hash_of(belief.data(), nw)] = mm.modes.size();
        m.belief = belief; m.reaching_probability = reach_p; m.remaining = remaining;
        m.there.assign(64, -1); m.not_there.assign(64, -1);
        m.sampler = planner_sampler;
        mm.modes.push_back(std::move(m));
        return mm.modes.size() - 1;
    };
    auto mode_of_hash = [&](uint64_t h) -> int64_t {
        const auto it = mode_hash_map.find(h);
        return it == mode_hash_map.end() ? -1 : (int64_t)it->second;
    };
    auto add_sample = [&](size_t mode, double x, double y) -> uint64_t {      // PRM::add_sample: the node; its edges come later, on the GPU
        mm.modes[mode].xy.push_back(x); mm.modes[mode].xy.push_back(y);
        return mm.modes[mode].xy.size() / 2 - 1;
    };
    auto get_transitions = [&](size_t mode_id, int zone, size_t out[2]) -> int {      // :183-282
        int n_out = 0;
        if (is_final(mm.modes[mode_id].belief)) return 0;
        for (int pass = 0; pass < 2; ++pass) {                   // object there, then object not there
            std::vector<int64_t> &map = pass == 0 ? mm.modes[mode_id].there : mm.modes[mode_id].not_there;
            if (map[zone] >= 0) { out[n_out++] = (size_t)map[zone]; continue; }
            const MmMode &mode = mm.modes[mode_id];
            std::vector<double> sb;
            double reach_p;
            if (pass == 0) {
                sb.assign(nw, 0.0);
                sb[zone] = 1.0;
                normalize(sb);
                reach_p = mode.reaching_probability * transition_probability(mode.belief, sb);
            } else {
                sb = mode.belief;
                sb[zone] = 0.0;
                reach_p = mode.reaching_probability * transition_probability(mode.belief, sb);
                double sum = 0.0;
                for (double v : sb) sum = sum + v;
                if (!(sum > 0.0)) return -1;                     // the reference asserts
                normalize(sb);
            }
            int64_t succ = mode_of_hash(BeliefSpace::hash_of(sb.data(), nw));
            if (succ < 0) {
                std::vector<int> remaining;
                for (int z : mode.remaining) if (z != zone) remaining.push_back(z);
                succ = (int64_t)add_mode(remaining, reach_p, sb);
                int goal_zone = -1;
                if (pass == 0) goal_zone = zone;
                else for (uint32_t w = 0; w < nw; ++w) if (sb[w] == 1.0) { goal_zone = (int)w; break; }
                if (goal_zone >= 0) mm.modes[succ].finals.push_back(add_sample((size_t)succ, zone_pos[goal_zone][0], zone_pos[goal_zone][1]));     // initial goal state
            }
            MmTransition t;
            t.zone = (uint32_t)zone; t.from = (uint32_t)mode_id; t.to = (uint32_t)succ; t.observation = 1;      // sic: `true` in both branches
            mm.tr.push_back(std::move(t));
            (pass == 0 ? mm.modes[mode_id].there : mm.modes[mode_id].not_there)[zone] = (int64_t)mm.tr.size() - 1;
            out[n_out++] = mm.tr.size() - 1;
        }
        return n_out;
    };
    {
        std::vector<int> remaining(n_zones);
        for (int z = 0; z < n_zones; ++z) remaining[z] = z;
        add_mode(remaining, 1.0, std::vector<double>(initial_belief, initial_belief + nw));
        add_sample(0, start[0], start[1]);                       // :339 planning start
    }
    const uint64_t total = n_iter_per_belief * mm.n_beliefs;
    const uint64_t n_outer = (uint64_t)((double)total / 200.0);
    const double two_pi = 2.0 * 3.14159265358979323846;          // 2.0 * PI as f64
    for (uint64_t i = 0; i < n_outer; ++i) {
        const size_t mode_id = (size_t)drng.gen_range_usize(mm.modes.size());          // :351
        for (int s = 0; s < 190; ++s) {                                                // :357 grow_graph(.., 190)
            MmMode &m = mm.modes[mode_id];
            const double x = m.sampler.gen_range_f64(s_low[0], s_up[0]);
            const double y = m.sampler.gen_range_f64(s_low[1], s_up[1]);
            add_sample(mode_id, x, y);
        }
        for (int j = 0; j < 10; ++j) {                                                 // :360-389
            if (mm.modes[mode_id].remaining.empty()) continue;
            const size_t zi = (size_t)drng.gen_range_usize(mm.modes[mode_id].remaining.size());
            const int zone = mm.modes[mode_id].remaining[zi];
            size_t tids[2];
            const int nt = get_transitions(mode_id, zone, tids);
            if (nt < 0) { set_err("belief without mass (the reference asserts)"); mm.clear(); return PORRT_ERR_INVALID; }
            // sample_observation_of_zone (:482-493): the radius draw is made and discarded; libm as Rust's f64::cos / sin
            (void)zone_sampler.gen_range_f64(0.0, visibility);
            const double angle = zone_sampler.gen_range_f64(0.0, two_pi);
            // (cos and sin of one operand: LLVM -- rustc as well as this compiler -- merges them into one sincos call on glibc; said explicitly)
            double sn, cs;
            ::sincos(angle, &sn, &cs);
            double ts[2] = {zone_pos[zone][0] + visibility * cs, zone_pos[zone][1] + visibility * sn};
            for (int d = 0; d < 2; ++d) {                                              // f64::clamp(low, up - 0.0001)
                const double lo = s_low[d], hi = s_up[d] - 0.0001;
                if (ts[d] < lo) ts[d] = lo;
                if (ts[d] > hi) ts[d] = hi;
            }
            const uint64_t obs = add_sample(mode_id, ts[0], ts[1]);
            for (int k = 0; k < nt; ++k) {
                MmTransition &t = mm.tr[tids[k]];
                const uint64_t dst = add_sample(t.to, ts[0], ts[1]);
                t.pairs.push_back(obs); t.pairs.push_back(dst);
            }
        }
    }
    mm.host_s = now_s() - t0;
    // the modes' roadmaps, all at once on the GPU
    const double t1 = now_s();
    mm.device_s = 0;
    if (getenv("PORRT_MM_ONE_BY_ONE")) {           // developer switch: one launch sequence per mode (the first version; same result)
        for (MmMode &m : mm.modes) {
            const int r = roadmap_of_points(m.xy, max_step, search_radius, m.efrom, m.eto, mm.device_s);
            if (r) { mm.clear(); return r; }
        }
    } else {
        const int r = roadmaps_of_modes(max_step, search_radius);
        if (r) { mm.clear(); return r; }
    }
    mm.roadmap_s = now_s() - t1;
    have_results = false;                        // the context's single-graph getters do not describe a mode tree
    mm.valid = true;
    return PORRT_OK;
}

int64_t porrt_ctx::prm_plan_path(const double start[2], const double goal[2], double *path_xy, uint64_t cap) {
    if (!have_results || mode != PORRT_MODE_PRM) { set_err("plan_path: grow a roadmap first (porrt_grow_prm)"); return PORRT_ERR_INVALID; }
    if (!start || !goal) { set_err("plan_path: start and goal"); return PORRT_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    int r = download(DL_TREE);
    if (r) return r;
    if ((r = ensure_edge_order())) return r;                        // PTOGraph::parents in push order, on the device
    const size_t N = n_nodes, E2 = 2 * (size_t)counters.n_edges;
    const HostKd *kd = host_kd_of(this);
    const size_t kd_start = kd->nearest(start[0], start[1]), kd_goal = kd->nearest(goal[0], goal[1]);
    if (prm.w_cap < E2 + 1) {
        if (prm.d_w) (void)hipFree(prm.d_w);
        prm.d_w = nullptr; prm.w_cap = 0;
        HIPCHK(hipMalloc((void **)&prm.d_w, (E2 + E2 / 8 + 1) * sizeof(double)));
        prm.w_cap = E2 + E2 / 8 + 1;
        prm.w_tag = ~0ull;
    }
    if (prm.dist_cap < N) {
        void *drop[] = {prm.d_dist, prm.d_dirty[0], prm.d_dirty[1]};
        for (void *q : drop) if (q) (void)hipFree(q);
        prm.d_dist = nullptr; prm.d_dirty[0] = prm.d_dirty[1] = nullptr; prm.dist_cap = 0;
        HIPCHK(hipMalloc((void **)&prm.d_dist, (N + N / 8 + 1) * sizeof(double)));
        HIPCHK(hipMalloc((void **)&prm.d_dirty[0], N + N / 8 + 1));
        HIPCHK(hipMalloc((void **)&prm.d_dirty[1], N + N / 8 + 1));
        prm.dist_cap = N + N / 8 + 1;
    }
    if (!prm.d_flags) HIPCHK(hipMalloc((void **)&prm.d_flags, 8 * sizeof(uint32_t)));
    const dim3 grid((unsigned)((N + 255) / 256)), block(256);
    if (prm.w_tag != results_tag) {
        hipLaunchKernelGGL(k_prm_weights, grid, block, 0, stream, (uint32_t)N, (const unsigned long long *)eo.d_adj_off, (const uint32_t *)eo.d_adj_id,
                           (const double *)d_nx.p, (const double *)d_ny.p, prm.d_w);
        prm.w_tag = results_tag;
    }
    // dijkstra from the goal's node (pto_graph.rs:275-303): sweeps until nothing changes, eight between two looks
    HIPCHK(hipMemsetAsync(prm.d_dirty[0], 0, N, stream));
    HIPCHK(hipMemsetAsync(prm.d_dirty[1], 0, N, stream));
    hipLaunchKernelGGL(k_prm_sssp_init, grid, block, 0, stream, (uint32_t)N, (uint32_t)kd_goal, (const unsigned long long *)eo.d_adj_off,
                       (const uint32_t *)eo.d_adj_id, prm.d_dist, prm.d_dirty[0], prm.d_dirty[1]);
    int cur = 1;                                                    // the init marked the goal's neighbours in buffer 1
    uint32_t h_flags[8];
    for (uint64_t sweeps = 0;; sweeps += 8) {
        HIPCHK(hipMemsetAsync(prm.d_flags, 0, sizeof h_flags, stream));
        for (uint32_t k = 0; k < 8; ++k, cur ^= 1)
            hipLaunchKernelGGL(k_prm_sssp_sweep, grid, block, 0, stream, (uint32_t)N, (const unsigned long long *)eo.d_adj_off, (const uint32_t *)eo.d_adj_id,
                               (const double *)prm.d_w, prm.d_dist, prm.d_dirty[cur], prm.d_dirty[cur ^ 1], prm.d_flags, k);
        HIPCHK(hipMemcpyAsync(h_flags, prm.d_flags, sizeof h_flags, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        if (!h_flags[7]) break;
        if (sweeps > 16u * 1000u * 1000u) { set_err("plan_path: no fixpoint"); return PORRT_ERR_DEVICE; }
    }
    std::vector<double> dist(N);
    HIPCHK(hipMemcpy(dist.data(), prm.d_dist, N * sizeof(double), hipMemcpyDeviceToHost));
    if (std::isinf(dist[kd_start])) return 0;                       // prm.rs:117-119: an empty path
    // extract_path (pto_graph.rs:305-326): from the start always to the first parent of least cost-to-goal + edge; the
    // parents lists of the few path nodes are read from the device adjacency
    uint64_t n_path = 0;
    size_t node = kd_start;
    std::vector<uint32_t> par;
    for (size_t guard = 0;; ++guard) {
        if (n_path < cap && path_xy) { path_xy[2 * n_path] = h_nx[node]; path_xy[2 * n_path + 1] = h_ny[node]; }
        ++n_path;
        if (dist[node] == 0.0) break;
        if (guard > N) { set_err("plan_path: zero-length cycle (the reference would not terminate)"); return PORRT_ERR_INVALID; }
        unsigned long long off[2];
        HIPCHK(hipMemcpy(off, eo.d_adj_off + node, sizeof off, hipMemcpyDeviceToHost));
        par.resize(off[1] - off[0]);
        if (!par.empty()) HIPCHK(hipMemcpy(par.data(), eo.d_adj_id + off[0], par.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        size_t best = 0;
        double best_cost = 0.0;
        bool have = false;
        for (uint32_t p2 : par) {                                   // min_by: the first minimum over the parents list
            const double cost = dist[p2] + HostKd::norm2(h_nx[p2], h_ny[p2], h_nx[node], h_ny[node]);
            if (!have || cost < best_cost) { best = p2; best_cost = cost; have = true; }
        }
        if (!have) { set_err("plan_path: node without parents"); return PORRT_ERR_INVALID; }
        node = best;
    }
    return (int64_t)n_path;
}

// Best path cost without downloading the tree (k_best_cost).  1 = found, 0 = no final node, -1 = scratch too small
// (the caller then walks on the host), other negatives = errors.
int porrt_ctx::read_best_cost(double *cost, uint64_t *final_id) {
    BestCost r;
    HIPCHK(hipMemcpy(&r, d_bcout.p, sizeof r, hipMemcpyDeviceToHost));
    if (r.overflow) return -1;
    if (r.final_id == 0xFFFFFFFFu) return 0;
    double c;
    memcpy(&c, &r.cost_bits, 8);
    if (cost) *cost = c;
    if (final_id) *final_id = r.final_id;
    return 1;
}

int porrt_ctx::best_cost_device(double *cost, uint64_t *final_id) {
    if (!have_results) { set_err("no results: call porrt_grow first"); return PORRT_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemsetAsync(d_bccursor.p, 0, sizeof(uint32_t), stream));
    hipLaunchKernelGGL(k_best_cost, dim3(1, 1), dim3(1024), 0, stream, (const RunConst *)d_rc.p, (uint32_t)n_steps);
    HIPCHK(hipStreamSynchronize(stream));
    return read_best_cost(cost, final_id);
}

// A member's bookkeeping after the leader of a porrt_grow_batch ran the steps: n_iter_done iterations in `steps` steps
// (the budget, or where the loop condition of rrt.rs:109 / pto.rs:67 ended this member).
int porrt_ctx::finish_batch_member(uint64_t n_iter_done, uint32_t steps, uint32_t own_steps, float device_ms) {
    // fetched by grow_batch for all members at once (batch_hc, batch_nodes)
    const Counters hc = batch_hc;
    const uint32_t n_final_nodes = batch_nodes;
    if (hc.err & ERR_CAND_OVERFLOW) return -100;
    if (hc.err & ERR_RNG_RETRY) { set_err("a float draw would have been redrawn: grow this context on its own"); return PORRT_ERR_INVALID; }
    counters = hc;
    if (getenv("PORRT_DEBUG_ALL") && hc.tim[10]) fprintf(stderr, "[porrt] member %u: longest k_kd_claim workgroup %.1f us, mean %.1f us\n", batch_slot, 1e-2 * (double)hc.tim[2], hc.tim[8] ? 1e-2 * (double)hc.tim[0] / (double)hc.tim[8] : 0.0);
    if (getenv("PORRT_DEBUG") && batch_slot == 0) {
        fprintf(stderr, "[porrt] batch member 0: samples served through the lists in memory %u; kd claim losers max %u mean %.1f over %u launches; g_nd %u; deferred ties %u (pooled ids %u)\n",
                hc.n_heavy, hc.dbg[0], hc.dbg[2] ? (double)hc.dbg[1] / hc.dbg[2] : 0.0, hc.dbg[2], hc.dbg[3], hc.pend_cnt, hc.pool_n);
        for (int t = 0; t < 8; ++t)
            if (hc.tim[t + 8]) fprintf(stderr, "[porrt] phase %d: %.2f us per wave over %llu waves (total %.1f wave-ms)\n", t, 1e-2 * (double)hc.tim[t] / (double)hc.tim[t + 8],
                                       (unsigned long long)hc.tim[t + 8], 1e-5 * (double)hc.tim[t]);
    }
    n_iter = n_iter_done;
    n_steps = steps;                 // n_at[steps] is this member's final size (carried along by k_row_sched if it ended earlier)
    n_nodes = n_final_nodes;
    const unsigned long long all = rc.all_worlds;
    complete = mode == PORRT_MODE_PTO ? (hc.n_final > 0 && (hc.finality & all) == all) : hc.n_final > 0;
    have_results = true;
    ++results_tag;
    const uint64_t calls = n_iter_done - n_iter_done / 100;
    if (has_inj) inj_pos += calls;
    else crng.advance((u128)2 * calls);
    if (mode == PORRT_MODE_PTO && n_iter_done < batch_world_draws) {
        // the worlds of the whole plan were drawn up front (one per iteration, pto.rs:142); the loop ended earlier: the discrete
        // sampler stands where the reference's would -- after n_iter_done draws
        if (has_inj_worlds) inj_wpos = batch_wpos0 + n_iter_done;
        else { drng = batch_drng0; for (uint64_t t = 0; t < n_iter_done; ++t) (void)drng.gen_range_usize((uint64_t)n_worlds); }
    }
    memset(&metrics, 0, sizeof metrics);
    metrics.n_iter = n_iter;
    metrics.n_nodes = n_nodes;
    metrics.n_steps = own_steps;
    metrics.n_tie_fallbacks = hc.tie_fallbacks + ((hc.err & ERR_GPATH_OVERFLOW) ? 1 : 0);
    metrics.device_s = device_ms * 1e-3;
    metrics.total_s = metrics.device_s;
    if (hc.err & ERR_RASTER) { set_err("raster access outside the map, door pixel without zone id, or two zones on one segment (the reference panics here)"); return PORRT_ERR_RASTER; }
    if (hc.err & ERR_EDGE_OVERFLOW) { set_err("edge pool overflow"); return PORRT_ERR_CAPACITY; }
    if (hc.err & (ERR_TIE_POOL | ERR_PAGE_OVERFLOW)) { set_err("deferred-tie pool / region page pool overflow"); return PORRT_ERR_CAPACITY; }
    if (mode == PORRT_MODE_PTO && !complete) { set_err("final nodes are not reached for each world"); return PORRT_INCOMPLETE; }
    return PORRT_OK;
}

// `want` streams of the current device of which any two run their kernels side by side, found by trying: a pair of single-wave
// kernels that each wait 150 us takes as long as one of them on two hardware queues and twice that on one.  Fewer than `want` (the caller then keeps
// its own streams) if 4 * want candidates do not hold such a set.
static std::vector<hipStream_t> pick_parallel_streams(uint32_t want) {
    std::vector<hipStream_t> chosen, rejected;
    auto pair_s = [](hipStream_t a, hipStream_t b) {
        (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
        const double t0 = now_s();
        hipLaunchKernelGGL(k_wait_us, dim3(1), dim3(64), 0, a, 150u);
        hipLaunchKernelGGL(k_wait_us, dim3(1), dim3(64), 0, b, 150u);
        (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
        return now_s() - t0;
    };
    double t_one = 0.0;         // one such kernel alone (launch and wait included): what "side by side" is measured against
    for (uint32_t tries = 0; chosen.size() < want && tries < 4u * want; ++tries) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) break;
        hipLaunchKernelGGL(k_wait_us, dim3(1), dim3(64), 0, st, 1u);          // first use: the stream gets its queue
        if (t_one == 0.0) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipStreamSynchronize(st);
                const double t0 = now_s();
                hipLaunchKernelGGL(k_wait_us, dim3(1), dim3(64), 0, st, 150u);
                (void)hipStreamSynchronize(st);
                best = std::min(best, now_s() - t0);
            }
            t_one = best;
        }
        bool ok = true;
        for (hipStream_t c : chosen) {
            const double t = std::min(pair_s(c, st), pair_s(c, st));
            if (t > 1.6 * t_one) { ok = false; break; }           // two on one queue: 2 x
        }
        (ok ? chosen : rejected).push_back(st);
    }
    for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
    if (getenv("PORRT_DEBUG")) fprintf(stderr, "[porrt] sub-batch streams: %zu of %u found side by side (%zu candidates shared a queue)\n", chosen.size(), want, rejected.size());
    if (chosen.size() < want) { for (hipStream_t st : chosen) (void)hipStreamDestroy(st); chosen.clear(); }
    return chosen;
}

// porrt_grow_batch: the same growth for several contexts of one device at once, each with its own n_iter_min / n_iter_max
// (nmin[q], nmax[q]) and the loop condition of rrt.rs:109 / pto.rs:67.
// Context 0 leads: every step kernel is launched once with one grid row per context, so the contexts' dependent
// load chains overlap inside each kernel instead of queueing behind each other.  When every member has the same fixed budget
// (nmin == nmax, the same for all) the launch's (i0, nb) serve every row; otherwise each row follows its own step plan on the
// device (RunConst::sched_*, k_sched_init) and k_row_sched takes it out of the later launches once its loop condition ends it --
// results are those of separate porrt_grow calls either way.
static int grow_batch(porrt_ctx *const *cs, uint32_t n, const double *starts, double max_step, double search_radius, const uint64_t *nmin_in,
                      const uint64_t *nmax_in, uint32_t K, int mode) {
    porrt_ctx *L = cs[0];
    if (K == 0 || K > 4096) { L->set_err("batch_K must be in 1..4096"); return PORRT_ERR_INVALID; }
    if (mode != PORRT_MODE_RRT && mode != PORRT_MODE_PTO) { L->set_err("bad mode"); return PORRT_ERR_INVALID; }
    if (!(max_step > 0.0) || !(search_radius >= 0.0)) { L->set_err("max_step / search_radius"); return PORRT_ERR_INVALID; }
    std::vector<uint64_t> nmin(nmin_in, nmin_in + n), nmax(nmax_in, nmax_in + n);
    uint64_t n_iter = 0;                         // the longest budget: what the preparation kernels' grids cover
    bool sched = false;
    for (uint32_t q = 0; q < n; ++q) {
        if (nmax[q] < nmin[q]) nmax[q] = nmin[q];                                  // as porrt_grow
        if (nmax[q] == 0 || nmax[q] + 2 >= 0x7FFFFFF0ull) { L->set_err("n_iter_min / n_iter_max"); return PORRT_ERR_INVALID; }
        if (nmin[q] != nmax[q] || nmax[q] != nmax[0]) sched = true;
        n_iter = std::max(n_iter, nmax[q]);
    }
    // the rows' plans as the device will hold them (k_sched_init): steps up to n_iter_min, then up to n_iter_max
    auto plan = [&](uint32_t q, uint32_t b, uint64_t &i0, uint32_t &nb) {
        const uint64_t s1 = (nmin[q] + K - 1) / K;
        if (b < s1) { i0 = (uint64_t)b * K; nb = (uint32_t)std::min<uint64_t>(K, nmin[q] - i0); }
        else { i0 = std::min<uint64_t>(nmin[q] + (uint64_t)(b - s1) * K, nmax[q]); nb = (uint32_t)std::min<uint64_t>(K, nmax[q] - i0); }
    };
    uint32_t B_pot = 0, first_dec = 0xFFFFFFFFu;
    for (uint32_t q = 0; q < n; ++q) {
        const uint64_t s1 = (nmin[q] + K - 1) / K, s2 = (nmax[q] - nmin[q] + K - 1) / K;
        B_pot = std::max<uint32_t>(B_pot, (uint32_t)(s1 + s2));
        first_dec = std::min<uint32_t>(first_dec, (uint32_t)s1);
    }
    std::vector<uint32_t> nbmax(sched ? B_pot : 0u, 0u);
    if (sched)
        for (uint32_t q = 0; q < n; ++q)
            for (uint32_t b2 = 0; b2 < B_pot; ++b2) { uint64_t i0; uint32_t nb; plan(q, b2, i0, nb); nbmax[b2] = std::max(nbmax[b2], nb); }
    for (uint32_t q = 0; q < n; ++q) {
        if (!cs[q] || cs[q]->device != L->device) { L->set_err("porrt_grow_batch: contexts of one device"); return PORRT_ERR_INVALID; }
        for (uint32_t r = 0; r < q; ++r) if (cs[r] == cs[q]) { L->set_err("porrt_grow_batch: a context appears twice"); return PORRT_ERR_INVALID; }
        if (mode == PORRT_MODE_PTO && !cs[q]->has_grid) { cs[q]->set_err("PTO mode needs a grid"); return PORRT_ERR_INVALID; }
        cs[q]->have_results = false;
        ++cs[q]->results_tag;
        // what was derived from the previous graph goes with it, as in porrt_grow (the belief graph's CSR and the costs are
        // sized for the old node count)
        cs[q]->bg.release();
        cs[q]->dp.release();
        cs[q]->batch_leader = nullptr;
    }
    HIPCHK_CTX(L, hipSetDevice(L->device));
    { int r = L->ensure_side_stream(); if (r) return r; }
    for (porrt_ctx *m : L->batch_members) if (m && m->batch_leader == L) m->batch_leader = nullptr;      // the previous batch is over
    L->batch_members.clear();
    const uint64_t batch_gen = ++L->batch_gen_counter;
    std::vector<Pcg64> c0(n), d0(n);
    std::vector<size_t> ip0(n), iw0(n);
    for (uint32_t q = 0; q < n; ++q) { c0[q] = cs[q]->crng; d0[q] = cs[q]->drng; ip0[q] = cs[q]->inj_pos; iw0[q] = cs[q]->inj_wpos; }
    if (L->rcarr_cap < n) {
        if (L->d_rcarr) (void)hipFree(L->d_rcarr);
        L->d_rcarr = nullptr; L->rcarr_cap = 0;
        HIPCHK_CTX(L, hipMalloc(&L->d_rcarr, (size_t)n * sizeof(RunConst)));
        L->rcarr_cap = n;
        if (L->graph_exec) { (void)hipGraphExecDestroy(L->graph_exec); L->graph_exec = nullptr; }
    }
    const uint32_t vwords = (K + 63) / 64;
    const bool host_dbg = getenv("PORRT_DEBUG_HOST") != nullptr;       // where the host's time goes in a batch (developer)
    for (int attempt = 0; attempt < 12; ++attempt) {
        const double th0 = now_s();
        // every member: buffers, run constants, root, samples
        for (uint32_t q = 0; q < n; ++q) {
            hipStream_t own = cs[q]->stream;
            cs[q]->stream = L->stream;          // the members' preparation is queued on the leader's stream: no host sync
            int r = cs[q]->grow_once(starts + 2 * q, max_step, search_radius, nmax[q], nmax[q], K, mode, false, 1);
            cs[q]->stream = own;
            if (r) { if (cs[q] != L) L->set_err(cs[q]->err); return r; }
            if (cs[q]->run_lds_bytes != L->run_lds_bytes) { L->set_err("porrt_grow_batch: the contexts' rasters need different LDS tiles (max_step * ppm differs)"); return PORRT_ERR_INVALID; }
        }
        const double th1 = now_s();
        L->opt_group = L->opt_group_req < 0 ? (n >= 8 ? 16u : 0u) : (uint32_t)L->opt_group_req;
        L->kd_lazy = L->opt_kd_lazy && L->opt_group != 0 && mode == PORRT_MODE_RRT && !L->opt_kd_after;
        L->kd_built_after = 0;
        // (pipelined steps search step b + 1 while step b is connected: not with rows that may end after step b)
        L->pipe_on = mode == PORRT_MODE_RRT && L->opt_group == 0 && L->opt_pipeline != 0 && !sched;
        L->lag_on = false;
        L->rc_staging.resize(n);                 // one upload for all members (the vector outlives the copy: it is a member)
        for (uint32_t q = 0; q < n; ++q) {
            RunConst &rq = cs[q]->rc;
            rq.q_stride = L->pipe_on ? rq.part_stride : 0u;
            rq.sched_i0 = sched ? cs[q]->d_sched_i0.p : nullptr;
            rq.sched_nb = sched ? cs[q]->d_sched_nb.p : nullptr;
            rq.sched_min = (uint32_t)nmin[q]; rq.sched_max = (uint32_t)nmax[q]; rq.sched_K = K;
            rq.hint_max = n >= 8 ? kHintMaxSquares : 0xFFFFFFFFu;
            rq.kd_lazy = (L->opt_kd_lazy && L->opt_group != 0 && mode == PORRT_MODE_RRT && !L->opt_kd_after) ? 1u : 0u;
            rq.sched_steps = (uint32_t)std::min<uint64_t>((uint64_t)B_pot + 2, cs[q]->d_sched_nb.n);
            if (sched && (uint64_t)(nmin[q] + K - 1) / K + (nmax[q] - nmin[q] + K - 1) / K + 2 > cs[q]->d_sched_nb.n) { L->set_err("porrt_grow_batch: step plan"); return PORRT_ERR_DEVICE; }
            L->rc_staging[q] = rq;
        }
        HIPCHK_CTX(L, hipMemcpyAsync(L->d_rcarr, L->rc_staging.data(), (size_t)n * sizeof(RunConst), hipMemcpyHostToDevice, L->stream));
        if (mode == PORRT_MODE_RRT) {            // the members' preparation, all at once (k_batch_prep)
            const RunConst *rows = L->d_rcarr;
            hipLaunchKernelGGL(k_batch_prep, dim3(32, n), dim3(256), 0, L->stream, rows);
            if (sched) hipLaunchKernelGGL(k_sched_init, dim3(1, n), dim3(256), 0, L->stream, rows);
            hipLaunchKernelGGL(k_init_root, dim3(1, n), dim3(1), 0, L->stream, rows, 0.0, 0.0, 0ull, 0, 1);
            hipLaunchKernelGGL(k_gen_samples, dim3((unsigned)((n_iter + 255) / 256), n), dim3(256), 0, L->stream, rows, (const PcgJump *)nullptr, 0ull,
                               (unsigned long long)n_iter, 0ull, 0ull, 0ull, 0ull, 0ull);
            hipLaunchKernelGGL(k_sort_samples, dim3(sched ? B_pot : (unsigned)((n_iter + K - 1) / K), n), dim3(256), 0, L->stream, rows, 0u, 0ull, (unsigned long long)n_iter, K);
        } else if (sched) {
            hipLaunchKernelGGL(k_sched_init, dim3(1, n), dim3(256), 0, L->stream, (const RunConst *)L->d_rcarr);
        }
        // the leader: all steps, one hipGraph (or eager), grid rows = contexts
        L->launch_rcp = L->d_rcarr;
        L->launch_Q = n;
        L->side_active = false;
        L->pipe_near_done = 0xFFFFFFFFu;
        L->commit_pend_b = 0xFFFFFFFFu;
        L->kd_b0 = 0; L->kd_last_b = 0; L->kd_last_nb = 0; L->kd_gidx = 0; L->kd_hint_ns = 0;
        L->kd_pend[0] = L->kd_pend[1] = L->kd_pend[2] = false;
        L->kd_group = kd_group_for(K, L->opt_kd_group, n);
        ScopedEvents<2> bevs;
        HIPCHK_CTX(L, bevs.create());
        hipEvent_t e0 = bevs.e[0], e1 = bevs.e[1];
        HIPCHK_CTX(L, hipEventRecord(e0, L->stream));
        size_t ev_used = 0;
        const bool prof = L->opt_profile;
        if (prof) {
            const size_t want = (size_t)(n_iter / K + 6) * 4 + 4;
            while (L->ev_pool.size() < want) {
                hipEvent_t ev;
                HIPCHK_CTX(L, hipEventCreate(&ev));
                L->ev_pool.push_back(ev);
            }
        }
        // rows with plans of their own: step by step, k_row_sched before every step at which a row's loop may end; the host reads
        // the number of rows still running two steps behind its launches and stops when it is zero (the steps launched past that
        // point find every row ended and do nothing)
        int sched_rc = PORRT_OK;
        auto sched_steps = [&]() -> uint32_t {
            constexpr uint32_t LAG = 2, RING = 4;
            if (L->active_cap < B_pot + 2u) {
                if (L->d_active) (void)hipFree(L->d_active);
                if (L->h_active) (void)hipHostFree(L->h_active);
                L->d_active = nullptr; L->h_active = nullptr; L->active_cap = 0;
                if (hipMalloc((void **)&L->d_active, (B_pot + 2u) * sizeof(uint32_t)) != hipSuccess ||
                    hipHostMalloc((void **)&L->h_active, (B_pot + 2u) * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { sched_rc = PORRT_ERR_DEVICE; return 0; }
                L->active_cap = B_pot + 2u;
            }
            ScopedEvents<RING> ring;
            if (ring.create() != hipSuccess) { sched_rc = PORRT_ERR_DEVICE; return 0; }
            (void)hipMemsetAsync(L->d_active, 0, (B_pot + 2u) * sizeof(uint32_t), L->stream);
            // Rows that have ended still cost every later launch their workgroups' start and exit (a TAMP-shaped batch: most rows
            // end at n_iter_min, a few run on to n_iter_max).  When the count the host last saw has fallen to three quarters of the
            // rows launched, the rows that still have work are gathered into a compact array of run constants and the step kernels
            // are launched on that one (k_rows_compact / k_rows_gather; the count is two steps old, so never too small: rows only end).
            // Not with a kd chain beside the steps (its kernels are launched on the same rows, from another stream).
            const bool can_compact = L->opt_compact && n >= 64u && (L->kd_lazy || mode == PORRT_MODE_PTO);
            if (can_compact && L->rcarr_c_cap < n) {
                for (int k = 0; k < 2; ++k) { if (L->d_rcarr_c[k]) (void)hipFree(L->d_rcarr_c[k]); L->d_rcarr_c[k] = nullptr; }
                if (L->d_live_idx) (void)hipFree(L->d_live_idx);
                L->d_live_idx = nullptr; L->rcarr_c_cap = 0;
                if (hipMalloc((void **)&L->d_rcarr_c[0], (size_t)n * sizeof(RunConst)) != hipSuccess || hipMalloc((void **)&L->d_rcarr_c[1], (size_t)n * sizeof(RunConst)) != hipSuccess ||
                    hipMalloc((void **)&L->d_live_idx, (size_t)n * sizeof(uint32_t)) != hipSuccess) { sched_rc = PORRT_ERR_DEVICE; return 0; }
                L->rcarr_c_cap = n;
            }
            int pp = 0;
            L->n_compactions = 0;
            uint32_t cb = 0;
            for (; cb < B_pot; ++cb) {
                if (cb >= first_dec) {
                    hipLaunchKernelGGL(k_row_sched, dim3((n + 63u) / 64u), dim3(64), 0, L->stream, (const RunConst *)L->d_rcarr, n, cb, L->d_active);
                    (void)hipMemcpyAsync(L->h_active + cb, L->d_active + cb, sizeof(uint32_t), hipMemcpyDeviceToHost, L->stream);
                    (void)hipEventRecord(ring.e[cb % RING], L->stream);
                }
                if (can_compact && cb >= first_dec + LAG + 1u) {
                    // h_active[cb - LAG - 1] was read when step cb - 1 was launched: rows running then, an upper bound of the rows with work now
                    const uint32_t seen = L->h_active[cb - LAG - 1u], slots = std::min<uint32_t>(n, (seen + 7u) & ~7u);
                    if (slots && slots * 4u <= L->launch_Q * 3u) {
                        hipLaunchKernelGGL(k_rows_compact, dim3(1), dim3(1024), 0, L->stream, (const RunConst *)L->d_rcarr, n, cb, slots, L->d_live_idx);
                        hipLaunchKernelGGL(k_rows_gather, dim3(slots), dim3(256), 0, L->stream, (const RunConst *)L->d_rcarr, (const uint32_t *)L->d_live_idx, L->d_rcarr_c[pp]);
                        L->launch_rcp = L->d_rcarr_c[pp];
                        L->launch_Q = slots;
                        pp ^= 1;
                        ++L->n_compactions;
                    }
                }
                L->launch_step(cb, 0u, nbmax[cb], vwords, L->run_lds_bytes, prof, ev_used, 0u, 0u);
                if (cb >= first_dec + LAG) {
                    if (hipEventSynchronize(ring.e[(cb - LAG) % RING]) != hipSuccess) { sched_rc = PORRT_ERR_DEVICE; break; }
                    if (L->h_active[cb - LAG] == 0u) { ++cb; break; }
                }
            }
            L->launch_rcp = L->d_rcarr;          // what follows the steps (the last rewire commit, the kd structure where a tie asks, the results) takes every row
            L->launch_Q = n;
            L->join_side();
            return cb;
        };
        auto all_steps = [&]() {
            uint64_t ci = 0;
            uint32_t cb = 0;
            while (ci < n_iter) {
                const uint32_t nb = (uint32_t)std::min<uint64_t>(K, n_iter - ci);
                const uint64_t i2 = ci + nb;                                                 // start of step cb + 1
                const uint32_t nb2 = i2 < n_iter ? (uint32_t)std::min<uint64_t>(K, n_iter - i2) : 0;
                L->launch_step(cb, (uint32_t)ci, nb, vwords, L->run_lds_bytes, prof, ev_used, (uint32_t)i2, nb2);
                ci += nb;
                ++cb;
            }
            L->join_side();
            return cb;
        };
        const double th2 = now_s();
        uint32_t steps = 0;
        if (sched) {
            steps = sched_steps();
            if (sched_rc) { L->set_err("porrt_grow_batch: step schedule (device)"); return sched_rc; }
        } else if (L->opt_graph && !prof && !L->sub_eager) {
            const uint64_t key[6] = {(uint64_t)mode, K, n_iter, L->run_lds_bytes, (uint64_t)(uintptr_t)L->launch_rcp, L->kd_group | ((uint64_t)L->opt_early_wave << 8) | ((uint64_t)n << 32) | ((uint64_t)L->opt_group << 48) | ((uint64_t)L->pipe_on << 56)};
            if (!L->graph_exec || memcmp(key, L->graph_key, sizeof key)) {
                if (L->graph_exec) { (void)hipGraphExecDestroy(L->graph_exec); L->graph_exec = nullptr; }
                hipGraph_t g = nullptr;
                HIPCHK_CTX(L, hipStreamBeginCapture(L->stream, hipStreamCaptureModeThreadLocal));
                (void)all_steps();
                HIPCHK_CTX(L, hipStreamEndCapture(L->stream, &g));
                HIPCHK_CTX(L, hipGraphInstantiate(&L->graph_exec, g, nullptr, nullptr, 0));
                (void)hipGraphDestroy(g);
                memcpy(L->graph_key, key, sizeof key);
            }
            HIPCHK_CTX(L, hipGraphLaunch(L->graph_exec, L->stream));
            steps = (uint32_t)((n_iter + K - 1) / K);
        } else {
            steps = all_steps();
        }
        const double th3 = now_s();
        HIPCHK_CTX(L, hipEventRecord(e1, L->stream));
        // every member's counters and final tree size: gathered on the device, one copy, one sync for all
        if (L->batch_out_cap < n) {
            if (L->d_batch_out) (void)hipFree(L->d_batch_out);
            L->d_batch_out = nullptr; L->batch_out_cap = 0;
            HIPCHK_CTX(L, hipMalloc((void **)&L->d_batch_out, (size_t)n * sizeof(BatchOut)));
            L->batch_out_cap = n;
        }
        hipLaunchKernelGGL(k_batch_gather, dim3(n), dim3(64), 0, L->stream, (const RunConst *)L->d_rcarr, sched ? steps : (uint32_t)((n_iter + K - 1) / K), L->d_batch_out);
        std::vector<BatchOut> h_out(n);
        HIPCHK_CTX(L, hipMemcpyAsync(h_out.data(), L->d_batch_out, (size_t)n * sizeof(BatchOut), hipMemcpyDeviceToHost, L->stream));
        HIPCHK_CTX(L, hipStreamSynchronize(L->stream));
        if (L->kd_lazy) {
            bool need = false;
            for (uint32_t q = 0; q < n; ++q) need = need || (h_out[q].cnt.n_lca && !(h_out[q].cnt.err & (ERR_CAND_OVERFLOW | ERR_RNG_RETRY)));
            if (need) {
                // a tie between two nodes off the goal path in some row: the whole kd structure after all (every row of the launch:
                // the kernels take rows as they come), then the records that waited for it
                // (for the steps up to the last such tie of any row: a record names older nodes only)
                uint32_t S = 0;
                for (uint32_t q = 0; q < n; ++q) S = std::max(S, h_out[q].cnt.lca_next);
                S = std::min(S, steps);
                std::vector<std::pair<uint32_t, uint32_t>> segs;
                if (sched || S < steps) segs.emplace_back(S - 1u, K);                          // (rows cut their own steps: row_nb)
                else segs.emplace_back(steps - 1u, (uint32_t)(n_iter - (uint64_t)(steps - 1u) * K));
                // ... built for the rows that asked, not for the batch: in a batch of hundreds some row nearly always does, and the build
                // of a row that has no such tie orders nothing (its records were settled from the goal path)
                std::vector<uint32_t> asking;
                for (uint32_t q = 0; q < n; ++q)
                    if (h_out[q].cnt.n_lca && !(h_out[q].cnt.err & (ERR_CAND_OVERFLOW | ERR_RNG_RETRY))) asking.push_back(q);
                const bool subset = asking.size() < n;
                if (subset) {
                    if (L->rcarr_c_cap < n) {
                        for (int k = 0; k < 2; ++k) { if (L->d_rcarr_c[k]) (void)hipFree(L->d_rcarr_c[k]); L->d_rcarr_c[k] = nullptr; }
                        if (L->d_live_idx) (void)hipFree(L->d_live_idx);
                        L->d_live_idx = nullptr; L->rcarr_c_cap = 0;
                        HIPCHK_CTX(L, hipMalloc((void **)&L->d_rcarr_c[0], (size_t)n * sizeof(RunConst)));
                        HIPCHK_CTX(L, hipMalloc((void **)&L->d_rcarr_c[1], (size_t)n * sizeof(RunConst)));
                        HIPCHK_CTX(L, hipMalloc((void **)&L->d_live_idx, (size_t)n * sizeof(uint32_t)));
                        L->rcarr_c_cap = n;
                    }
                    HIPCHK_CTX(L, hipMemcpyAsync(L->d_live_idx, asking.data(), asking.size() * sizeof(uint32_t), hipMemcpyHostToDevice, L->stream));
                    hipLaunchKernelGGL(k_rows_gather, dim3((unsigned)asking.size()), dim3(256), 0, L->stream, (const RunConst *)L->d_rcarr, (const uint32_t *)L->d_live_idx, L->d_rcarr_c[0]);
                    L->launch_rcp = L->d_rcarr_c[0];
                    L->launch_Q = (uint32_t)asking.size();
                }
                int r = L->kd_full_build(segs);
                if (subset) {
                    HIPCHK_CTX(L, hipStreamSynchronize(L->stream));        // (`asking` is the copy's source until it has run)
                    L->launch_rcp = L->d_rcarr;
                    L->launch_Q = n;
                }
                if (r) { L->set_err("porrt_grow_batch: kd structure after the steps"); return r; }
                L->kd_built_after = 1;
                if (host_dbg) fprintf(stderr, "[porrt] batch of %u: kd structure after the steps for %zu rows\n", n, asking.size());
                hipLaunchKernelGGL(k_batch_gather, dim3(n), dim3(64), 0, L->stream, (const RunConst *)L->d_rcarr, sched ? steps : (uint32_t)((n_iter + K - 1) / K), L->d_batch_out);
                HIPCHK_CTX(L, hipMemcpyAsync(h_out.data(), L->d_batch_out, (size_t)n * sizeof(BatchOut), hipMemcpyDeviceToHost, L->stream));
                HIPCHK_CTX(L, hipStreamSynchronize(L->stream));
            }
        }
        const double th4 = now_s();
        for (uint32_t q = 0; q < n; ++q) { cs[q]->batch_hc = h_out[q].cnt; cs[q]->batch_nodes = h_out[q].nodes; }
        {
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { L->set_err(std::string("kernel launch: ") + hipGetErrorString(e)); return PORRT_ERR_DEVICE; }
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        L->launch_rcp = L->d_rc.p;
        L->launch_Q = 1;
        bool retry = false;
        int worst = PORRT_OK;
        for (uint32_t q = 0; q < n; ++q) {
            // where the member's loop ended: its own stop (k_row_sched), or the whole plan
            uint64_t it_q = nmax[q];
            uint32_t steps_q = steps;
            if (sched && h_out[q].cnt.sched_stop != 0xFFFFFFFFu) { it_q = h_out[q].cnt.sched_iter; steps_q = h_out[q].cnt.sched_stop; }
            int r = cs[q]->finish_batch_member(it_q, steps, steps_q, ms);
            if (r == -100) retry = true;
            else if (r < 0) { if (cs[q] != L) L->set_err(cs[q]->err); return r; }
            else if (r > worst) worst = r;
        }
        if (host_dbg) fprintf(stderr, "[porrt] batch of %u (host): members %.2f ms, upload + preparation launches %.2f ms, step launches %.2f ms (%u steps), "
                                      "wait + gather (+ kd build) %.2f ms, finish %.2f ms; device %.2f ms\n", n, 1e3 * (th1 - th0), 1e3 * (th2 - th1), 1e3 * (th3 - th2), steps,
                              1e3 * (th4 - th3), 1e3 * (now_s() - th4), (double)ms);
        if (!retry && prof && sched && getenv("PORRT_DEBUG_STEPS")) {
            for (uint32_t s2 = 0; s2 < steps && (size_t)(4 * s2 + 3) < ev_used; ++s2) {
                float a = 0, r2 = 0;
                (void)hipEventElapsedTime(&a, L->ev_pool[4 * s2 + 0], L->ev_pool[4 * s2 + 1]);
                (void)hipEventElapsedTime(&r2, L->ev_pool[4 * s2 + 2], L->ev_pool[4 * s2 + 3]);
                fprintf(stderr, "[porrt] batch step %u near %.1f us connect %.1f us\n", s2, a * 1e3, r2 * 1e3);
            }
        }
        if (!retry && prof && !sched) {
            // events were recorded around k_near [0,1] and the connect kernel [2,3] of every step (all members at once);
            // the leader's metrics carry the batch totals
            double scan = 0, conn = 0, pairs = 0, bytes = 0;
            uint64_t launches = 0;
            for (uint32_t s = 0; s < steps && (size_t)(4 * s + 3) < ev_used; ++s) {
                float a = 0, r2 = 0;
                (void)hipEventElapsedTime(&a, L->ev_pool[4 * s + 0], L->ev_pool[4 * s + 1]);
                (void)hipEventElapsedTime(&r2, L->ev_pool[4 * s + 2], L->ev_pool[4 * s + 3]);
                scan += a * 1e-3;
                conn += r2 * 1e-3;
                ++launches;
                if (getenv("PORRT_DEBUG_STEPS")) fprintf(stderr, "[porrt] batch step %u near %.1f us connect %.1f us\n", s, a * 1e3, r2 * 1e3);
            }
            std::vector<uint32_t> nat(steps + 1);
            for (uint32_t q = 0; q < n; ++q) {
                HIPCHK_CTX(L, hipMemcpy(nat.data(), cs[q]->d_nat.p, (steps + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
                uint64_t it = 0;
                for (uint32_t s = 0; s < steps; ++s) {
                    const uint64_t nbq = std::min<uint64_t>(K, n_iter - it);
                    it += nbq;
                    pairs += 2.0 * (double)nbq * (double)nat[s];
                    bytes += 16.0 * (double)nat[s] + (16.0 + 20.0) * (double)nbq;      // 16 N_b + 36 K
                }
            }
            L->metrics.scan_s = scan; L->metrics.connect_s = conn; L->metrics.scan_launches = launches;
            L->metrics.scan_pairs = pairs; L->metrics.scan_bytes = bytes;
        }
        if (!retry) {
            for (uint32_t q = 0; q < n; ++q) { cs[q]->batch_leader = L; cs[q]->batch_slot = q; cs[q]->batch_size = n; cs[q]->batch_gen = batch_gen; }
            L->batch_members.assign(cs, cs + n);
            return worst;
        }
        for (uint32_t q = 0; q < n; ++q) {          // neighbour lists overflowed somewhere: regrow them everywhere and replay
            cs[q]->opt_cand_cap = (uint32_t)std::min<uint64_t>((uint64_t)cs[q]->opt_cand_cap * 4, nmax[q] + 2);
            cs[q]->crng = c0[q]; cs[q]->drng = d0[q]; cs[q]->inj_pos = ip0[q]; cs[q]->inj_wpos = iw0[q];
            cs[q]->have_results = false;
        }
    }
    L->set_err("neighbour list capacity");
    return PORRT_ERR_CAPACITY;
}

extern "C" {

porrt_ctx *porrt_create(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    porrt_ctx *c = new porrt_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
    // (the side stream is made when the context first leads a launch sequence: the thousand members of a large batch never do)
    if (hipEventCreateWithFlags(&c->ev_step_done, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_kd[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_kd[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_kd[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_steered, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { delete c; return nullptr; }
    c->crng.seed_from_u64(0);   // sample_space.rs:18
    c->drng.seed_from_u64(0);   // sample_space.rs:47
    c->validities[0] = 1;       // map_io.rs:108-111 init_without_zones
    memset(&c->metrics, 0, sizeof c->metrics);
    memset(&c->counters, 0, sizeof c->counters);
    for (int w = 0; w < 64; ++w) c->w2g[w][0] = c->w2g[w][1] = 0.0;
    return c;
}

void porrt_destroy(porrt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (hipStream_t st : c->sub_streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipStream_t st : c->dl_streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (void *q : c->dl_pin) if (q) (void)hipHostFree(q);
    if (c->dl_out_stream) { (void)hipStreamSynchronize(c->dl_out_stream); (void)hipStreamDestroy(c->dl_out_stream); }
    if (c->d_tree_out) (void)hipFree(c->d_tree_out);
    // a batch leader going away takes its RunConst array with it: its members must not look for it any more
    for (porrt_ctx *m : c->batch_members) if (m && m != c && m->batch_leader == c) m->batch_leader = nullptr;
    if (c->batch_leader && c->batch_leader != c)
        for (porrt_ctx *&m : c->batch_leader->batch_members) if (m == c) m = nullptr;
    if (c->arena.base) (void)hipFree(c->arena.base);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    if (c->ev_step_done) (void)hipEventDestroy(c->ev_step_done);
    for (int p2 = 0; p2 < 3; ++p2) if (c->ev_kd[p2]) (void)hipEventDestroy(c->ev_kd[p2]);
    if (c->ev_steered) (void)hipEventDestroy(c->ev_steered);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->d_rcarr) (void)hipFree(c->d_rcarr);
    if (c->d_batch_out) (void)hipFree(c->d_batch_out);
    c->mm_scratch.free_all();
    for (int k = 0; k < 2; ++k) if (c->d_rcarr_c[k]) (void)hipFree(c->d_rcarr_c[k]);
    if (c->d_live_idx) (void)hipFree(c->d_live_idx);
    if (c->d_active) (void)hipFree(c->d_active);
    if (c->h_active) (void)hipHostFree(c->h_active);

    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *porrt_last_error(const porrt_ctx *c) { return c ? c->err.c_str() : "null context (no HIP device?)"; }

int porrt_set_grid(porrt_ctx *c, const uint8_t *occ, uint32_t W, uint32_t H, const double low[2], const double up[2], int domain) {
    if (!c) return PORRT_ERR_INVALID;
    if (!occ || W == 0 || H == 0 || !(up[0] > low[0])) { c->set_err("set_grid: empty grid or bad bounds"); return PORRT_ERR_INVALID; }
    if (domain != PORRT_DOMAIN_SHELF && domain != PORRT_DOMAIN_DOOR) { c->set_err("set_grid: bad domain"); return PORRT_ERR_INVALID; }
    c->occ.assign(occ, occ + (size_t)W * H);
    c->zones.clear();
    c->W = W; c->H = H;
    c->low[0] = low[0]; c->low[1] = low[1];
    c->ppm = (double)W / (up[0] - low[0]);      // map_shelves_io.rs:89
    c->domain = domain;
    c->has_grid = true;
    c->cls_dirty = true;
    c->n_zones = 0; c->n_worlds = 1; c->n_validities = 1; c->validities[0] = 1; c->visibility = 0.0;
    if (c->goal_kind == 2) c->goal_kind = 0;
    return PORRT_OK;
}

int porrt_set_zones(porrt_ctx *c, const uint8_t *zone_ids, double visibility) {
    if (!c) return PORRT_ERR_INVALID;
    if (!c->has_grid || !zone_ids) { c->set_err("set_zones: call set_grid first"); return PORRT_ERR_INVALID; }
    const size_t n = (size_t)c->W * c->H;
    int max_id = 0;
    for (size_t p = 0; p < n; ++p)
        if (zone_ids[p] != 255 && zone_ids[p] > max_id) max_id = zone_ids[p];
    const int nz = max_id + 1;                   // map_shelves_io.rs:116-130
    if (nz > 64 || (c->domain == PORRT_DOMAIN_DOOR && nz > 6)) { c->set_err("set_zones: too many zones (64 shelves / 6 doors)"); return PORRT_ERR_INVALID; }
    // integer centroids in u32 arithmetic, then to_coordinates with its swapped offsets (map_shelves_io.rs:132-148,172-177)
    std::vector<uint32_t> si(nz, 0), sj(nz, 0), cnt(nz, 0);
    for (uint32_t i = 0; i < c->H; ++i)
        for (uint32_t j = 0; j < c->W; ++j) {
            uint8_t z = zone_ids[(size_t)i * c->W + j];
            if (z != 255) { si[z] += i; sj[z] += j; cnt[z]++; }
        }
    for (int z = 0; z < nz; ++z) {
        if (cnt[z] == 0) { c->set_err("set_zones: zone id without pixels (the reference divides by zero)"); return PORRT_ERR_INVALID; }
        uint32_t ci = si[z] / cnt[z], cj = sj[z] / cnt[z];
        c->zone_pos[z][0] = (double)cj / c->ppm + c->low[1];
        c->zone_pos[z][1] = (double)(c->H - 1 - ci) / c->ppm + c->low[0];
    }
    c->zones.assign(zone_ids, zone_ids + n);
    c->n_zones = nz;
    c->visibility = visibility;
    if (c->domain == PORRT_DOMAIN_SHELF) {       // map_shelves_io.rs:106-114
        c->n_worlds = nz;
        c->n_validities = 1;
        c->validities[0] = ones(nz);
    } else {                                     // map_io.rs:113-128,198-214
        c->n_worlds = 1 << nz;
        for (int z = 0; z < nz; ++z) {
            uint64_t m = 0;
            for (int w = 0; w < c->n_worlds; ++w)
                if (w & (1 << z)) m |= 1ULL << w;
            c->validities[z] = m;
        }
        c->validities[nz] = ones(c->n_worlds);
        c->n_validities = nz + 1;
    }
    c->cls_dirty = true;
    return PORRT_OK;
}

int porrt_set_sampler(porrt_ctx *c, const double low[2], const double up[2], uint64_t seed) {
    if (!c) return PORRT_ERR_INVALID;
    for (int i = 0; i < 2; ++i)
        if (!(low[i] < up[i]) || !std::isfinite(low[i]) || !std::isfinite(up[i])) { c->set_err("set_sampler: low >= up"); return PORRT_ERR_INVALID; }
    for (int i = 0; i < 2; ++i) { c->s_low[i] = low[i]; c->s_up[i] = up[i]; }
    c->crng.seed_from_u64(seed);
    c->drng.seed_from_u64(seed);
    c->has_inj = false; c->inj_xy.clear(); c->inj_pos = 0;
    c->has_inj_worlds = false; c->inj_worlds.clear(); c->inj_wpos = 0;
    return PORRT_OK;
}

int porrt_set_discrete_seed(porrt_ctx *c, uint64_t seed) {
    if (!c) return PORRT_ERR_INVALID;
    c->drng.seed_from_u64(seed);
    return PORRT_OK;
}

int porrt_set_samples(porrt_ctx *c, const double *xy, size_t n) {
    if (!c || (!xy && n)) return PORRT_ERR_INVALID;
    c->inj_xy.assign(xy, xy + 2 * n);
    c->inj_pos = 0;
    c->has_inj = true;
    c->inj_dirty = true;
    return PORRT_OK;
}

int porrt_set_worlds(porrt_ctx *c, const uint32_t *worlds, size_t n) {
    if (!c || (!worlds && n)) return PORRT_ERR_INVALID;
    c->inj_worlds.assign(worlds, worlds + n);
    c->inj_wpos = 0;
    c->has_inj_worlds = true;
    return PORRT_OK;
}

int porrt_set_square_goal(porrt_ctx *c, const double *centers, const uint64_t *masks, uint32_t G, double l1_radius) {
    if (!c) return PORRT_ERR_INVALID;
    if (!centers || !masks || G == 0 || G > 64) { c->set_err("set_square_goal: need 1..64 goals"); return PORRT_ERR_INVALID; }
    double w2g[64][2];
    for (int w = 0; w < 64; ++w) {              // common.rs:310-333
        w2g[w][0] = w2g[w][1] = 0.0;
        bool has = false;
        for (uint32_t g = 0; g < G; ++g)
            if ((masks[g] >> w) & 1) {
                if (has) { c->set_err("set_square_goal: validities overlap"); return PORRT_ERR_INVALID; }
                w2g[w][0] = centers[2 * g]; w2g[w][1] = centers[2 * g + 1];
                has = true;
            }
    }
    memcpy(c->w2g, w2g, sizeof w2g);
    c->goal_kind = 1; c->G = G; c->g_l1 = l1_radius;
    for (uint32_t g = 0; g < G; ++g) { c->gcx[g] = centers[2 * g]; c->gcy[g] = centers[2 * g + 1]; c->gmask[g] = masks[g]; }
    return PORRT_OK;
}

int porrt_set_observation_goal(porrt_ctx *c, uint32_t zone_id) {
    if (!c) return PORRT_ERR_INVALID;
    if (c->zones.empty() || (int)zone_id >= c->n_zones) { c->set_err("set_observation_goal: unknown zone"); return PORRT_ERR_INVALID; }
    c->goal_kind = 2; c->obs_zone = zone_id;
    return PORRT_OK;
}

int porrt_grow(porrt_ctx *c, const double start[2], double max_step, double search_radius, uint64_t n_iter_min, uint64_t n_iter_max,
               uint32_t batch_K, int mode) {
    if (!c || !start) return PORRT_ERR_INVALID;
    return abi_guard([&]() { return c->grow(start, max_step, search_radius, n_iter_min, n_iter_max, batch_K, mode); });
}

// joins the threads of a scope on every way out of it (an exception between emplace and join would otherwise terminate)
struct ThreadJoiner {
    std::vector<std::thread> th;
    ~ThreadJoiner() { for (auto &t : th) if (t.joinable()) t.join(); }
};

static int grow_batch_each(porrt_ctx *const *ctxs, uint32_t n_ctx, const double *starts, double max_step, double search_radius, const uint64_t *n_iter_min,
                           const uint64_t *n_iter_max, uint32_t batch_K, int mode) {
    if (!ctxs || !n_ctx || !starts || !ctxs[0] || !n_iter_min || !n_iter_max) return PORRT_ERR_INVALID;
    for (uint32_t q = 0; q < n_ctx; ++q) if (!ctxs[q]) return PORRT_ERR_INVALID;
    uint32_t G = ctxs[0]->opt_batch_streams ? ctxs[0]->opt_batch_streams : (n_ctx >= 32 ? 2u : 1u);
    G = std::min(G, n_ctx);
    if (G <= 1) { ctxs[0]->last_launch_mode = 0; return grow_batch(ctxs, n_ctx, starts, max_step, search_radius, n_iter_min, n_iter_max, batch_K, mode); }
    for (uint32_t q = 0; q < n_ctx; ++q)          // checked here for the whole call: the sub-batches only see their own members
        for (uint32_t r = 0; r < q; ++r) if (ctxs[r] == ctxs[q]) { ctxs[0]->set_err("porrt_grow_batch: a context appears twice"); return PORRT_ERR_INVALID; }
    // contiguous runs of the argument, led by their first member; every run is a complete porrt_grow_batch of its own (its own
    // streams, hipGraph, results), driven from a host thread of its own so that the launch sequences are fed side by side
    std::vector<int> rcs(G, PORRT_OK);
    std::vector<uint32_t> lo(G + 1);
    for (uint32_t g = 0; g <= G; ++g) lo[g] = (uint32_t)((uint64_t)n_ctx * g / G);
    // Each sub-batch needs a main and a side stream that run side by side with the other sub-batches' -- and the runtime spreads
    // streams over a handful of hardware queues by rules of its own: two streams on one queue take turns (the contexts' own
    // streams do about every other time: 140 instead of 160 M expansions/s on the bench).  So the streams are chosen by
    // measurement, once per leading context: pick_parallel_streams.
    porrt_ctx *top = ctxs[0];
    if (hipSetDevice(top->device) != hipSuccess) { top->set_err("hipSetDevice"); return PORRT_ERR_DEVICE; }
    const uint32_t per = top->opt_kd_inline ? 1u : 2u;       // streams per sub-batch (main + side, or main alone)
    if (top->sub_streams.size() < per * G && top->sub_streams_tried != per * G) {
        for (hipStream_t st : top->sub_streams) (void)hipStreamDestroy(st);
        top->sub_streams = pick_parallel_streams(per * G);
        if (top->sub_streams.size() < per * G) top->sub_streams_tried = per * G;      // (e.g. under a profiler that serialises kernels: every pair test fails)
    }
    const bool have_streams = top->sub_streams.size() >= per * G;
    top->last_launch_mode = have_streams ? (int)G : -(int)G;
    auto part = [&](uint32_t g) {
        porrt_ctx *Lg = ctxs[lo[g]];
        hipStream_t own = Lg->stream, own2 = Lg->stream2;
        bool swapped2 = false;
        // On these streams the steps are launched one by one: a replayed hipGraph puts its branches on streams of the runtime's
        // choosing, and two replays side by side then share a queue more often than not (135-141 against 160 M expansions/s);
        // the launches (~900 per sub-batch) stay ahead of the GPU from a host thread each.
        if (have_streams) { Lg->stream = top->sub_streams[g]; if (per == 2u) { Lg->stream2 = top->sub_streams[G + g]; swapped2 = true; } Lg->sub_eager = true; }
        Lg->opt_kd_inline = top->opt_kd_inline;
        Lg->opt_compact = top->opt_compact;
        rcs[g] = grow_batch(ctxs + lo[g], lo[g + 1] - lo[g], starts + 2 * (size_t)lo[g], max_step, search_radius, n_iter_min + lo[g], n_iter_max + lo[g], batch_K, mode);
        // the side stream goes back only if it was lent one of the measured streams: a leader that had none and made its own inside
        // grow_batch (ensure_side_stream: no measured set, or kd_inline) keeps it for its lifetime instead of dropping it every call
        Lg->stream = own; if (swapped2) Lg->stream2 = own2; Lg->sub_eager = false;
    };
    {
        ThreadJoiner tj;
        std::vector<uint32_t> inline_parts;                 // (a host that cannot start another thread: those sub-batches run here, afterwards)
        for (uint32_t g = 1; g < G; ++g) {
            try { tj.th.emplace_back(part, g); } catch (const std::system_error &) { inline_parts.push_back(g); }
        }
        part(0);
        for (uint32_t g : inline_parts) part(g);
    }
    int worst = PORRT_OK;
    for (uint32_t g = 0; g < G; ++g) {
        if (rcs[g] < 0) { if (g) ctxs[0]->set_err(ctxs[lo[g]]->err); return rcs[g]; }
        worst = std::max(worst, rcs[g]);
    }
    return worst;
}

int porrt_grow_batch_each(porrt_ctx *const *ctxs, uint32_t n_ctx, const double *starts, double max_step, double search_radius, const uint64_t *n_iter_min,
                          const uint64_t *n_iter_max, uint32_t batch_K, int mode) {
    return abi_guard([&]() { return grow_batch_each(ctxs, n_ctx, starts, max_step, search_radius, n_iter_min, n_iter_max, batch_K, mode); });
}

int porrt_grow_batch(porrt_ctx *const *ctxs, uint32_t n_ctx, const double *starts, double max_step, double search_radius, uint64_t n_iter_min,
                     uint64_t n_iter_max, uint32_t batch_K, int mode) {
    if (!n_ctx) return PORRT_ERR_INVALID;
    return abi_guard([&]() {
        const std::vector<uint64_t> mn(n_ctx, n_iter_min), mx(n_ctx, n_iter_max);
        return grow_batch_each(ctxs, n_ctx, starts, max_step, search_radius, mn.data(), mx.data(), batch_K, mode);
    });
}

uint64_t porrt_num_nodes(const porrt_ctx *c) { return c && c->have_results ? c->n_nodes : 0; }
uint64_t porrt_num_iterations(const porrt_ctx *c) { return c && c->have_results ? c->n_iter : 0; }

int porrt_get_tree(const porrt_ctx *cc, double *xy, int64_t *parent, double *dist_root) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c) return PORRT_ERR_INVALID;
    int r = c->download(porrt_ctx::DL_TREE | (dist_root ? porrt_ctx::DL_DIST : 0u));
    if (r) return r;
    for (size_t j = 0; j < c->n_nodes; ++j) {
        if (xy) { xy[2 * j] = c->h_nx[j]; xy[2 * j + 1] = c->h_ny[j]; }
        if (parent) parent[j] = c->h_parent[j];
        if (dist_root) dist_root[j] = c->h_dist[j];
    }
    return PORRT_OK;
}

// ---- host ranges the caller has pinned for porrt_get_trees (porrt_host_pin): start -> (bytes, the device's address of the start)
namespace {
struct PinnedRange { size_t bytes; char *dev; };
std::mutex g_pin_mutex;
std::map<uintptr_t, PinnedRange> g_pins;
// the device's address of host address p if [p, p + bytes) lies inside one pinned range, else null
void *pinned_dev_ptr(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    auto it = g_pins.upper_bound((uintptr_t)p);
    if (it == g_pins.begin()) return nullptr;
    --it;
    const uintptr_t off = (uintptr_t)p - it->first;
    if (off > it->second.bytes || bytes > it->second.bytes - off) return nullptr;
    return it->second.dev + off;
}
} // namespace

int porrt_host_pin(void *p, size_t bytes) {
    if (!p || !bytes) return PORRT_ERR_INVALID;
    return abi_guard([&]() {
        std::lock_guard<std::mutex> lk(g_pin_mutex);
        if (g_pins.count((uintptr_t)p)) return (int)PORRT_ERR_INVALID;
        if (hipHostRegister(p, bytes, hipHostRegisterMapped) != hipSuccess) { (void)hipGetLastError(); return (int)PORRT_ERR_DEVICE; }
        void *d = nullptr;
        if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostUnregister(p); return (int)PORRT_ERR_DEVICE; }
        g_pins[(uintptr_t)p] = PinnedRange{bytes, (char *)d};
        return (int)PORRT_OK;
    });
}

int porrt_host_unpin(void *p) {
    if (!p) return PORRT_ERR_INVALID;
    return abi_guard([&]() {
        std::lock_guard<std::mutex> lk(g_pin_mutex);
        auto it = g_pins.find((uintptr_t)p);
        if (it == g_pins.end()) return (int)PORRT_ERR_INVALID;
        g_pins.erase(it);
        if (hipHostUnregister(p) != hipSuccess) { (void)hipGetLastError(); return (int)PORRT_ERR_DEVICE; }
        return (int)PORRT_OK;
    });
}

// porrt_get_tree for many contexts of one device at once (the trees of a porrt_grow_batch): a few worker threads, each with a
// pinned staging slot and a copy stream of its own, fetch the trees straight from the device arrays (four asynchronous copies
// per tree at the link's speed) and lay them out in the caller's arrays while the other workers' copies are in flight.  The
// contexts' cached host copies are not touched.  xy / parent / dist_root: n pointers each (an entry or a whole array may be NULL).
int porrt_get_trees(porrt_ctx *const *ctxs, uint32_t n_ctx, double *const *xy, int64_t *const *parent, double *const *dist_root) {
    if (!ctxs || !n_ctx || !ctxs[0]) return PORRT_ERR_INVALID;
    porrt_ctx *top = ctxs[0];
    size_t maxN = 0;
    for (uint32_t q = 0; q < n_ctx; ++q) {
        if (!ctxs[q] || ctxs[q]->device != top->device) { top->set_err("porrt_get_trees: contexts of one device"); return PORRT_ERR_INVALID; }
        if (!ctxs[q]->have_results || ctxs[q]->mm.valid) { top->set_err("porrt_get_trees: a context without a grown tree"); return PORRT_ERR_INVALID; }
        maxN = std::max<size_t>(maxN, ctxs[q]->n_nodes);
    }
    if (hipSetDevice(top->device) != hipSuccess) { top->set_err("hipSetDevice"); return PORRT_ERR_DEVICE; }
    // every output array inside a range the caller pinned (porrt_host_pin): one kernel writes all the trees into the caller's arrays
    {
        std::vector<TreeOut> desc(n_ctx);
        bool direct = true;
        for (uint32_t q = 0; q < n_ctx && direct; ++q) {
            const porrt_ctx *c = ctxs[q];
            const size_t N = c->n_nodes;
            TreeOut &t = desc[q];
            t.nx = c->d_nx.p; t.ny = c->d_ny.p; t.dist = c->d_distA.p; t.parent = c->d_parent.p; t.n = N;
            t.oxy = nullptr; t.odist = nullptr; t.oparent = nullptr;
            if (xy && xy[q]) { t.oxy = (double *)pinned_dev_ptr(xy[q], N * 16); direct = direct && t.oxy; }
            if (parent && parent[q]) { t.oparent = (long long *)pinned_dev_ptr(parent[q], N * 8); direct = direct && t.oparent; }
            if (dist_root && dist_root[q]) { t.odist = (double *)pinned_dev_ptr(dist_root[q], N * 8); direct = direct && t.odist; }
        }
        if (direct) {
            if (!top->dl_out_stream) {
                // the highest priority the device offers: the kernel is a few hundred workgroups that mostly wait for the link, and a
                // batch growing beside it would otherwise keep them waiting for wave slots (82 ms instead of 16 for 256 trees)
                int lo = 0, hi = 0;
                (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
                if (hipStreamCreateWithPriority(&top->dl_out_stream, hipStreamNonBlocking, hi) != hipSuccess) { top->dl_out_stream = nullptr; top->set_err("hipStreamCreate"); return PORRT_ERR_DEVICE; }
            }
            if (top->d_tree_out_cap < n_ctx) {
                if (top->d_tree_out) (void)hipFree(top->d_tree_out);
                top->d_tree_out = nullptr; top->d_tree_out_cap = 0;
                if (hipMalloc((void **)&top->d_tree_out, (size_t)n_ctx * sizeof(TreeOut)) != hipSuccess) { top->set_err("hipMalloc (tree descriptors)"); return PORRT_ERR_DEVICE; }
                top->d_tree_out_cap = n_ctx;
            }
            hipStream_t st = top->dl_out_stream;
            // a few workgroups per tree: the link, not the GPU, sets the pace, and the batch growing beside this keeps its wave slots
            if (hipMemcpyAsync(top->d_tree_out, desc.data(), (size_t)n_ctx * sizeof(TreeOut), hipMemcpyHostToDevice, st) != hipSuccess) { top->set_err("porrt_get_trees: descriptor upload"); return PORRT_ERR_DEVICE; }
            hipLaunchKernelGGL(k_trees_out, dim3(top->opt_tree_out_blocks, n_ctx), dim3(256), 0, st, (const TreeOut *)top->d_tree_out);
            if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { top->set_err("porrt_get_trees: direct write failed"); return PORRT_ERR_DEVICE; }
            return PORRT_OK;
        }
    }
    const uint32_t W = std::min<uint32_t>(top->opt_fetch_workers, n_ctx);          // (4 / 8 / 12 / 16 workers fetch 256 trees in 25 / 23 / 23.5 / 23.5 ms: the copies set the time)
    const size_t slot = (maxN * 28u + 4095u) & ~(size_t)4095u;          // nx, ny, dist_root (f64) and parent (i32) of one tree
    // every worker's slot must hold the largest tree of THIS call; slots are kept across calls, each with its own size
    if (top->dl_pin.size() < W) { top->dl_pin.resize(W, nullptr); top->dl_pin_cap.resize(W, 0); }
    for (uint32_t w = 0; w < W; ++w) {
        if (top->dl_pin[w] && top->dl_pin_cap[w] >= slot) continue;
        if (top->dl_pin[w]) { (void)hipHostFree(top->dl_pin[w]); top->dl_pin[w] = nullptr; top->dl_pin_cap[w] = 0; }
        void *q = nullptr;
        if (hipHostMalloc(&q, slot + slot / 8, hipHostMallocDefault) != hipSuccess) { top->set_err("hipHostMalloc (tree staging)"); return PORRT_ERR_DEVICE; }
        top->dl_pin[w] = q;
        top->dl_pin_cap[w] = slot + slot / 8;
    }
    while (top->dl_streams.size() < W) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { top->set_err("hipStreamCreate"); return PORRT_ERR_DEVICE; }
        top->dl_streams.push_back(st);
    }
    std::vector<int> rcs(W, PORRT_OK);
    auto work = [&](uint32_t w) {
        if (hipSetDevice(top->device) != hipSuccess) { rcs[w] = PORRT_ERR_DEVICE; return; }
        hipStream_t st = top->dl_streams[w];
        for (uint32_t q = w; q < n_ctx; q += W) {
            const porrt_ctx *c = ctxs[q];
            const size_t N = c->n_nodes;
            double *px = (double *)top->dl_pin[w], *py = px + N, *pd = py + N;
            int *pp = (int *)(pd + N);
            double *oxy = xy ? xy[q] : nullptr, *od = dist_root ? dist_root[q] : nullptr;
            int64_t *op = parent ? parent[q] : nullptr;
            hipError_t e = hipSuccess;
            if (oxy) { e = hipMemcpyAsync(px, c->d_nx.p, N * 8, hipMemcpyDeviceToHost, st); if (e == hipSuccess) e = hipMemcpyAsync(py, c->d_ny.p, N * 8, hipMemcpyDeviceToHost, st); }
            if (od && e == hipSuccess) e = hipMemcpyAsync(pd, c->d_distA.p, N * 8, hipMemcpyDeviceToHost, st);
            if (op && e == hipSuccess) e = hipMemcpyAsync(pp, c->d_parent.p, N * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { rcs[w] = PORRT_ERR_DEVICE; return; }
            if (oxy) for (size_t j = 0; j < N; ++j) { oxy[2 * j] = px[j]; oxy[2 * j + 1] = py[j]; }
            if (od) memcpy(od, pd, N * 8);
            if (op) for (size_t j = 0; j < N; ++j) op[j] = pp[j];
        }
    };
    {
        ThreadJoiner tj;
        std::vector<uint32_t> inline_work;
        for (uint32_t w = 1; w < W; ++w) {
            try { tj.th.emplace_back(work, w); } catch (const std::system_error &) { inline_work.push_back(w); }
        }
        work(0);
        for (uint32_t w : inline_work) work(w);
    }
    for (uint32_t w = 0; w < W; ++w) if (rcs[w]) { top->set_err("porrt_get_trees: device copy failed"); return rcs[w]; }
    return PORRT_OK;
}

uint64_t porrt_num_final(const porrt_ctx *c) { return c && c->have_results ? c->counters.n_final : 0; }

int porrt_get_final_ids(const porrt_ctx *cc, uint64_t *ids) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c) return PORRT_ERR_INVALID;
    int r = c->download(porrt_ctx::DL_TREE);
    if (r) return r;
    for (size_t k = 0; k < c->h_final_ids.size(); ++k) ids[k] = c->h_final_ids[k];
    return PORRT_OK;
}

int porrt_get_final_masks(const porrt_ctx *cc, uint64_t *masks) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c) return PORRT_ERR_INVALID;
    int r = c->download(porrt_ctx::DL_TREE | porrt_ctx::DL_MASKS);
    if (r) return r;
    for (size_t k = 0; k < c->h_final_ids.size(); ++k) masks[k] = c->h_finalmask[c->h_final_ids[k]];
    return PORRT_OK;
}

int porrt_get_reach(const porrt_ctx *cc, uint64_t *masks) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c) return PORRT_ERR_INVALID;
    int r = c->download(porrt_ctx::DL_MASKS);
    if (r) return r;
    for (size_t j = 0; j < c->n_nodes; ++j) masks[j] = c->h_reach[j];
    return PORRT_OK;
}

int porrt_get_node_validity(const porrt_ctx *cc, uint32_t *v) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c) return PORRT_ERR_INVALID;
    int r = c->download(porrt_ctx::DL_MASKS);
    if (r) return r;
    for (size_t j = 0; j < c->n_nodes; ++j) v[j] = c->h_vid[j];
    return PORRT_OK;
}

uint64_t porrt_num_edges(const porrt_ctx *c) { return c && c->have_results && (c->mode == PORRT_MODE_PTO || c->mode == PORRT_MODE_PRM) ? c->counters.n_edges : 0; }

int porrt_get_edges(const porrt_ctx *cc, uint32_t *from, uint32_t *to, uint32_t *validity_id) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c) return PORRT_ERR_INVALID;
    int r = c->download(porrt_ctx::DL_EDGES);
    if (r) return r;
    for (size_t e = 0; e < c->h_efrom.size(); ++e) { from[e] = c->h_efrom[e]; to[e] = c->h_eto[e]; validity_id[e] = c->h_etv[e]; }
    return PORRT_OK;
}

int porrt_is_final_set_complete(const porrt_ctx *c) { return c && c->have_results && c->complete; }
int porrt_n_worlds(const porrt_ctx *c) { return c ? c->n_worlds : 0; }
int porrt_get_validities(const porrt_ctx *c, uint64_t *masks) {
    if (!c) return PORRT_ERR_INVALID;
    for (int i = 0; i < c->n_validities; ++i) masks[i] = c->validities[i];
    return c->n_validities;
}
int porrt_get_zone_positions(const porrt_ctx *c, double *xy) {
    if (!c) return PORRT_ERR_INVALID;
    for (int z = 0; z < c->n_zones; ++z) { xy[2 * z] = c->zone_pos[z][0]; xy[2 * z + 1] = c->zone_pos[z][1]; }
    return c->n_zones;
}

// rrt.rs:183-193 (first final node of minimal path cost), 48-61, 223-227
uint64_t porrt_best_solution(const porrt_ctx *cc, double *path_xy, uint64_t cap, double *cost) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c || c->download(porrt_ctx::DL_TREE)) return 0;
    if (c->h_final_ids.empty()) return 0;
    uint64_t best_len = 0, best_id = 0;
    double best_cost = 0;
    bool have = false;
    std::vector<uint64_t> ids;
    for (uint64_t fid : c->h_final_ids) {
        ids.clear();
        for (int64_t p = (int64_t)fid; p >= 0; p = c->h_parent[p]) {
            ids.push_back((uint64_t)p);
            if (ids.size() > c->n_nodes) return 0;   // defensive: a parent cycle cannot be walked
        }
        std::reverse(ids.begin(), ids.end());
        double sum = 0.0;
        for (size_t a = 0; a + 1 < ids.size(); ++a) {
            double dx = c->h_nx[ids[a + 1]] - c->h_nx[ids[a]], dy = c->h_ny[ids[a + 1]] - c->h_ny[ids[a]];
            volatile double xx = dx * dx, yy = dy * dy;
            sum += sqrt(xx + yy);
        }
        if (!have || sum < best_cost) { have = true; best_cost = sum; best_id = fid; best_len = ids.size(); }
    }
    if (cost) *cost = best_cost;
    if (path_xy && cap >= best_len) {
        uint64_t pos = best_len;
        for (int64_t p = (int64_t)best_id; p >= 0; p = c->h_parent[p]) {
            --pos;
            path_xy[2 * pos] = c->h_nx[p];
            path_xy[2 * pos + 1] = c->h_ny[p];
        }
    }
    return best_len;
}

// the cost of porrt_best_solution's path without fetching the tree (evaluated on the device; same arithmetic, same
// first-minimum rule); falls back to the host walk when the device scratch is too small
int porrt_best_cost(const porrt_ctx *cc, double *cost, uint64_t *final_id) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c) return PORRT_ERR_INVALID;
    int r = c->best_cost_device(cost, final_id);
    if (r != -1) return r;
    double hc = 0;
    const uint64_t n = porrt_best_solution(cc, nullptr, 0, &hc);
    if (!n) return 0;
    if (cost) *cost = hc;
    if (final_id) *final_id = ~0ull;        // not tracked by the host fallback
    return 1;
}

// porrt_best_cost for the members of the last porrt_grow_batch in one kernel launch (one workgroup per context).
// costs[q] = +inf where a member has no solution.  Members of different batches (or none) are evaluated one by one.
int porrt_best_cost_batch(porrt_ctx *const *ctxs, uint32_t n_ctx, double *costs) {
    if (!ctxs || !n_ctx || !costs) return PORRT_ERR_INVALID;
    for (uint32_t q = 0; q < n_ctx; ++q) if (!ctxs[q]) return PORRT_ERR_INVALID;
    const double inf = std::numeric_limits<double>::infinity();
    // runs of the argument that are exactly some leader's last batch, in its order (porrt_grow_batch leaves one such run per
    // sub-batch): one launch per run, all of them in flight before the first wait
    std::vector<char> together(n_ctx, 0);
    std::vector<porrt_ctx *> launched;
    for (uint32_t q0 = 0; q0 < n_ctx;) {
        porrt_ctx *L = ctxs[q0]->batch_leader;
        const uint32_t m = L ? L->batch_size : 0;
        bool run = L != nullptr && m > 0 && q0 + m <= n_ctx;
        for (uint32_t r = 0; r < m && run; ++r) {
            const porrt_ctx *c = ctxs[q0 + r];
            run = c->have_results && c->batch_leader == L && c->batch_slot == r && c->n_steps == L->n_steps &&
                  c->batch_gen == L->batch_gen_counter;        // the leader's LAST batch: its RunConst array holds these members
        }
        if (!run) { ++q0; continue; }
        if (hipSetDevice(L->device) != hipSuccess) return PORRT_ERR_DEVICE;
        for (uint32_t r = 0; r < m; ++r) (void)hipMemsetAsync(ctxs[q0 + r]->d_bccursor.p, 0, sizeof(uint32_t), L->stream);
        hipLaunchKernelGGL(k_best_cost, dim3(1, m), dim3(1024), 0, L->stream, (const RunConst *)L->d_rcarr, (uint32_t)L->n_steps);
        launched.push_back(L);
        for (uint32_t r = 0; r < m; ++r) together[q0 + r] = 1;
        q0 += m;
    }
    for (porrt_ctx *L : launched) if (hipStreamSynchronize(L->stream) != hipSuccess) return PORRT_ERR_DEVICE;
    for (uint32_t q = 0; q < n_ctx; ++q) {
        double c = inf;
        int r = together[q] ? ctxs[q]->read_best_cost(&c, nullptr) : -1;
        if (r == -1) r = porrt_best_cost(ctxs[q], &c, nullptr);
        if (r < 0) return r;
        costs[q] = r ? c : inf;
    }
    return PORRT_OK;
}

int porrt_grow_prm(porrt_ctx *c, const double start[2], double max_step, double search_radius, uint64_t n_iter) {
    return c ? c->grow_prm(start, max_step, search_radius, n_iter) : PORRT_ERR_INVALID;
}

int porrt_grow_mm_prm(porrt_ctx *c, const double start[2], const double *initial_belief, uint32_t n_worlds, double max_step, double search_radius,
                      uint64_t n_iter_per_belief) {
    return c ? c->grow_mm_prm(start, initial_belief, n_worlds, max_step, search_radius, n_iter_per_belief) : PORRT_ERR_INVALID;
}
uint64_t porrt_mm_num_modes(const porrt_ctx *c) { return c && c->mm.valid ? c->mm.modes.size() : 0; }
uint64_t porrt_mm_num_transitions(const porrt_ctx *c) { return c && c->mm.valid ? c->mm.tr.size() : 0; }
uint64_t porrt_mm_num_beliefs(const porrt_ctx *c) { return c && c->mm.valid ? c->mm.n_beliefs : 0; }
int porrt_mm_get_mode(const porrt_ctx *c, uint64_t m, double *belief, double *reaching_probability, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *n_final) {
    if (!c || !c->mm.valid || m >= c->mm.modes.size()) return PORRT_ERR_INVALID;
    const MmMode &md = c->mm.modes[m];
    if (belief) memcpy(belief, md.belief.data(), md.belief.size() * sizeof(double));
    if (reaching_probability) *reaching_probability = md.reaching_probability;
    if (n_nodes) *n_nodes = md.xy.size() / 2;
    if (n_edges) *n_edges = md.efrom.size();
    if (n_final) *n_final = md.finals.size();
    return PORRT_OK;
}
int porrt_mm_get_mode_graph(const porrt_ctx *c, uint64_t m, double *xy, uint32_t *efrom, uint32_t *eto, uint64_t *final_ids) {
    if (!c || !c->mm.valid || m >= c->mm.modes.size()) return PORRT_ERR_INVALID;
    const MmMode &md = c->mm.modes[m];
    if (xy && !md.xy.empty()) memcpy(xy, md.xy.data(), md.xy.size() * sizeof(double));
    if (efrom && !md.efrom.empty()) memcpy(efrom, md.efrom.data(), md.efrom.size() * sizeof(uint32_t));
    if (eto && !md.eto.empty()) memcpy(eto, md.eto.data(), md.eto.size() * sizeof(uint32_t));
    if (final_ids && !md.finals.empty()) memcpy(final_ids, md.finals.data(), md.finals.size() * sizeof(uint64_t));
    return PORRT_OK;
}
int porrt_mm_get_transition(const porrt_ctx *c, uint64_t t, uint32_t *zone, uint32_t *from_mode, uint32_t *to_mode, int *observation, uint64_t *n_pairs) {
    if (!c || !c->mm.valid || t >= c->mm.tr.size()) return PORRT_ERR_INVALID;
    const MmTransition &tr = c->mm.tr[t];
    if (zone) *zone = tr.zone;
    if (from_mode) *from_mode = tr.from;
    if (to_mode) *to_mode = tr.to;
    if (observation) *observation = tr.observation;
    if (n_pairs) *n_pairs = tr.pairs.size() / 2;
    return PORRT_OK;
}
int porrt_mm_get_transition_pairs(const porrt_ctx *c, uint64_t t, uint64_t *pairs) {
    if (!c || !c->mm.valid || t >= c->mm.tr.size() || !pairs) return PORRT_ERR_INVALID;
    if (!c->mm.tr[t].pairs.empty()) memcpy(pairs, c->mm.tr[t].pairs.data(), c->mm.tr[t].pairs.size() * sizeof(uint64_t));
    return PORRT_OK;
}
int porrt_mm_get_seconds(const porrt_ctx *c, double *host_s, double *roadmap_s, double *device_s) {
    if (!c || !c->mm.valid) return PORRT_ERR_INVALID;
    if (host_s) *host_s = c->mm.host_s;
    if (roadmap_s) *roadmap_s = c->mm.roadmap_s;
    if (device_s) *device_s = c->mm.device_s;
    return PORRT_OK;
}

int64_t porrt_prm_plan_path(porrt_ctx *c, const double start[2], const double goal[2], double *path_xy, uint64_t cap) {
    return c ? c->prm_plan_path(start, goal, path_xy, cap) : PORRT_ERR_INVALID;
}

// ---- belief-space expansion (pto.rs:185-259)
int porrt_build_belief_graph(porrt_ctx *c, const double *start_belief, uint32_t n_worlds) {
    return c ? c->build_belief_graph(start_belief, n_worlds) : PORRT_ERR_INVALID;
}
uint64_t porrt_bg_num_beliefs(const porrt_ctx *c) { return c && c->bg.valid ? c->bg.B : 0; }
uint64_t porrt_bg_num_nodes(const porrt_ctx *c) { return c && c->bg.valid ? c->bg.N * c->bg.B : 0; }
uint64_t porrt_bg_num_edges(const porrt_ctx *c) { return c && c->bg.valid ? c->bg.n_edges : 0; }
int porrt_bg_get_beliefs(const porrt_ctx *c, double *out) {
    if (!c || !c->bg.valid || !out) return PORRT_ERR_INVALID;
    memcpy(out, c->bg.beliefs.data(), c->bg.beliefs.size() * sizeof(double));
    return PORRT_OK;
}
int porrt_bg_get_observable_zones(const porrt_ctx *c, uint64_t *masks) {
    if (!c || !c->bg.valid || !masks) return PORRT_ERR_INVALID;
    memcpy(masks, c->bg.h_vis.data(), c->bg.h_vis.size() * sizeof(uint64_t));
    return PORRT_OK;
}
int porrt_bg_get_node_types(const porrt_ctx *cc, uint8_t *types) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c || !c->bg.valid || !types) return PORRT_ERR_INVALID;
    HIPCHK_CTX(c, hipSetDevice(c->device));
    HIPCHK_CTX(c, hipMemcpy(types, c->bg.d_types, c->bg.N * c->bg.B, hipMemcpyDeviceToHost));
    return PORRT_OK;
}
static int bg_get_csr(porrt_ctx *c, const unsigned long long *d_off, const uint32_t *d_ids, uint64_t *off, uint32_t *ids) {
    if (!c || !c->bg.valid || !off) return PORRT_ERR_INVALID;
    HIPCHK_CTX(c, hipSetDevice(c->device));
    HIPCHK_CTX(c, hipMemcpy(off, d_off, (c->bg.N * c->bg.B + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (ids && c->bg.n_edges) HIPCHK_CTX(c, hipMemcpy(ids, d_ids, c->bg.n_edges * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return PORRT_OK;
}
int porrt_bg_get_children(const porrt_ctx *cc, uint64_t *off, uint32_t *ids) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    return c ? bg_get_csr(c, c->bg.d_child_off, c->bg.d_child_id, off, ids) : PORRT_ERR_INVALID;
}
int porrt_bg_get_parents(const porrt_ctx *cc, uint64_t *off, uint32_t *ids) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    return c ? bg_get_csr(c, c->bg.d_par_off, c->bg.d_par_id, off, ids) : PORRT_ERR_INVALID;
}
int porrt_bg_get_seconds(const porrt_ctx *c, double *out, uint32_t n) {
    if (!c || !c->bg.valid || !out) return PORRT_ERR_INVALID;
    const double v[8] = {c->bg.t_total, c->bg.t_device, c->bg.t_tables, c->bg.t_reach, c->bg.t_post, c->bg.t_adj, c->bg.t_alloc, c->bg.t_edges};
    for (uint32_t k = 0; k < n && k < 8; ++k) out[k] = v[k];
    return PORRT_OK;
}

// ---- expected costs over the belief graph (pto.rs:261-275, belief_graph.rs:89-175)
int porrt_bg_compute_expected_costs(porrt_ctx *c) { return c ? c->compute_expected_costs() : PORRT_ERR_INVALID; }
int porrt_bg_get_expected_costs(const porrt_ctx *cc, double *out) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c || !c->dp.valid || !c->bg.valid || !out) return PORRT_ERR_INVALID;
    HIPCHK_CTX(c, hipSetDevice(c->device));
    HIPCHK_CTX(c, hipMemcpy(out, c->dp.d_dist, c->dp.n * sizeof(double), hipMemcpyDeviceToHost));
    return PORRT_OK;
}
int porrt_bg_expected_cost_of(const porrt_ctx *cc, uint64_t belief_node, double *out) {
    porrt_ctx *c = const_cast<porrt_ctx *>(cc);
    if (!c || !c->dp.valid || !out || belief_node >= c->dp.n) return PORRT_ERR_INVALID;
    HIPCHK_CTX(c, hipSetDevice(c->device));
    HIPCHK_CTX(c, hipMemcpy(out, c->dp.d_dist + belief_node, sizeof(double), hipMemcpyDeviceToHost));
    return PORRT_OK;
}
int porrt_bg_get_dp_info(const porrt_ctx *c, double *total_s, double *device_s, uint32_t *sweeps) {
    if (!c || !c->dp.valid) return PORRT_ERR_INVALID;
    if (total_s) *total_s = c->dp.t_total;
    if (device_s) *device_s = c->dp.t_device;
    if (sweeps) *sweeps = c->dp.sweeps;
    return PORRT_OK;
}
uint64_t porrt_bg_get_dp_sweep_rows(const porrt_ctx *c) { return c && c->dp.valid ? c->dp.sweep_rows : 0; }

// PTO::extract_policy: returns the number of policy nodes (or a negative error); fills the arrays when they hold that many
int64_t porrt_bg_extract_policy(porrt_ctx *c, uint64_t *original_ids, int64_t *parents, uint8_t *is_leaf, uint64_t cap, double *expected_costs) {
    if (!c) return PORRT_ERR_INVALID;
    if (!c->dp.have_policy) {
        int r = c->extract_policy();
        if (r) return r;
    }
    const uint64_t n = c->dp.pol_original.size();
    if (expected_costs && hipMemcpy(expected_costs, c->dp.d_dist, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return PORRT_ERR_DEVICE;
    if (cap >= n) {
        for (uint64_t k = 0; k < n; ++k) {
            if (original_ids) original_ids[k] = c->dp.pol_original[k];
            if (parents) parents[k] = c->dp.pol_parent[k];
            if (is_leaf) is_leaf[k] = c->dp.pol_leaf[k];
        }
    }
    return (int64_t)n;
}

// conditional_dijkstra on an explicit graph (host arrays in, dist out): the form the reference's own tests call it in
int porrt_conditional_dijkstra(int device, uint64_t n, const double *xy, const uint32_t *belief_row, const double *beliefs, uint32_t n_belief_rows,
                               uint32_t n_worlds, const uint8_t *types, const uint64_t *child_off, const uint32_t *child_ids,
                               const uint64_t *parent_off, const uint32_t *parent_ids, const uint64_t *finals, uint64_t n_final, double *dist) {
    if (!n || !xy || !belief_row || !beliefs || !types || !child_off || !parent_off || !dist || (n_final && !finals) || n >= 0xFFFFFFFFull)
        return PORRT_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return PORRT_ERR_DEVICE;
    for (uint64_t i = 0; i < n; ++i) if (belief_row[i] >= n_belief_rows) return PORRT_ERR_INVALID;
    for (uint64_t k = 0; k < n_final; ++k) if (finals[k] >= n) return PORRT_ERR_INVALID;
    const uint64_t nc = child_off[n], np = parent_off[n];
    for (uint64_t k = 0; k < nc; ++k) if (child_ids[k] >= n) return PORRT_ERR_INVALID;
    for (uint64_t k = 0; k < np; ++k) if (parent_ids[k] >= n) return PORRT_ERR_INVALID;
    std::vector<double> hx(n), hy(n);
    for (uint64_t i = 0; i < n; ++i) { hx[i] = xy[2 * i]; hy[i] = xy[2 * i + 1]; }
    std::vector<void *> owned;
    auto up = [&](const void *src, size_t bytes) -> void * {
        void *d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(bytes, 8)) != hipSuccess) return nullptr;
        owned.push_back(d);
        if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    };
    DpConst c{};
    c.n = n; c.B = 1; c.nw = n_worlds;
    c.nx = (const double *)up(hx.data(), n * 8); c.ny = (const double *)up(hy.data(), n * 8);
    c.bvec = (const uint32_t *)up(belief_row, n * 4);
    c.beliefs = (const double *)up(beliefs, (size_t)n_belief_rows * n_worlds * 8);
    c.types = (const uint8_t *)up(types, n);
    c.child_off = (const unsigned long long *)up(child_off, (n + 1) * 8); c.par_off = (const unsigned long long *)up(parent_off, (n + 1) * 8);
    c.child_id = (const uint32_t *)up(child_ids, nc * 4); c.par_id = (const uint32_t *)up(parent_ids, np * 4);
    int r = PORRT_ERR_DEVICE;
    if (c.nx && c.ny && c.bvec && c.beliefs && c.types && c.child_off && c.par_off && c.child_id && c.par_id) {
        DpState st;
        std::string err;
        std::vector<unsigned long long> f(finals, finals + n_final);
        r = dp_run(st, c, false, f, nullptr, err);
        if (r == PORRT_OK && hipMemcpy(dist, st.d_dist, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) r = PORRT_ERR_DEVICE;
    }
    for (void *d : owned) (void)hipFree(d);
    return r;
}

int porrt_get_metrics(const porrt_ctx *c, porrt_metrics *out) {
    if (!c || !out) return PORRT_ERR_INVALID;
    *out = c->metrics;
    return PORRT_OK;
}

// Runs sqrt and divide on n pseudo-random doubles on the device and counts results that differ from the
// host's correctly rounded ones.  Bit-exact parity with the reference needs both counts to be zero.
int porrt_selftest(porrt_ctx *c, uint64_t n, uint64_t *sqrt_mismatch, uint64_t *div_mismatch) {
    if (!c || !n) return PORRT_ERR_INVALID;
    std::vector<double> a(n), b(n), rs(n), rd(n);
    Pcg64 r;
    r.seed_from_u64(12345);
    for (uint64_t i = 0; i < n; ++i) {
        // mix of magnitudes: squared distances (1e-12..8), pixel-scale values, general doubles
        double u = r.gen_range_f64(0.0, 1.0), v = r.gen_range_f64(0.0, 1.0);
        int e = (int)(r.next_u64() % 80) - 40;
        a[i] = ldexp(u, (i % 3 == 0) ? e : ((i % 3 == 1) ? -10 : 2));
        b[i] = ldexp(v + 1e-3, (i % 5 == 0) ? -e : 0);
    }
    double *da = nullptr, *db = nullptr, *ds = nullptr, *dd = nullptr;
    if (hipSetDevice(c->device) != hipSuccess) return PORRT_ERR_DEVICE;
    if (hipMalloc((void **)&da, n * 8) != hipSuccess || hipMalloc((void **)&db, n * 8) != hipSuccess ||
        hipMalloc((void **)&ds, n * 8) != hipSuccess || hipMalloc((void **)&dd, n * 8) != hipSuccess) {
        c->set_err("selftest: hipMalloc");
        return PORRT_ERR_DEVICE;
    }
    (void)hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_selftest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, da, db, ds, dd, (unsigned long long)n);
    (void)hipStreamSynchronize(c->stream);
    hipError_t e1 = hipMemcpy(rs.data(), ds, n * 8, hipMemcpyDeviceToHost);
    hipError_t e2 = hipMemcpy(rd.data(), dd, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(ds); (void)hipFree(dd);
    if (e1 != hipSuccess || e2 != hipSuccess) { c->set_err("selftest: copy back"); return PORRT_ERR_DEVICE; }
    uint64_t ms = 0, md = 0;
    for (uint64_t i = 0; i < n; ++i) {
        volatile double hs = sqrt(a[i]), hd = a[i] / b[i];
        double xs = hs, xd = hd;
        if (memcmp(&xs, &rs[i], 8)) ++ms;
        if (memcmp(&xd, &rd[i], 8)) ++md;
    }
    if (sqrt_mismatch) *sqrt_mismatch = ms;
    if (div_mismatch) *div_mismatch = md;
    return PORRT_OK;
}

porrt_tree_device_view porrt_tree_device(const porrt_ctx *c) {
    porrt_tree_device_view v;
    memset(&v, 0, sizeof v);
    if (!c || !c->have_results || c->mode != PORRT_MODE_RRT) return v;
    v.nx = c->d_nx.p; v.ny = c->d_ny.p; v.dist_root = c->d_distA.p; v.parent = c->d_parent.p; v.n_nodes = c->n_nodes;
    return v;
}

// read-only views of what the options chose: "launch_mode" (last porrt_grow_batch led by this context: 0 = one launch sequence,
// G = G sequences side by side on streams chosen by measurement, -G = G sequences on the contexts' own streams because the probe
// found no set of parallel streams), "pipeline", "group_lanes" (in force for the last grow)
int porrt_get_option(const porrt_ctx *c, const char *name, int64_t *value) {
    if (!c || !name || !value) return PORRT_ERR_INVALID;
    if (!strcmp(name, "launch_mode")) *value = c->last_launch_mode;
    else if (!strcmp(name, "pipeline")) *value = c->lag_on ? 4 : (c->pipe_on ? 1 : 0);
    else if (!strcmp(name, "group_lanes")) *value = c->opt_group;
    else if (!strcmp(name, "kd_lazy")) *value = c->kd_lazy ? 1 : 0;                 // in force for the last grow (of a batch: ask its first context)
    else if (!strcmp(name, "kd_lca_steps")) *value = (int64_t)c->counters.lca_next;      // this context's own need: 1 + the last step with a tie that took the structure
    else if (!strcmp(name, "compactions")) *value = c->n_compactions;               // how often the last batch this context led gathered its running rows
    else if (!strcmp(name, "kd_built_after")) *value = c->kd_built_after;           // 1: a tie of the last grow needed the whole kd structure, built after the steps
    else return PORRT_ERR_INVALID;
    return PORRT_OK;
}

int porrt_set_option(porrt_ctx *c, const char *name, int64_t value) {
    if (!c || !name) return PORRT_ERR_INVALID;
    if (!strcmp(name, "profile")) c->opt_profile = value != 0;
    else if (!strcmp(name, "cand_cap")) c->opt_cand_cap = (uint32_t)std::max<int64_t>(64, std::min<int64_t>(value, 1 << 26));
    else if (!strcmp(name, "graph")) c->opt_graph = value != 0;
    else if (!strcmp(name, "group_lanes")) { if (value != -1 && value != 0 && value != 16 && value != 32 && value != 64) { c->set_err("group_lanes: -1 (auto), 0, 16, 32 or 64"); return PORRT_ERR_INVALID; } c->opt_group_req = (int)value; }
    else if (!strcmp(name, "dp_sweeps")) c->opt_dp_sweeps = value != 0;
    else if (!strcmp(name, "box_table")) { c->opt_box_table = value != 0; c->cls_dirty = true; }
    else if (!strcmp(name, "gtrack_side")) c->opt_gtrack_side = value != 0;
    else if (!strcmp(name, "pipeline")) {
#ifndef PORRT_DEV_NONCOOP
        // 3 = the persistent step loop launched WITHOUT the co-residency guarantee of a cooperative launch (its barriers then rest on
        // the two-second give-up alone): a developer build only (-DPORRT_DEV_NONCOOP), never through the shipped option parser
        if (value == 3) { c->set_err("pipeline = 3 (non-cooperative persistent launch) is a developer build option"); return PORRT_ERR_INVALID; }
#endif
        c->opt_pipeline = (value >= 2 && value <= 4) ? (int)value : (value ? 1 : 0);
    }
    else if (!strcmp(name, "batch_streams")) c->opt_batch_streams = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 8));
    else if (!strcmp(name, "kd_after")) c->opt_kd_after = value != 0;
    else if (!strcmp(name, "kd_ride")) c->opt_kd_ride = value != 0;
    else if (!strcmp(name, "kd_lazy")) c->opt_kd_lazy = value == 2 ? 2 : (value != 0);
    else if (!strcmp(name, "kd_claim_threads")) { if (value != 0 && value != 256 && value != 512 && value != 1024) return PORRT_ERR_INVALID; c->opt_claim_threads = (uint32_t)value; }
    else if (!strcmp(name, "kd_inline")) c->opt_kd_inline = value != 0;
    else if (!strcmp(name, "compact_rows")) c->opt_compact = value != 0;
    else if (!strcmp(name, "fetch_workers")) c->opt_fetch_workers = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(value, 32));
    else if (!strcmp(name, "tree_out_blocks")) c->opt_tree_out_blocks = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(value, 1024));
    else if (!strcmp(name, "host_ranks")) { c->opt_host_ranks = value != 0; c->eo.tag = ~0ull; }
    else if (!strcmp(name, "early_wave_steps")) c->opt_early_wave = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 64));
    else if (!strcmp(name, "kd_group")) c->opt_kd_group = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(value, 8));
    else { c->set_err(std::string("unknown option ") + name); return PORRT_ERR_INVALID; }
    // a captured launch sequence has the options of its capture in it: the next grow captures again
    if (strcmp(name, "profile") && c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    return PORRT_OK;
}

} // extern "C"

#include "porrt_exchange.hpp"
#include "porrt_formats.hpp"
