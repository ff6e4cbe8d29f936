// porrt_device.hpp -- gfx950 kernels of the batched RRT*/PTO expansion engine.
//
// One grow step ("batch") of K samples runs three kernels on the main stream:
//   k_near           one wave per sample: nearest neighbour, steer, point validity, radius search -- exact f64
//                    brute force over the region pages the query disc touches (replaces
//                    KdTree::nearest_neighbor[_filtered] / nearest_neighbors, src/nearest_neighbor.rs:48-126;
//                    src/common.rs:215-225, map_shelves_io.rs:158-170, map_io.rs:165-181)
//   k_connect_rrt / k_connect_pto   Bresenham raycasts on the LDS-resident grid, best parent (wave argmin),
//                    new node, rewire phase 1 / reachability        (src/rrt.rs:123-161, src/pto.rs:95-124,
//                                                                  map_shelves_io.rs:187-203, map_io.rs:216-241,
//                                                                  pto_reachability.rs:42-52)
//                    + one extra workgroup that files the step's new nodes into the region pages
//   k_commit_rrt / k_commit_pto     rewire phase 2 (deterministic winner) / reach sync, step counter, NN bounds
//                    of the next step's samples
// plus k_gen_samples (Pcg64 + gen_range on the device, sample_space.rs:30-36, rrt.rs:176-181) once per grow,
// and on a side stream (RRT*) k_kd_locate / k_kd_claim: the STRUCTURE of the reference's kd-tree, kept only to
// resolve equal-cost parents in its pre-order.
//
// Arithmetic is IEEE f64 without contraction (-ffp-contract=off): it has to reproduce the reference's rounding
// exactly (integer parents bit-exact, coordinates bit-exact).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace porrt {

// timing events of one call: destroyed on every way out of it (an early error return used to leak them)
template <int NEV>
struct ScopedEvents {
    hipEvent_t e[NEV];
    ScopedEvents() { for (int i = 0; i < NEV; ++i) e[i] = nullptr; }
    hipError_t create() {
        for (int i = 0; i < NEV; ++i) { const hipError_t r = hipEventCreate(&e[i]); if (r != hipSuccess) return r; }
        return hipSuccess;
    }
    ~ScopedEvents() { for (int i = 0; i < NEV; ++i) if (e[i]) (void)hipEventDestroy(e[i]); }
    ScopedEvents(const ScopedEvents &) = delete;
    ScopedEvents &operator=(const ScopedEvents &) = delete;
};

constexpr int kConnectWaves = 4;     // samples per connect workgroup (one wave each)
constexpr uint32_t kTileRMax = 31;             // LDS tile half-width limit (pixels); above it rays read global
constexpr int kEmpty = 0x7FFFFFFF;       // empty kd child slot (atomicMin claims it)
constexpr uint32_t kOnG = 0x80000000u;
constexpr int kParentPending = -2;       // parent of a new node whose equal-cost parents await the kd structure (k_tie_fix)

enum : uint32_t {
    ERR_RASTER = 1u,        // pixel outside the raster / door pixel without zone / two zones on one segment
    ERR_CAND_OVERFLOW = 2u, // neighbour list capacity
    ERR_RNG_RETRY = 4u,     // gen_range would have redrawn (host regenerates the stream exactly)
    ERR_EDGE_OVERFLOW = 8u,
    ERR_GPATH_OVERFLOW = 16u,
    ERR_PAGE_OVERFLOW = 32u,
    ERR_TIE_POOL = 64u       // deferred-tie records
};

// pixel classes of the pre-classified raster (host builds it in set_grid/set_zones)
enum : uint8_t { CLS_FREE = 0, CLS_LOW = 1, CLS_HIGH = 2, CLS_HIGH0 = 3 /* raw pixel 0 */, CLS_ZONE = 16, CLS_BAD = 255 };

struct __attribute__((aligned(8))) KdRec {
    double x, y;
    int child[2];
};

struct __attribute__((aligned(32))) KdBox {
    double lox, hix, loy, hiy;      // [lo, hi) per axis (nearest_neighbor.rs:32: `<` goes left, equal goes right)
};

struct KdMove {                 // a node on its way down
    uint32_t t;                 // index among the launch's new nodes (id = N + t)
    int cur;                    // parent: node id, or index of a new node once below one
    uint32_t side, dcur, gex;
    bool onpath;
    double vx, vy;
    KdBox box;
};

struct Counters {
    uint32_t n_final;
    uint32_t n_edges;
    uint32_t err;
    uint32_t tie_fallbacks;
    unsigned long long finality;
    uint32_t g_len;
    uint32_t n_heavy;
    uint32_t g_nd_len;          // non-duplicate levels of G
    uint32_t g_first_dup[2];    // first level of even / odd depth held by a duplicate of the goal point
    uint32_t n_pages;           // pool pages handed out (region pages)
    uint32_t kd_done;           // nodes with complete kd records (release-stored by k_kd_claim)
    uint32_t pend_cnt, pool_n;  // deferred ties: records, pooled candidate ids
    uint32_t n_deferred, pend_lo;
    uint32_t n_losers;          // k_kd_link -> k_kd_claim
    uint32_t kd_snap;           // step up to whose start the kd structure is complete (release-stored by k_kd_claim)
    uint32_t dbg[4];            // developer statistics (kd claim: max / sum of losers, launches; non-duplicate levels of G)
    uint32_t clone_n;           // valid samples of the running step steered exactly onto the goal point (k_nn2 -> k_conn2)
    uint32_t clone_k[64];
    uint32_t coop_bar;          // persistent step loop (k_coop_rrt): arrivals at its grid barrier (monotone)
    uint32_t coop_filed;        //   steps whose nodes have their positions, ids and page slots (what the kd kernels beside it wait for)
    uint32_t coop_abort;        //   a barrier or a waiting kernel gave up (nothing should ever set it: a guard against a hung GPU)
    uint32_t sched_stop;        // porrt_grow_batch with a loop condition: the first step this row did not run (0xFFFFFFFF: still running)
    uint32_t sched_iter;        //   and the iterations it ran (rrt.rs:109 / pto.rs:67: i when the loop ended)
    uint32_t n_lca;             // kd_lazy: ties between nodes at different places off the goal path (the kd structure is built after the steps then)
    uint32_t lca_next;          // kd_lazy: 1 + the last step with such a tie -- the structure is built for the steps before it only
    unsigned long long tim[16]; // developer builds (-DPORRT_TIMING): phase durations summed over waves, 10 ns units, and wave counts
};

// phase timers of developer builds: T0 at the start of a phase, TACC(slot) adds the time since (one lane per wave)
#ifdef PORRT_TIMING
#define PORRT_T0() unsigned long long t__0 = wall_clock64()
#define PORRT_TACC(rc, slot)                                                                          \
    do {                                                                                              \
        const unsigned long long t__1 = wall_clock64();                                               \
        if ((threadIdx.x & 63u) == 0u) { atomicAdd(&(rc).cnt->tim[slot], t__1 - t__0); atomicAdd(&(rc).cnt->tim[(slot) + 8], 1ull); } \
        t__0 = t__1;                                                                                  \
    } while (0)
#else
#define PORRT_T0() do {} while (0)
#define PORRT_TACC(rc, slot) do {} while (0)
#endif
// -DPORRT_TIMING=1: the timers of k_nn2, =2: those of k_conn2 (they share the slots)
#ifndef PORRT_TIMING_FROM
#define PORRT_TIMING_FROM 0
#endif
#ifndef PORRT_TIMING_UNTIL
#define PORRT_TIMING_UNTIL 0x7FFFFFFF
#endif
#if defined(PORRT_TIMING) && PORRT_TIMING == 1
#define PORRT_TACC_A(rc, slot) do { if (b >= (uint32_t)PORRT_TIMING_FROM && b < (uint32_t)PORRT_TIMING_UNTIL) PORRT_TACC(rc, slot); else t__0 = wall_clock64(); } while (0)
#define PORRT_TACC_B(rc, slot) do { (void)t__0; } while (0)
#elif defined(PORRT_TIMING)
#define PORRT_TACC_A(rc, slot) do { (void)t__0; } while (0)
// (-DPORRT_TIMING_FROM=<step> [-DPORRT_TIMING_UNTIL=<step>]: only those steps are accumulated -- the steady state without the dense first
// steps, or the dense first steps alone)
#define PORRT_TACC_B(rc, slot) do { if (b >= (uint32_t)PORRT_TIMING_FROM && b < (uint32_t)PORRT_TIMING_UNTIL) PORRT_TACC(rc, slot); else t__0 = wall_clock64(); } while (0)
#else
#define PORRT_TACC_A(rc, slot) do {} while (0)
#define PORRT_TACC_B(rc, slot) do {} while (0)
#endif
// =3: the timers of k_kd_locate
#if defined(PORRT_TIMING) && PORRT_TIMING == 3
#undef PORRT_TACC_B
#define PORRT_TACC_B(rc, slot) do { (void)t__0; } while (0)
#define PORRT_TACC_C(rc, slot) PORRT_TACC(rc, slot)
#elif defined(PORRT_TIMING)
#define PORRT_TACC_C(rc, slot) do { (void)t__0; } while (0)
#else
#define PORRT_TACC_C(rc, slot) do {} while (0)
#endif

// =4: where a heavy sample's time goes (more hits than its LDS list holds: served by the whole wave, heavy_sample_wave) -- slot 0 its
// hits read back, 1 the rays and costs of the hits, 4-7 as for =2 but for heavy samples only, 3 the whole of it
#if defined(PORRT_TIMING) && PORRT_TIMING == 4
#undef PORRT_TACC_B
#define PORRT_TACC_B(rc, slot) do { (void)t__0; } while (0)
#define PORRT_TACC_H(rc, slot) do { if (b >= (uint32_t)PORRT_TIMING_FROM && b < (uint32_t)PORRT_TIMING_UNTIL) PORRT_TACC(rc, slot); else t__0 = wall_clock64(); } while (0)
#define PORRT_TACC_S(rc, slot) do { if (TeamT::kSize == 64 && b >= (uint32_t)PORRT_TIMING_FROM && b < (uint32_t)PORRT_TIMING_UNTIL) PORRT_TACC(rc, slot); else t__0 = wall_clock64(); } while (0)
#elif defined(PORRT_TIMING)
#define PORRT_TACC_H(rc, slot) do { (void)t__0; } while (0)
#define PORRT_TACC_S(rc, slot) PORRT_TACC_B(rc, slot)
#else
#define PORRT_TACC_H(rc, slot) do {} while (0)
#define PORRT_TACC_S(rc, slot) do {} while (0)
#endif

struct BestCost {
    unsigned long long cost_bits;   // f64 bits of the best cost (+inf bits when there is no final node)
    uint32_t final_id, path_len, overflow, pad;
};

typedef float flt2 __attribute__((ext_vector_type(2)));
struct RunConst {
    // node SoA
    double *nx, *ny;
    int *rep;                   // cell -> some node in that cell (3-level pyramid), only ever used for bounds
    flt2 *rep_f;              // cell -> that node's position rounded to f32 (levels 0 and 1; NaN = empty): k_nn2's unsteered-sample test
    double bx0, by0, binv_w, binv_h;   // box the pyramid covers
    // region pages (see scan_disc)
    uint32_t *rg_cnt;           // [2][kRegions] nodes per region, by step parity: step b searches [b & 1] while its new nodes are filed into [(b + 1) & 1]
    unsigned long long *rg_occ; // [2][kOccWords] bit r = region r holds a node (same parity as rg_cnt): the nearest-neighbour search of
                                // k_nn2 finds the occupied regions of a thin tree without walking the empty ones
    uint32_t *rg_dir;           // [region][j]: j-th page of the region, j >= 1
    double *pg_xy;              // [page][slot] (x, y)
    int *pg_id;                 // [page][slot] node id
    double *pg_d;               // [page][slot] dist_root as of the step's start (RRT*): the radius search gets it with the coordinates
    uint32_t *slot_of;          // node -> page * 64 + slot (the rewire commit keeps pg_d in step with distA)
    uint32_t rg_maxp, pg_cap;   // directory stride, pages in the pool
    double *distA, *distB;      // dist_root: A = snapshot read by the step, B = rewire accumulator
    int *parent;
    unsigned long long *reachA, *reachB;
    uint8_t *vid;               // PTO validity id per node
    uint8_t *final_flag;
    unsigned long long *final_mask;
    uint32_t *n_at;             // n_at[b] = tree size at the start of step b
    unsigned long long *valid_mask; // per step: bit k = sample k produced a node
    Counters *cnt;
    // sample stream for the run (one entry per iteration)
    double *sx, *sy;
    uint32_t *sworld;
    const double *inj_xy;       // injected ContinuousSampler values or null
    unsigned long long inj_base, inj_n;
    // per-sample step scratch
    double *q_x, *q_y;          // steered state
    double *kq_x, *kq_y;        // copy for the kd insertion ([step][sample])
    int *kq_vid;
    int *q_nn;
    int *q_vid;
    uint32_t *cand_cnt;         // neighbour lists of the step: counts and ids are double-buffered by step parity
    int *cand_id;               // (k_kd_locate reads them beside the next step's search), written by k_near only
    double *cand_val;
    double *cand_xy;            // coordinates of the listed neighbours (same layout and parity as cand_id, 16 B each): the
                                // search has them in registers, connect would otherwise gather two more cache lines per neighbour
    uint32_t cand_cap;
    uint32_t cand_K;            // samples per parity buffer
    // radius tables: T2[n] = largest d^2 whose sqrt rounds to <= heuristic_radius(n)
    const double *rad_T2;
    // edges (PTO)
    uint32_t *e_from, *e_to, *e_tv;
    uint32_t e_cap;
    // kd-tree structure of the reference (RRT* tie order, see k_kd_insert): child ids (kEmpty = none),
    // parent id, depth, and where the node's root path leaves the goal path G (bit 31: the node is ON G)
    KdRec *kd_rec;              // packed {x, y, child[2]}: one load per level of a descent
    KdBox *kd_box;              // the cell a node was inserted into: a point's root path passes the node iff it lies inside
    KdBox *loc_box;             // per new node of the step: cell of the empty slot k_kd_locate stopped at
    uint32_t loc_stride;        // loc_* hold two groups (slot lpar of the kd kernels; one side stream uses slot 0 only)
    KdMove *kd_losers;          // nodes that lost the bid for their slot ([kClaimMax])
    // deferred equal-cost parents (see connect_rrt_sample / k_tie_fix)
    int *pend_new, *pend_pool;
    uint32_t *pend_off, *pend_n, *pend_cur, *pend_state;
    uint32_t pend_cap, pool_cap;
    // k_best_cost scratch (paths of the final nodes), its cursor and result slot
    int *bc_scratch;
    uint32_t bc_cap;
    uint32_t *bc_cursor;
    BestCost *bc_out;
    unsigned long long *kd_hint; // [kHG * kHG] (depth << 32 | id) of the deepest node whose cell covers the square
    int *kd_up;
    uint32_t *kd_depth;
    uint32_t *kd_gexit;
    int *loc_cur;               // per new node of the step: where k_kd_locate stopped (node, depth, G exit, flags)
    uint32_t *loc_dcur, *loc_gex, *loc_flags, *g_nd;
    double *g_nd_x, *g_nd_y;    // coordinates of the non-duplicate levels of G (contiguous, no indirection)
    KdBox *g_nd_box;            // the goal point's cell after each of those levels (lazily tracked runs: g_track_step) -- nested, so the level a node leaves G at is a binary search
    int *g_id;                  // G[i] = node at depth i on the kd descent path of the goal point
    double *g_x, *g_y;          // its coordinates, contiguous (the path is scanned, not chased)
    uint32_t g_cap;
    uint32_t *g_snap;           // [step][4]: g_len, g_nd_len, g_first_dup[0..1] at the start of the step's kd insertion
    double gp_x, gp_y;
    // grid
    const uint8_t *cls;
    const uint32_t *sat;        // (H + 1) x (W + 1) summed-area table of the pixels that are not CLS_FREE: sat[i][j] = their number in rows < i, columns < j
    const uint8_t *clr;         // per pixel: Chebyshev distance to the nearest pixel that is not CLS_FREE or lies outside (0 on such a
                                // pixel, capped at 255): a segment whose end pixels are closer than that crosses free pixels only
    uint32_t W, H;
    double low0, low1, ppm;
    int domain, has_grid;
    int n_validities;
    unsigned long long validities[65];
    unsigned long long all_worlds;
    // goal
    int goal_kind;              // 0 none, 1 square, 2 observation
    uint32_t G;
    double gcx[64], gcy[64];
    unsigned long long gmask[64];
    double g_l1;
    double w2g_x[64], w2g_y[64];
    double zone_x, zone_y, visibility;
    // sampler box + params
    double s_low0, s_low1, s_up0, s_up1;
    double max_step;
    int mode;
    uint16_t *perm;             // [step][part_stride]: the step's sample indices ordered by where the samples lie (k_sort_samples)
    double *ssx, *ssy;          // [step][part_stride]: the samples in that order (one coalesced load instead of perm -> sx, sy)
    double *bq_x, *bq_y;        // [part_stride] steered states in slot order and the sample they belong to (0xFFFF: nothing for
    uint16_t *bq_k;             //   the connect groups to do), written by k_nn2 for k_conn2
    double *t2_at;              // t2_at[b] = rad_T2[n_at[b]] (RRT*: the step's radius threshold, written with n_at)
    uint32_t tile_R;            // LDS tile half-width in pixels (0 = no tile: read the raster from global)
    uint32_t part_stride;       // sample stride of the arrays double-buffered by step parity
    uint32_t q_stride;          // q_x / q_y / q_nn / q_vid of step b start at (b & 1) * q_stride: part_stride when the steps are
                                // pipelined (k_step_rrt: step b + 1 is searched while step b is connected), else 0
    // what the preparation of a grow needs, for the kernels that prepare all members of a porrt_grow_batch at once (k_batch_prep,
    // and k_init_root / k_gen_samples / k_sort_samples with one grid row per member)
    double start_x, start_y;
    unsigned long long root_reach;
    int root_vid;
    unsigned long long rng_st_lo, rng_st_hi, rng_inc_lo, rng_inc_hi;      // the continuous sampler at the start of the grow
    struct PcgJump *jump;       // its jump table
    RunConst *self;             // the context's own copy on the device (kernels launched for it alone read that one)
    uint32_t *zero0;            // [zero0, zero0 + zero_words): counters, region counts, valid masks, kd hints, deferred-tie states
    unsigned long long zero_words;
    // The row's own step schedule (porrt_grow_batch whose members have their own n_iter_min / n_iter_max, or a loop condition that
    // can end a member early): step b takes iterations [sched_i0[b], sched_i0[b] + sched_nb[b]).  The plan is static -- steps of K
    // up to n_iter_min, then steps of K up to n_iter_max -- and k_row_sched zeroes sched_nb[b] from the step on
    // at which the loop condition ends the row.  Null: every row runs the launch's (i0, nb).
    uint32_t *sched_i0, *sched_nb;
    uint32_t sched_min, sched_max, sched_K, sched_steps;
    uint32_t sched_max_nodes;           // nodes this grow may create (n_iter_max + 2): what its preparation clears
    uint32_t cand_par3;                 // 1: neighbour lists, counts and values in three buffers by step % 3 (cand_slot)
    uint32_t hint_max;                  // kd_hint_block: cells covering more hint squares leave the hints alone (a batch of many rows)
    uint32_t kd_lazy;                   // 1: only the goal path of the kd-tree is kept beside the steps (g_track_step); see there
};

// Pointers read out of RunConst have no known address space, so hipcc emits flat_* accesses and drains both
// memory counters around each of them.  Everything lives in hipMalloc'ed HBM: viewing a pointer as global
// (address space 1) gives global_* instructions with exact vmcnt bookkeeping.
#define GPTR(T) T __attribute__((address_space(1))) *
template <class T>
__device__ __forceinline__ GPTR(T) as_global(T *p) { return (GPTR(T))(uintptr_t)p; }
template <class T>
__device__ __forceinline__ GPTR(const T) as_global(const T *p) { return (GPTR(const T))(uintptr_t)p; }
template <class T>
__device__ __forceinline__ T g_atomic_add(GPTR(T) p, T v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T g_atomic_or(GPTR(T) p, T v) { return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T g_atomic_min(GPTR(T) p, T v) { return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ------------------------------------------------------------------ small helpers
// what a kernel launched for step b with (i0, nb) does for THIS row
__device__ __forceinline__ uint32_t row_nb(const RunConst &rc, uint32_t b, uint32_t nb_arg) {
    return rc.sched_nb ? ((GPTR(const uint32_t))(uintptr_t)rc.sched_nb)[b] : nb_arg;
}
__device__ __forceinline__ uint32_t row_i0(const RunConst &rc, uint32_t b, uint32_t i0_arg) {
    return rc.sched_i0 ? ((GPTR(const uint32_t))(uintptr_t)rc.sched_i0)[b] : i0_arg;
}


// neighbour list of sample k in step b (parity buffers)
__device__ __forceinline__ uint32_t q_off(const RunConst &rc, uint32_t b) { return (b & 1u) * rc.q_stride; }
// (two buffers by step parity; three, and the per-sample values too, when a step's lists are still read by the rewire commit while
// the search two steps on writes: the one-kernel-per-step form of the single query, RunConst::cand_par3)
__device__ __forceinline__ uint32_t cand_slot(const RunConst &rc, uint32_t b) { return rc.cand_par3 ? b % 3u : (b & 1u); }
__device__ __forceinline__ size_t cand_off(const RunConst &rc, uint32_t b, uint32_t k) { return ((size_t)cand_slot(rc, b) * rc.cand_K + k) * rc.cand_cap; }
__device__ __forceinline__ uint32_t cand_cnt_at(const RunConst &rc, uint32_t b, uint32_t k) { return cand_slot(rc, b) * rc.cand_K + k; }
__device__ __forceinline__ size_t cand_val_off(const RunConst &rc, uint32_t b, uint32_t k) {
    return rc.cand_par3 ? ((size_t)cand_slot(rc, b) * rc.cand_K + k) * rc.cand_cap : (size_t)k * rc.cand_cap;
}
__device__ __forceinline__ uint32_t cand_count(const RunConst &rc, uint32_t b, uint32_t k) {
    const uint32_t c = as_global(rc.cand_cnt)[cand_cnt_at(rc, b, k)];
    return c < rc.cand_cap ? c : rc.cand_cap;
}
__device__ __forceinline__ unsigned long long f64_bits(double d) { return (unsigned long long)__double_as_longlong(d); }

// Rust `f64 as u32`: truncate toward zero, saturate, NaN -> 0
__device__ __forceinline__ uint32_t f64_as_u32(double v) {
    if (!(v == v)) return 0u;
    if (v <= 0.0) return 0u;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}

// map_shelves_io.rs:165-170 == map_io.rs:176-181
__device__ __forceinline__ void to_pixel(const RunConst &rc, double x, double y, uint32_t &i, uint32_t &j) {
    double t = (y - rc.low1) * rc.ppm;
    i = f64_as_u32((double)(rc.H - 1) - t);
    j = f64_as_u32((x - rc.low0) * rc.ppm);
}

// common.rs:203-213 with a = node, b = query
__device__ __forceinline__ double dist2(double ax, double ay, double bx, double by) {
    double dx = bx - ax, dy = by - ay;
    double xx = dx * dx, yy = dy * dy;
    return xx + yy;
}

// line_drawing 0.8 octant transforms (Bresenham<i32>, used at map_shelves_io.rs:196 / map_io.rs:225)
__device__ __forceinline__ void oct_to(int o, int x, int y, int &ox, int &oy) {
    switch (o) {
    case 0: ox = x; oy = y; break;
    case 1: ox = y; oy = x; break;
    case 2: ox = y; oy = -x; break;
    case 3: ox = -x; oy = y; break;
    case 4: ox = -x; oy = -y; break;
    case 5: ox = -y; oy = -x; break;
    case 6: ox = -y; oy = x; break;
    default: ox = x; oy = -y; break;
    }
}
__device__ __forceinline__ void oct_from(int o, int x, int y, int &ox, int &oy) {
    switch (o) {
    case 0: ox = x; oy = y; break;
    case 1: ox = y; oy = x; break;
    case 2: ox = -y; oy = x; break;
    case 3: ox = -x; oy = y; break;
    case 4: ox = -x; oy = -y; break;
    case 5: ox = -y; oy = -x; break;
    case 6: ox = y; oy = -x; break;
    default: ox = x; oy = -y; break;
    }
}

// Raster accessors for the raycasts.  GlobalGrid reads the pre-classified raster in HBM/L2; TileGrid reads
// the wave's private LDS tile (a (2R+1)^2 window around the new node: every neighbour lies within
// radius <= max_step of it, so all of a sample's rays stay inside) and falls back to global outside it.
struct GlobalGrid {
    static constexpr int kRayChunk = 6;     // pixels of a ray fetched together (traversed_class_px): a trip to memory per chunk
    const uint8_t *p;
    uint32_t W;
    __device__ __forceinline__ int at(uint32_t i, uint32_t j) const { return as_global(p)[i * W + j]; }
    static constexpr bool kAskTable = false;
};
// ... and where the kernel has the registers for it (the group kernels, the roadmap kernels): segment_box_free before a walk -- one trip
// instead of one per chunk
struct TableGrid : GlobalGrid {
    static constexpr bool kAskTable = true;
};
struct TileGrid {
    static constexpr int kRayChunk = 1;     // (the tile is in LDS: nothing to gain, and the one-wave-per-sample kernels have no registers to spare)
    const uint8_t *lds;     // TW x TW bytes
    const uint8_t *glob;
    uint32_t W;
    int oi, oj;             // raster coordinates of tile[0][0]
    uint32_t TW;
    __device__ __forceinline__ int at(uint32_t i, uint32_t j) const {
        const uint32_t ri = (uint32_t)((int)i - oi), rj = (uint32_t)((int)j - oj);
        if (ri < TW && rj < TW) return lds[ri * TW + rj];
        return as_global(glob)[i * W + j];
    }
    static constexpr bool kAskTable = false;     // (a walk through the LDS tile makes no trips, and the one-wave-per-sample kernels have no registers for the question)
};

// Is every pixel of the bounding box of two pixels free?  Four independent loads from the summed-area table.  (Pixels inside the raster.)
__device__ __forceinline__ bool segment_box_free(const RunConst &rc, uint32_t ai, uint32_t aj, uint32_t bi, uint32_t bj) {
    const uint32_t i0 = ai < bi ? ai : bi, i1 = (ai < bi ? bi : ai) + 1u, j0 = aj < bj ? aj : bj, j1 = (aj < bj ? bj : aj) + 1u, Ws = rc.W + 1u;
    auto S = as_global(rc.sat);
    return S[i1 * Ws + j1] - S[i0 * Ws + j1] - S[i1 * Ws + j0] + S[i0 * Ws + j0] == 0u;
}

// Traversed-space class of the segment a -> b (map_shelves_io.rs:187-203, map_io.rs:216-241), end points given as pixels
// (to_pixel).  Returns CLS_FREE / CLS_LOW / CLS_HIGH / CLS_ZONE+z.  Raster faults set *err and read as CLS_HIGH.
template <class Grid>
__device__ int traversed_class_px(const RunConst &rc, const Grid &grid, uint32_t ai, uint32_t aj, uint32_t bi, uint32_t bj, uint32_t *err) {
    if (ai >= rc.H || bi >= rc.H || aj >= rc.W || bj >= rc.W) {
        *err |= ERR_RASTER;
        return CLS_HIGH;
    }
    // The walk stays inside the bounding box of its end pixels: where that box holds free pixels only -- most segments of a tree, whose
    // nodes keep clear of nothing but are rarely around a corner from each other -- the answer takes one trip instead of a chain of chunks.
    if constexpr (Grid::kAskTable) {
        if (rc.sat != nullptr && segment_box_free(rc, ai, aj, bi, bj)) return CLS_FREE;       // (option "box_table" = 0: no table)
    }
    int x0 = (int)ai, y0 = (int)aj, x1 = (int)bi, y1 = (int)bj;
    int o = 0;
    {
        int dx = x1 - x0, dy = y1 - y0;
        if (dy < 0) { dx = -dx; dy = -dy; o += 4; }
        if (dx < 0) { int t = dx; dx = dy; dy = -t; o += 2; }
        if (dx < dy) o += 1;
    }
    int sx, sy, ex, ey;
    oct_to(o, x0, y0, sx, sy);
    oct_to(o, x1, y1, ex, ey);
    const int ddx = ex - sx, ddy = ey - sy;
    int e = ddy - ddx;
    int worst = CLS_FREE;
    // oct_from(o, x, y) is linear in (x, y): pi = fa x + fb y, pj = fc x + fd y (the same eight cases, as coefficients)
    const int fa = (o == 0 || o == 7) ? 1 : ((o == 3 || o == 4) ? -1 : 0);
    const int fb = (o == 1 || o == 6) ? 1 : ((o == 2 || o == 5) ? -1 : 0);
    const int fc = (o == 1 || o == 2) ? 1 : ((o == 5 || o == 6) ? -1 : 0);
    const int fd = (o == 0 || o == 3) ? 1 : ((o == 4 || o == 7) ? -1 : 0);
    // The walk does not depend on what it reads (only its early return does), and it stays inside the bounding box of its end
    // pixels: a chunk's pixels are addressed first and fetched together -- one trip to the raster per kRayChunk pixels instead of
    // one per pixel -- and then tested in the reference's order.
    constexpr int kRayChunk = Grid::kRayChunk;
    for (int x = sx, y = sy; x <= ex;) {
        int c[kRayChunk];
        const int n = ex - x + 1 < kRayChunk ? ex - x + 1 : kRayChunk;
#pragma unroll
        for (int u = 0; u < kRayChunk; ++u) {
            c[u] = CLS_FREE;
            if (u < n) {
                const int pi = fa * x + fb * y, pj = fc * x + fd * y;
                c[u] = grid.at((uint32_t)pi, (uint32_t)pj);
                if (e >= 0) { y += 1; e -= ddx; }
                e += ddy;
                ++x;
            }
        }
#pragma unroll
        for (int u = 0; u < kRayChunk; ++u) {
            if (u >= n) break;
            if (rc.domain == 0) {
                if (c[u] == CLS_HIGH0) return CLS_HIGH;       // lowest_pixel == 0: early return
                worst = c[u] > worst ? c[u] : worst;
            } else if (c[u] != CLS_FREE) {
                if (c[u] == CLS_HIGH || c[u] == CLS_HIGH0) return CLS_HIGH;   // Obstacle: immediate return
                if (c[u] == CLS_BAD) { *err |= ERR_RASTER; return CLS_HIGH; }
                if (worst >= CLS_ZONE && worst != c[u]) { *err |= ERR_RASTER; return CLS_HIGH; } // two zones: reference asserts
                worst = c[u];
            }
        }
    }
    return worst;
}
template <class Grid>
__device__ __forceinline__ int traversed_class(const RunConst &rc, const Grid &grid, double ax, double ay, double bx, double by, uint32_t *err) {
    uint32_t ai, aj, bi, bj;
    to_pixel(rc, ax, ay, ai, aj);
    to_pixel(rc, bx, by, bi, bj);
    return traversed_class_px(rc, grid, ai, aj, bi, bj, err);
}

// Is the segment between two pixels free without looking at it?  The Bresenham walk stays inside the bounding box of its
// end pixels, so it is when that box lies inside the all-free window around one of them (clr of that end).
__device__ __forceinline__ bool segment_in_clearance(uint32_t ai, uint32_t aj, uint32_t bi, uint32_t bj, uint32_t clr_b) {
    const uint32_t di = ai > bi ? ai - bi : bi - ai, dj = aj > bj ? aj - bj : bj - aj;
    return (di > dj ? di : dj) < clr_b;
}

// class of one state (map_shelves_io.rs:158-163, map_io.rs:165-174); global raster
__device__ __forceinline__ int state_class(const RunConst &rc, double x, double y, uint32_t *err) {
    uint32_t i, j;
    to_pixel(rc, x, y, i, j);
    if (i >= rc.H || j >= rc.W) { *err |= ERR_RASTER; return CLS_HIGH; }
    int c = as_global(rc.cls)[i * rc.W + j];
    if (c == CLS_BAD) { *err |= ERR_RASTER; return CLS_HIGH; }
    return c == CLS_HIGH0 ? CLS_HIGH : c;
}

// PTOFuncs validity ids (map_shelves_io.rs:464-488, map_io.rs:487-513); -1 = None
__device__ __forceinline__ int class_to_validity(const RunConst &rc, int cls) {
    if (cls == CLS_FREE) return rc.n_validities - 1;
    if (cls >= CLS_ZONE) return cls - CLS_ZONE;
    return -1;
}

// GoalFuncs::goal (common.rs:336-345; rrt.rs:330-336 + map_shelves_io.rs:259-265)
template <class Grid>
__device__ bool goal_hit(const RunConst &rc, const Grid &grid, double x, double y, unsigned long long &mask, uint32_t *err) {
    if (rc.goal_kind == 1) {
        for (uint32_t g = 0; g < rc.G; ++g) {
            double d = fabs(rc.gcx[g] - x);
            d += fabs(rc.gcy[g] - y);
            if (d < rc.g_l1) { mask = rc.gmask[g]; return true; }
        }
        return false;
    }
    if (rc.goal_kind == 2) {
        double D = sqrt(dist2(x, y, rc.zone_x, rc.zone_y));
        if (D < rc.visibility) {
            int c = traversed_class(rc, grid, x, y, rc.zone_x, rc.zone_y, err);
            if (c != CLS_HIGH) { mask = 1ull; return true; }
        }
    }
    return false;
}

// ------------------------------------------------------------------ Pcg64 on the device
typedef unsigned __int128 u128;
__device__ __forceinline__ u128 mk128(unsigned long long lo, unsigned long long hi) { return ((u128)hi << 64) | lo; }

struct PcgJump { // LCG jump tables: entry i advances 2^i steps
    unsigned long long mult_lo[64], mult_hi[64], plus_lo[64], plus_hi[64];
};

__device__ __forceinline__ unsigned long long pcg_output(u128 s) {
    uint32_t rot = (uint32_t)(s >> 122);
    unsigned long long xsl = (unsigned long long)(s >> 64) ^ (unsigned long long)s;
    return (xsl >> rot) | (xsl << ((64 - rot) & 63));
}

// rand 0.8 UniformFloat<f64>::sample_single, one draw; *retry is set when the reference would redraw
__device__ __forceinline__ double gen_range_once(unsigned long long r, double low, double high, bool *retry) {
    double v12 = __longlong_as_double((long long)((r >> 12) | 0x3FF0000000000000ull));
    double v01 = v12 - 1.0;
    double scale = high - low;
    double prod = v01 * scale;
    double res = prod + low;
    if (!(res < high)) *retry = true;
    return res;
}

// One thread per iteration of the run: goal-biased sample (rrt.rs:176-181, pto.rs:141-149).
__global__ void k_gen_samples(const RunConst *__restrict__ rcp, const PcgJump *__restrict__ jt, unsigned long long it0,
                              unsigned long long n, unsigned long long st_lo, unsigned long long st_hi,
                              unsigned long long inc_lo, unsigned long long inc_hi, unsigned long long draws_before) {
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row; jt == nullptr: table and sampler state from the run constants
    if (!jt) { jt = rc.jump; st_lo = rc.rng_st_lo; st_hi = rc.rng_st_hi; inc_lo = rc.rng_inc_lo; inc_hi = rc.rng_inc_hi; }
    unsigned long long t = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (rc.sched_nb && it0 + t >= rc.sched_max) return;       // a batch row with its own budget: its arrays end there
    unsigned long long idx = it0 + t;     // 0-based iteration of this grow call
    unsigned long long it = idx + 1;
    if (it % 100 == 0) {
        uint32_t w = rc.mode == 1 ? rc.sworld[idx] : 0u;
        double gx = 0.0, gy = 0.0;
        if (rc.goal_kind == 1) { gx = rc.w2g_x[w & 63]; gy = rc.w2g_y[w & 63]; }
        else if (rc.goal_kind == 2) { gx = rc.zone_x; gy = rc.zone_y; }
        rc.sx[idx] = gx;
        rc.sy[idx] = gy;
        return;
    }
    unsigned long long di = idx - idx / 100;   // sampler calls made before this iteration in this grow
    if (rc.inj_xy) {
        unsigned long long e = rc.inj_base + di;
        if (e >= rc.inj_n) { atomicOr(&rc.cnt->err, ERR_RNG_RETRY); rc.sx[idx] = 0.0; rc.sy[idx] = 0.0; return; }
        rc.sx[idx] = rc.inj_xy[2 * e];
        rc.sy[idx] = rc.inj_xy[2 * e + 1];
        return;
    }
    // jump the LCG ahead by 2*di steps from the state at the start of this grow
    (void)draws_before;
    unsigned long long delta = 2ull * di;
    u128 acc_mult = 1, acc_plus = 0;
    for (int b = 0; b < 64 && (delta >> b); ++b) {
        if ((delta >> b) & 1ull) {
            u128 m = mk128(jt->mult_lo[b], jt->mult_hi[b]), p = mk128(jt->plus_lo[b], jt->plus_hi[b]);
            acc_mult = acc_mult * m;
            acc_plus = acc_plus * m + p;
        }
    }
    const u128 MULT = mk128(0x4385DF649FCCF645ull, 0x2360ED051FC65DA4ull);
    u128 inc = mk128(inc_lo, inc_hi);
    u128 s = acc_mult * mk128(st_lo, st_hi) + acc_plus;
    bool retry = false;
    s = s * MULT + inc;
    double x = gen_range_once(pcg_output(s), rc.s_low0, rc.s_up0, &retry);
    s = s * MULT + inc;
    double y = gen_range_once(pcg_output(s), rc.s_low1, rc.s_up1, &retry);
    if (retry) atomicOr(&rc.cnt->err, ERR_RNG_RETRY);
    rc.sx[idx] = x;
    rc.sy[idx] = y;
}

// ------------------------------------------------------------------ bound pyramid
// rep[] holds, for every cell of a 256^2 / 32^2 / 4^2 pyramid over the sampling box, the id of SOME node
// inside that cell (last writer wins; races are harmless).  It is never used to answer a query: it only
// yields an upper bound on the nearest-neighbour distance so that the brute-force scan can reject almost
// every node with the cheap f32 key.  Results do not depend on its content.
constexpr int kRepLevels = 3;
__device__ __forceinline__ int rep_dim(int l) { return l == 0 ? 256 : (l == 1 ? 32 : 4); }
__device__ __forceinline__ int rep_off(int l) { return l == 0 ? 0 : (l == 1 ? 65536 : 65536 + 1024); }
constexpr int kRepTotal = 65536 + 1024 + 16;
constexpr uint32_t kRepCoarseUntil = 32768;     // nodes with lower ids are entered into the coarse levels too
// Behind the ids, for the two finest levels: the named node's POSITION, rounded to f32, 8 bytes a cell (one store, never torn).  k_nn2's
// "is there a node this close?" reads a cell's position with the cell instead of fetching the named node's coordinates by id -- a trip
// to memory less, eighteen random lines per sample less.  The rounding (2^-24 relative per coordinate) is paid for by the test's margin.
constexpr int kRepF = 65536 + 1024;             // cells of levels 0 and 1 (same offsets as the ids)
constexpr int kRepInts = kRepTotal + 2 * kRepF; // the whole buffer in 32-bit words: ids, then positions (all bytes 0xFF = empty: id -1, NaN)
static_assert(kRepTotal % 2 == 0, "the positions start 8-byte aligned");

__device__ __forceinline__ void rep_cell(const RunConst &rc, double x, double y, int G, int &cx, int &cy) {
    double fx = (x - rc.bx0) * rc.binv_w * (double)G, fy = (y - rc.by0) * rc.binv_h * (double)G;
    cx = fx < 0.0 ? 0 : (fx >= (double)G ? G - 1 : (int)fx);
    cy = fy < 0.0 ? 0 : (fy >= (double)G ? G - 1 : (int)fy);
}

__device__ __forceinline__ void rep_insert(const RunConst &rc, double x, double y, int id) {
#pragma unroll
    for (int l = 0; l < kRepLevels; ++l) {
        int cx, cy;
        const int G = rep_dim(l);
        rep_cell(rc, x, y, G, cx, cy);
        rc.rep[rep_off(l) + cy * G + cx] = id;
        if (l < 2) rc.rep_f[rep_off(l) + cy * G + cx] = flt2{(float)x, (float)y};
    }
}

// Upper bound of a sample's squared nearest-neighbour distance from the pyramid: the finest level whose 3x3
// neighbourhood of the sample's cell holds a node with id < N that passes the world filter (lanes 0..8 take one
// cell each).  The sample is wave-uniform.
template <bool PTO>
__device__ __forceinline__ double nn_bound_wave(const RunConst &rc, uint32_t N, double qx, double qy, uint32_t world, uint32_t lane) {
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    double m = INF;
    for (int l = 0; l < kRepLevels; ++l) {
        const int G = rep_dim(l);
        int cx, cy;
        rep_cell(rc, qx, qy, G, cx, cy);
        int r = -1;
        if (lane < 9u) {
            const int x = cx + (int)(lane % 3u) - 1, y = cy + (int)(lane / 3u) - 1;
            if (x >= 0 && y >= 0 && x < G && y < G) r = as_global(rc.rep)[rep_off(l) + y * G + x];
        }
        double d2 = INF;
        if (r >= 0 && (uint32_t)r < N) {
            bool pass = true;
            if (PTO) pass = (as_global(rc.reachA)[r] >> world) & 1ull;
            if (pass) d2 = dist2(as_global(rc.nx)[r], as_global(rc.ny)[r], qx, qy);
        }
        for (int off = 8; off > 0; off >>= 1) {           // lanes 0..15 hold everything
            const double o = __shfl_xor(d2, off);
            d2 = o < d2 ? o : d2;
        }
        m = __shfl(d2, 0);
        if (m < INF) break;
    }
    if (m == INF && !PTO) m = dist2(rc.nx[0], rc.ny[0], qx, qy);      // the root always exists
    return m;
}

// ------------------------------------------------------------------ region pages + near search
// The node coordinates the searches read are kept a second time, bucketed by REGION: a kRG x kRG grid over the
// box of the run, every region owning a list of 64-slot pages (x, y, id per slot; page 0 of region r is page r,
// further pages come from a pool through the directory rg_dir).  A search is still a brute-force scan with the
// reference's exact f64 arithmetic -- over the pages of the regions that meet the query disc's bounding box
// instead of over the whole array.  The cell function is monotone in each coordinate (clamped at the borders), so
// a node within rho of the query in either coordinate lies in a region between cell(q - rho) and cell(q + rho):
// nothing inside the disc is ever skipped, whatever the box is.
// One WAVE serves one sample: lanes <-> the 64 slots of a page (coalesced 1 KiB + 256 B loads, four pages in
// flight), hits are compacted with a ballot.
__device__ __forceinline__ uint32_t rank_before(const RunConst &rc, uint32_t b, uint32_t vwords, uint32_t k);
__device__ __forceinline__ uint32_t wave_sum(uint32_t v);
__device__ void commit_rrt_sample(const RunConst &rc, uint32_t b, uint32_t vwords, uint32_t k, uint32_t lane, uint32_t stride = 64u, bool lag = false);
constexpr int kRG = 40;
constexpr uint32_t kRegions = kRG * kRG;
constexpr uint32_t kPage = 64;
constexpr uint32_t kOccWords = (kRegions + 63u) / 64u;
typedef double dbl2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t region_of(const RunConst &rc, double x, double y) {
    int cx, cy;
    rep_cell(rc, x, y, kRG, cx, cy);
    return (uint32_t)(cy * kRG + cx);
}

// conservative search radius for a threshold on the squared distance (relative + absolute slack cover the
// roundings of d2 and of q -+ rho)
__device__ __forceinline__ double disc_radius(double bound_d2, double qx, double qy) {
    return sqrt(bound_d2) * (1.0 + 1e-9) + 1e-12 * (1.0 + fabs(qx) + fabs(qy));
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// a wave-uniform double, kept in scalar registers
__device__ __forceinline__ double uni_d(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned long long lo = uni((uint32_t)b), hi = uni((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(lo | (hi << 32)));
}

// visit(x, y, id, ok): called by all 64 lanes together; ok = this lane holds a node.  Every node whose region
// meets the box of the disc (q, rho) is visited exactly once.  q, rho and N (tree size) are wave-uniform.
// Three ways to walk, chosen per query / per group of 64 regions by what costs fewer round trips:
//   flat     the box covers so many regions that streaming the id-ordered arrays nx/ny is cheaper
//   sparse   few nodes per region (young tree): lane <-> region, slot by slot
//   pages    lane <-> slot, region after region (four pages in flight)
template <class Visit>
__device__ __forceinline__ void scan_disc(const RunConst &rc, uint32_t b, double qx, double qy, double rho, uint32_t N, uint32_t lane, Visit visit,
                                          uint32_t skip_region = 0xFFFFFFFFu) {
    int cx0, cy0, cx1, cy1;
    rep_cell(rc, qx - rho, qy - rho, kRG, cx0, cy0);
    rep_cell(rc, qx + rho, qy + rho, kRG, cx1, cy1);
    const uint32_t x0 = uni((uint32_t)cx0), y0 = uni((uint32_t)cy0);
    const uint32_t w = uni((uint32_t)(cx1 - cx0 + 1)), nreg = w * uni((uint32_t)(cy1 - cy0 + 1));
    if (nreg * 16u > N) {
        auto gx = as_global(rc.nx), gy = as_global(rc.ny);
        for (uint32_t j0 = 0; j0 < N; j0 += 128u) {
            double x[2], y[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t j = j0 + 64u * u + lane;
                x[u] = gx[j < N ? j : 0u];
                y[u] = gy[j < N ? j : 0u];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (j0 + 64u * u < N) visit(x[u], y[u], (int)(j0 + 64u * u + lane), j0 + 64u * u + lane < N);
        }
        return;
    }
    auto gcnt = as_global(rc.rg_cnt) + (b & 1u) * kRegions;
    auto gdir = as_global(rc.rg_dir);
    auto gxy = as_global(reinterpret_cast<const dbl2 *>(rc.pg_xy));
    auto gid = as_global(rc.pg_id);
    for (uint32_t r0 = 0; r0 < nreg; r0 += 64) {
        const uint32_t r = r0 + lane;
        uint32_t reg = 0, cnt = 0;
        if (r < nreg) {
            const uint32_t ry = r / w;
            reg = (y0 + ry) * kRG + x0 + (r - ry * w);
            cnt = reg == skip_region ? 0u : gcnt[reg];          // a region the caller has already been through
        }
        // sparse (lane <-> region, slot by slot) pays when every region of the group holds fewer nodes than the group has
        // non-empty regions: then `max count` rounds of loads beat `regions` page loads.  Decided with ballots alone.
        const uint32_t npg = (uint32_t)__popcll(__ballot(cnt > 0u));        // pages if no region needs a second one
        if (__ballot(cnt >= npg || cnt > kPage) == 0ull && npg > 1u) {
            uint32_t cmax = cnt;
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(cmax, off);
                cmax = o > cmax ? o : cmax;
            }
            cmax = uni(cmax);
            for (uint32_t s0 = 0; s0 < cmax; s0 += 2) {
                dbl2 v[2];
                int id[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t sl = s0 + u < cnt ? s0 + u : 0u;
                    v[u] = gxy[(size_t)reg * kPage + sl];
                    id[u] = gid[(size_t)reg * kPage + sl];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (s0 + u < cmax) visit(v[u].x, v[u].y, id[u], s0 + u < cnt);
            }
            continue;
        }
        uint32_t page = reg;                                 // the first page of a region is static
        for (uint32_t j = 0;; ++j) {
            unsigned long long m = __ballot(cnt > j * kPage);
            if (!m) break;
            if (j && cnt > j * kPage) page = gdir[(size_t)reg * rc.rg_maxp + j];
            while (m) {
                uint32_t pg[2], pc[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    pg[u] = 0; pc[u] = 0;
                    if (m) {
                        const uint32_t l = (uint32_t)__builtin_ctzll(m);
                        m &= m - 1;
                        pg[u] = (uint32_t)__builtin_amdgcn_readlane((int)page, (int)l);
                        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)l) - j * kPage;
                        pc[u] = c < kPage ? c : kPage;
                    }
                }
                dbl2 v[2];
                int id[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {                 // only the filled slots are fetched
                    const bool ld = lane < pc[u];
                    v[u] = ld ? gxy[(size_t)pg[u] * kPage + lane] : dbl2{0.0, 0.0};
                    id[u] = ld ? gid[(size_t)pg[u] * kPage + lane] : -1;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (pc[u]) visit(v[u].x, v[u].y, id[u], lane < pc[u]);
            }
        }
    }
}

// One wave per sample: nearest neighbour (nearest_neighbor.rs:48-92: minimum of (norm2, id), world filter applied
// after the distance test), steer + point validity (common.rs:215-225, map_shelves_io.rs:158-170,
// map_io.rs:165-181), then the radius search around the steered state (nearest_neighbor.rs:94-126:
// norm2 <= radius  <=>  d2 <= T2 with T2 from the host table) into the sample's neighbour list.
// The launch may carry the PREVIOUS step's rewire phase 2 in extra workgroups (cb = that step, cnb its samples):
// the two touch disjoint data, and the step chain loses a kernel.
// (the search of one sample by one wave; k is wave-uniform)
template <bool PTO>
__device__ __forceinline__ void near_sample(const RunConst &rc, uint32_t b, uint32_t i0, uint32_t vwords, uint32_t k, uint32_t lane) {
    const uint32_t N = uni(as_global(rc.n_at)[b]);
    const double sqx = uni_d(as_global(rc.sx)[i0 + k]), sqy = uni_d(as_global(rc.sy)[i0 + k]);
    uint32_t world = 0;
    if (PTO) world = uni(as_global(rc.sworld)[i0 + k]);
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    double bestD = INF;
    int best = 0x7FFFFFFF;
    // rrt.rs:121 uses the size before insertion, pto.rs:88 after it (loaded here, needed by the radius search below)
    const double T2 = rc.rad_T2[N + (rc.mode == 1 ? 1u : 0u)];
    {
        auto reach = as_global(rc.reachA);
        // thr: no node with d2 above it can win or tie (sqrt is monotone; the factor keeps rounded ties in), so the
        // sqrt -- the expensive part -- is only taken for the few nodes that may improve the lane's best
        double thr = INF;
        auto visit = [&](double x, double y, int id, bool ok) {
            if (!ok) return;
            const double d2 = dist2(x, y, sqx, sqy);
            if (d2 > thr) return;
            bool pass = true;
            if (PTO) pass = (reach[id] >> world) & 1ull;
            if (!pass) return;
            const double D = sqrt(d2);                           // the reference compares rounded distances
            if (D < bestD || (D == bestD && id < best)) { bestD = D; best = id; thr = d2 * (1.0 + 1e-15); }
        };
        auto wave_best = [&]() {
            double rd = bestD;
            int ri = best;
            for (int off = 32; off > 0; off >>= 1) {
                const double od = __shfl_xor(rd, off);
                const int oi = __shfl_xor(ri, off);
                if (od < rd || (od == rd && oi < ri)) { rd = od; ri = oi; }
            }
            // the lane that holds the winner hands over the rest (its threshold and the node's coordinates)
            const unsigned long long own = __ballot(best == ri && bestD == rd);
            const int src = own ? (int)__builtin_ctzll(own) : 0;
            thr = __shfl(thr, src);
            bestD = rd; best = ri;
        };
        // The sample's own region first: the nearest node is almost always there, and its distance is the bound for
        // the disc the remaining regions are taken from.  Only when the region holds nothing usable does the bound
        // come from the pyramid.
        const uint32_t own = uni(region_of(rc, sqx, sqy));
        scan_disc(rc, b, sqx, sqy, 0.0, N, lane, visit);
        wave_best();
        double m2;
        if (best != 0x7FFFFFFF) m2 = thr;                        // d2(best) * (1 + 1e-15)
        else { m2 = nn_bound_wave<PTO>(rc, N, sqx, sqy, world, lane); thr = m2 * (1.0 + 1e-9); }
        scan_disc(rc, b, sqx, sqy, disc_radius(m2, sqx, sqy), N, lane, visit, own);
        wave_best();
    }
    const int nn = best == 0x7FFFFFFF ? 0 : best;   // nothing passed the filter: the root (nearest_neighbor.rs:90)
    const double fx = as_global(rc.nx)[nn], fy = as_global(rc.ny)[nn];
    double tx = sqx, ty = sqy;
    // common.rs:215-225
    double step = fabs(tx - fx);
    step += fabs(ty - fy);
    if (step > rc.max_step) {
        const double lambda = rc.max_step / step;
        double ux = (tx - fx) * lambda, uy = (ty - fy) * lambda;
        tx = fx + ux;
        ty = fy + uy;
    }
    uint32_t err = 0;
    int vid = 0;
    bool valid = true;
    if (rc.has_grid) {
        int cls = state_class(rc, tx, ty, &err);
        if (rc.mode == 0) valid = cls == CLS_FREE;            // RTTFuncs adapter (tamp_rrt.rs:40-42)
        else { vid = class_to_validity(rc, cls); valid = vid >= 0; }
        if (err) valid = false;
    }
    if (lane == 0) {
        const uint32_t qo = q_off(rc, b);
        as_global(rc.q_x)[qo + k] = tx;
        as_global(rc.q_y)[qo + k] = ty;
        // copy for the kd insertion, which runs beside the following steps (one slice per step)
        const size_t o2 = (size_t)b * rc.part_stride + k;
        as_global(rc.kq_x)[o2] = tx; as_global(rc.kq_y)[o2] = ty; as_global(rc.kq_vid)[o2] = valid ? vid : -1;
        as_global(rc.q_nn)[qo + k] = nn;
        as_global(rc.q_vid)[qo + k] = valid ? vid : -1;
        if (valid) atomicOr(&rc.valid_mask[(size_t)b * vwords + (k >> 6)], 1ull << (k & 63u));
        if (err) atomicOr(&rc.cnt->err, err);
    }
    if (!valid) return;
    auto cid = as_global(rc.cand_id) + cand_off(rc, b, k);
    auto cxy = as_global(reinterpret_cast<dbl2 *>(rc.cand_xy)) + cand_off(rc, b, k);
    const uint32_t cap = rc.cand_cap;
    uint32_t tot = 0;
    bool over = false;
    scan_disc(rc, b, tx, ty, disc_radius(T2, tx, ty), N, lane, [&](double x, double y, int id, bool ok) {
        const bool in = ok && dist2(x, y, tx, ty) <= T2;
        const unsigned long long hm = __ballot(in);
        const uint32_t pos = tot + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
        if (in) {
            if (pos < cap) {
                cid[pos] = id;
                dbl2 v;
                v.x = x; v.y = y;
                cxy[pos] = v;
            } else {
                over = true;
            }
        }
        tot += (uint32_t)__popcll(hm);
    });
    if (lane == 0) as_global(rc.cand_cnt)[cand_cnt_at(rc, b, k)] = tot;
    if (over) atomicOr(&rc.cnt->err, (uint32_t)ERR_CAND_OVERFLOW);
}

template <bool PTO>
__global__ __launch_bounds__(256) void k_near(const RunConst *__restrict__ rcp, uint32_t b, uint32_t i0, uint32_t nb, uint32_t vwords, uint32_t cb,
                                              uint32_t cnb) {
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row (porrt_grow_batch)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t near_blocks = (nb + 3u) / 4u;
    if (blockIdx.x >= near_blocks) {
        const uint32_t ck = uni((blockIdx.x - near_blocks) * 4u + (threadIdx.x >> 6));     // wave-uniform: addresses in SGPRs
        if (!PTO && ck < cnb && ck < row_nb(rc, cb, cnb)) commit_rrt_sample(rc, cb, vwords, ck, lane);
        return;
    }
    const uint32_t k = uni(blockIdx.x * 4u + (threadIdx.x >> 6));       // wave-uniform: addresses in SGPRs
    if (k >= row_nb(rc, b, nb)) return;
    near_sample<PTO>(rc, b, row_i0(rc, b, i0), vwords, k, lane);
}

// Add the step's new nodes to the region pages (run by ONE workgroup, an extra block of the connect kernels:
// positions and validity are final since k_near, ids follow from the valid mask).  Nothing reads the pages
// between k_near of this step and k_near of the next.
constexpr uint32_t kInsertLds = kRegions * 4u + 4096u * 2u + 16u;       // bytes of LDS scratch insert_step_pages needs
__device__ void insert_step_pages(const RunConst &rc, uint32_t b, uint32_t nb, uint32_t vwords, uint8_t *scratch) {
    uint32_t *s_add = reinterpret_cast<uint32_t *>(scratch);                            // [kRegions]
    uint16_t *s_off = reinterpret_cast<uint16_t *>(scratch + kRegions * 4u);            // [4096]
    uint32_t &s_np = *reinterpret_cast<uint32_t *>(scratch + kRegions * 4u + 4096u * 2u);
    uint32_t &s_base = *reinterpret_cast<uint32_t *>(scratch + kRegions * 4u + 4096u * 2u + 4u);
    const uint32_t T = blockDim.x;
    // the step's searches run beside this workgroup: they read the counts of parity b (and only slots below them);
    // the new counts go to the other parity, for the next step
    auto rg_old = as_global(rc.rg_cnt) + (b & 1u) * kRegions, rg_new = as_global(rc.rg_cnt) + ((b + 1u) & 1u) * kRegions;
    for (uint32_t r = threadIdx.x; r < kRegions; r += T) s_add[r] = 0;
    if (threadIdx.x < 64u) {                 // (the first wave: a word of the valid mask per lane)
        uint32_t add = threadIdx.x < vwords ? (uint32_t)__popcll(as_global(rc.valid_mask)[(size_t)b * vwords + threadIdx.x]) : 0u;
        add = wave_sum(add);
        if (threadIdx.x == 0) {
        s_np = 0; s_base = kRegions + rc.cnt->n_pages;
        const uint32_t n_next = as_global(rc.n_at)[b] + add;
        as_global(rc.n_at)[b + 1] = n_next;                              // tree size at the start of the next step
        as_global(rc.t2_at)[b + 1] = as_global(rc.rad_T2)[n_next];
        }
    }
    __syncthreads();
    const uint32_t qo = q_off(rc, b);
    for (uint32_t k = threadIdx.x; k < nb; k += T)
        if (rc.q_vid[qo + k] >= 0) s_off[k] = (uint16_t)atomicAdd(&s_add[region_of(rc, as_global(rc.q_x)[qo + k], as_global(rc.q_y)[qo + k])], 1u);
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < kRegions; r += T) {
        const uint32_t add = s_add[r];
        if (!add) continue;
        const uint32_t old = rg_old[r];
        const uint32_t p_old = old ? (old + kPage - 1) / kPage : 1u, p_new = (old + add + kPage - 1) / kPage;
        if (p_new > p_old) {
            const uint32_t need = p_new - p_old;
            const uint32_t at = s_base + atomicAdd(&s_np, need);
            if (at + need > rc.pg_cap || p_new > rc.rg_maxp) { atomicOr(&rc.cnt->err, (uint32_t)ERR_PAGE_OVERFLOW); continue; }
            for (uint32_t i = 0; i < need; ++i) as_global(rc.rg_dir)[(size_t)r * rc.rg_maxp + p_old + i] = at + i;
        }
    }
    __syncthreads();
    if (rc.cnt->err & ERR_PAGE_OVERFLOW) return;
    const uint32_t N = as_global(rc.n_at)[b];
    dbl2 *pxy = reinterpret_cast<dbl2 *>(rc.pg_xy);
    for (uint32_t k = threadIdx.x; k < nb; k += T) {
        if (rc.q_vid[qo + k] < 0) continue;
        const double x = as_global(rc.q_x)[qo + k], y = as_global(rc.q_y)[qo + k];
        const uint32_t r = region_of(rc, x, y);
        const uint32_t slot = rg_old[r] + s_off[k], j = slot / kPage;
        const uint32_t page = j ? __hip_atomic_load(&rc.rg_dir[(size_t)r * rc.rg_maxp + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : r;
        dbl2 v;
        v.x = x; v.y = y;
        pxy[(size_t)page * kPage + (slot % kPage)] = v;
        const uint32_t nid = N + rank_before(rc, b, vwords, k);
        as_global(rc.pg_id)[(size_t)page * kPage + (slot % kPage)] = (int)nid;
        as_global(rc.slot_of)[nid] = page * kPage + (slot % kPage);          // pg_d of the slot: the step's commit pass
        // (with pipelined steps the next step's search reads the coordinates before the connect pass has stored them: same bits)
        if (rc.q_stride) { as_global(rc.nx)[nid] = x; as_global(rc.ny)[nid] = y; }
    }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < kRegions; r += T) rg_new[r] = rg_old[r] + s_add[r];
    if (threadIdx.x < kOccWords) {          // (rg_old and s_add are final; a region never empties)
        unsigned long long w = 0;
        for (uint32_t i = 0; i < 64u; ++i) {
            const uint32_t r = threadIdx.x * 64u + i;
            if (r < kRegions && rg_old[r] + s_add[r] > 0u) w |= 1ull << i;
        }
        as_global(rc.rg_occ)[((b + 1u) & 1u) * kOccWords + threadIdx.x] = w;
    }
    if (threadIdx.x == 0) rc.cnt->n_pages += s_np;
}

// The same filing for k_file_commit (pipelined steps), where it is ON the chain of dependent kernels: one workgroup of 1024
// threads, a thread per sample, and every load that does not depend on another thread's work issued before the first
// barrier (the samples, the old counts of all regions -- kept in LDS --, the valid mask, the page counter): one round trip
// to memory instead of five.  Same result as insert_step_pages up to the order of a step's nodes inside a region's
// pages, which nothing depends on.
constexpr uint32_t kFileLds = 2u * kRegions * 4u + 4096u * 2u + 64u * 8u + 66u * 4u + 32u;
template <uint32_t T = 1024u>
__device__ void file_step_fast(const RunConst &rc, uint32_t b, uint32_t nb, uint32_t vwords, uint8_t *scratch) {
    uint32_t *s_add = reinterpret_cast<uint32_t *>(scratch);                               // [kRegions] new nodes per region
    uint32_t *s_old = s_add + kRegions;                                                    // [kRegions] counts before the step
    uint16_t *s_off = reinterpret_cast<uint16_t *>(s_old + kRegions);                      // [4096] a sample's place among its region's new nodes
    unsigned long long *s_word = reinterpret_cast<unsigned long long *>(s_off + 4096);     // [64] the valid mask
    uint32_t *s_pref = reinterpret_cast<uint32_t *>(s_word + 64);                          // [65] valid samples before word w
    uint32_t *s_misc = s_pref + 66;                                                        // np, base, err, N
    constexpr uint32_t SPT = 4u;                                                           // nb <= 4 T (1024 threads: batch_K <= 4096)
    const uint32_t tid = threadIdx.x, qo = q_off(rc, b);
    auto rg_old = as_global(rc.rg_cnt) + (b & 1u) * kRegions, rg_new = as_global(rc.rg_cnt) + ((b + 1u) & 1u) * kRegions;
    // ---- loads, all in flight together
    int vid[SPT];
    double x[SPT], y[SPT];
#pragma unroll
    for (uint32_t q = 0; q < SPT; ++q) {
        const uint32_t k = tid + q * T;
        vid[q] = k < nb ? as_global(rc.q_vid)[qo + k] : -1;
        x[q] = k < nb ? as_global(rc.q_x)[qo + k] : 0.0;
        y[q] = k < nb ? as_global(rc.q_y)[qo + k] : 0.0;
    }
    constexpr uint32_t RPT = (kRegions + T - 1u) / T;                                              // regions per thread
    uint32_t oldc[RPT];
#pragma unroll
    for (uint32_t h = 0; h < RPT; ++h) oldc[h] = tid + h * T < kRegions ? rg_old[tid + h * T] : 0u;
    const unsigned long long word = tid < vwords ? as_global(rc.valid_mask)[(size_t)b * vwords + tid] : 0ull;
    uint32_t n_pages = 0, N0 = 0;
    if (tid == T - 1u) { n_pages = rc.cnt->n_pages; N0 = as_global(rc.n_at)[b]; }
#pragma unroll
    for (uint32_t h = 0; h < RPT; ++h) if (tid + h * T < kRegions) { s_add[tid + h * T] = 0; s_old[tid + h * T] = oldc[h]; }
    if (tid < 64u) s_word[tid] = word;
    if (tid == T - 1u) { s_misc[0] = 0; s_misc[1] = kRegions + n_pages; s_misc[2] = 0; s_misc[3] = N0; }
    __syncthreads();
    double t2_next = 0.0;
    if (tid == 0) {
        uint32_t acc = 0;
        for (uint32_t w = 0; w < vwords; ++w) { s_pref[w] = acc; acc += (uint32_t)__popcll(s_word[w]); }
        const uint32_t n_next = s_misc[3] + acc;
        as_global(rc.n_at)[b + 1] = n_next;                              // tree size at the start of the next step
        t2_next = as_global(rc.rad_T2)[n_next];                          // (in flight while the others go on; stored at the end)
    }
    uint32_t reg[SPT];
#pragma unroll
    for (uint32_t q = 0; q < SPT; ++q) {
        reg[q] = 0;
        if (vid[q] >= 0) { reg[q] = region_of(rc, x[q], y[q]); s_off[tid + q * T] = (uint16_t)atomicAdd(&s_add[reg[q]], 1u); }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t h = 0; h < RPT; ++h) {
        const uint32_t r = tid + h * T;
        const uint32_t add = r < kRegions ? s_add[r] : 0u;
        if (!add) continue;
        const uint32_t old = s_old[r];
        const uint32_t p_old = old ? (old + kPage - 1) / kPage : 1u, p_new = (old + add + kPage - 1) / kPage;
        if (p_new > p_old) {
            const uint32_t need = p_new - p_old;
            const uint32_t at = s_misc[1] + atomicAdd(&s_misc[0], need);
            if (at + need > rc.pg_cap || p_new > rc.rg_maxp) { atomicOr(&rc.cnt->err, (uint32_t)ERR_PAGE_OVERFLOW); s_misc[2] = 1u; continue; }
            for (uint32_t i = 0; i < need; ++i) as_global(rc.rg_dir)[(size_t)r * rc.rg_maxp + p_old + i] = at + i;
        }
    }
    __threadfence_block();
    __syncthreads();
    if (s_misc[2]) return;
    const uint32_t N = s_misc[3];
    dbl2 *pxy = reinterpret_cast<dbl2 *>(rc.pg_xy);
#pragma unroll
    for (uint32_t q = 0; q < SPT; ++q) {
        if (vid[q] < 0) continue;
        const uint32_t k = tid + q * T, r = reg[q];
        const uint32_t slot = s_old[r] + s_off[k], j = slot / kPage;
        const uint32_t page = j ? __hip_atomic_load(&rc.rg_dir[(size_t)r * rc.rg_maxp + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : r;
        dbl2 v;
        v.x = x[q]; v.y = y[q];
        pxy[(size_t)page * kPage + (slot % kPage)] = v;
        const uint32_t nid = N + s_pref[k >> 6] + (uint32_t)__popcll(s_word[k >> 6] & ((1ull << (k & 63u)) - 1ull));
        as_global(rc.pg_id)[(size_t)page * kPage + (slot % kPage)] = (int)nid;
        as_global(rc.slot_of)[nid] = page * kPage + (slot % kPage);
        as_global(rc.nx)[nid] = x[q];        // the next step's search reads the coordinates before the connect pass has stored them: same bits
        as_global(rc.ny)[nid] = y[q];
    }
#pragma unroll
    for (uint32_t h = 0; h < RPT; ++h) {
        const uint32_t r = tid + h * T;
        if (r < kRegions) rg_new[r] = s_old[r] + s_add[r];
    }
    if (tid < kOccWords) {          // (a region never empties)
        unsigned long long w = 0;
        for (uint32_t i = 0; i < 64u; ++i) {
            const uint32_t r = tid * 64u + i;
            if (r < kRegions && s_old[r] + s_add[r] > 0u) w |= 1ull << i;
        }
        as_global(rc.rg_occ)[((b + 1u) & 1u) * kOccWords + tid] = w;
    }
    if (tid == 0) { as_global(rc.t2_at)[b + 1] = t2_next; rc.cnt->n_pages += s_misc[0]; }
}

// ------------------------------------------------------------------ connect
__device__ __forceinline__ uint32_t rank_before(const RunConst &rc, uint32_t b, uint32_t vwords, uint32_t k) {
    uint32_t r = 0;
    auto vm = as_global(rc.valid_mask) + (size_t)b * vwords;
    for (uint32_t w = 0; w < (k >> 6); ++w) r += __popcll(vm[w]);
    r += __popcll(vm[k >> 6] & ((1ull << (k & 63u)) - 1ull));
    return r;
}
// Data movement inside a row of 16 lanes by DPP (a modifier of a vector move: no trip through the LDS crossbar, which is what
// __shfl / ds_bpermute takes): rotate the row right by n lanes, or give every lane of a row the value of its lane n.  A group of 16
// lanes per sample is exactly a DPP row; reductions over the group are four rotate-and-combine steps that leave every lane with the
// result (any order of a commutative, associative combination gives the same bits: sums of integers, minima, lexicographic minima).
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned long long lo = dpp_u32<CTRL>((uint32_t)b), hi = dpp_u32<CTRL>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(lo | (hi << 32)));
}
constexpr int kDppRowRor = 0x120;        // + n: row_ror:n
constexpr int kDppRowBcast = 0x150;      // + n: row_newbcast:n (gfx90a and later)
#ifndef PORRT_DPP
#define PORRT_DPP 1
#endif

// The same by the `stride` lanes (a power of two <= 64, aligned in the wave) that serve sample k together, called by all of them:
// a word of the mask per lane and a sum over the lanes instead of every lane reading every word (k is not wave-uniform when a
// wave holds several samples, so those would be vector loads: up to 16 per lane at K = 1024).
__device__ __forceinline__ uint32_t rank_before_lanes(const RunConst &rc, uint32_t b, uint32_t vwords, uint32_t k, uint32_t lane, uint32_t stride) {
    auto vm = as_global(rc.valid_mask) + (size_t)b * vwords;
    const uint32_t wk = k >> 6;
    uint32_t r = 0;
    for (uint32_t w0 = 0; w0 <= wk; w0 += stride) {
        const uint32_t w = w0 + lane;
        unsigned long long v = w <= wk ? vm[w] : 0ull;
        if (w == wk) v &= (1ull << (k & 63u)) - 1ull;
        r += (uint32_t)__popcll(v);
    }
    if (PORRT_DPP && stride == 16u) {          // (a compile-time constant at every call: the group size)
        r += dpp_u32<kDppRowRor + 8>(r); r += dpp_u32<kDppRowRor + 4>(r); r += dpp_u32<kDppRowRor + 2>(r); r += dpp_u32<kDppRowRor + 1>(r);
        return r;
    }
    for (uint32_t off = stride >> 1; off > 0; off >>= 1) r += (uint32_t)__shfl_xor((int)r, (int)off);
    return r;
}

// Fill the calling wave's LDS tile with the (2R+1)^2 raster window centred on the pixel of (px,py).
__device__ __forceinline__ TileGrid load_tile(const RunConst &rc, uint8_t *tile, double px, double py, uint32_t lane, uint32_t stride) {
    TileGrid g;
    g.lds = tile; g.glob = rc.cls; g.W = rc.W; g.TW = 2u * rc.tile_R + 1u;
    uint32_t ci, cj;
    to_pixel(rc, px, py, ci, cj);
    g.oi = (int)ci - (int)rc.tile_R;
    g.oj = (int)cj - (int)rc.tile_R;
    const uint32_t n = g.TW * g.TW;
    for (uint32_t t = lane; t < n; t += stride) {
        const uint32_t ri = t / g.TW, rj = t - ri * g.TW;
        const int i = g.oi + (int)ri, j = g.oj + (int)rj;
        uint8_t c = CLS_BAD;
        if (i >= 0 && j >= 0 && (uint32_t)i < rc.H && (uint32_t)j < rc.W) c = as_global(rc.cls)[(uint32_t)i * rc.W + (uint32_t)j];
        tile[t] = c;
    }
    return g;
}

template <class T>
__device__ __forceinline__ T wave_min_u(T v) {
    for (int off = 32; off > 0; off >>= 1) {
        T o = __shfl_xor(v, off);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ unsigned long long wave_or(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off);
    return v;
}

// left of node w at tree depth `depth` (nearest_neighbor.rs:32: state[axis] < current.state[axis])
__device__ __forceinline__ bool kd_left(double x, double y, double wx, double wy, uint32_t depth) {
    return (depth & 1u) ? (y < wy) : (x < wx);
}

// True iff node u is visited before node v (u != v) by a pre-order walk of the reference's kd-tree -- the
// order in which KdTree::nearest_neighbors lists its results (nearest_neighbor.rs:101-117).  Every root path
// starts along the goal path G and leaves it at depth kd_gexit (or stays on it), which settles most pairs
// in O(1); two nodes that leave G at the same node are compared by walking up to their common ancestor.
__device__ bool kd_preorder_less(const RunConst &rc, int u, int v) {
    const uint32_t gu = as_global(rc.kd_gexit)[u], gv = as_global(rc.kd_gexit)[v];
    const bool ou = gu & kOnG, ov = gv & kOnG;
    const uint32_t iu = gu & ~kOnG, iv = gv & ~kOnG;
    if (ou && ov) return iu < iv;                       // both on G: the ancestor comes first
    if (ou) {                                           // u = G[iu]; v leaves G after G[iv]
        if (iu <= iv) return true;                      // u is an ancestor of v
        // u lies below G[iv] on the goal side, v on the other side
        return !kd_left(as_global(rc.nx)[v], as_global(rc.ny)[v], as_global(rc.g_x)[iv], as_global(rc.g_y)[iv], iv);
    }
    if (ov) {
        if (iv <= iu) return false;                     // v is an ancestor of u
        return kd_left(as_global(rc.nx)[u], as_global(rc.ny)[u], as_global(rc.g_x)[iu], as_global(rc.g_y)[iu], iu);
    }
    if (iu != iv) {                                     // both off G: the one leaving first splits them
        if (iu < iv) return kd_left(as_global(rc.nx)[u], as_global(rc.ny)[u], as_global(rc.g_x)[iu], as_global(rc.g_y)[iu], iu);
        return !kd_left(as_global(rc.nx)[v], as_global(rc.ny)[v], as_global(rc.g_x)[iv], as_global(rc.g_y)[iv], iv);
    }
    // same exit node, same (non-goal) side: plain LCA walk, bounded by the depth below the exit node
    int a = u, b = v;
    uint32_t da = as_global(rc.kd_depth)[a], db = as_global(rc.kd_depth)[b];
    int a_from = -1, b_from = -1;       // 0 = came up from a left child, 1 = right
    while (da > db) { const int p = as_global(rc.kd_up)[a]; a_from = as_global(rc.kd_rec)[p].child[1] == a; a = p; --da; }
    while (db > da) { const int p = as_global(rc.kd_up)[b]; b_from = as_global(rc.kd_rec)[p].child[1] == b; b = p; --db; }
    if (a == b) return a_from >= 0 ? false : true;                 // the one that did not move is the ancestor
    while (a != b) {
        const int pa = as_global(rc.kd_up)[a], pb = as_global(rc.kd_up)[b];
        a_from = as_global(rc.kd_rec)[pa].child[1] == a;
        b_from = as_global(rc.kd_rec)[pb].child[1] == b;
        a = pa; b = pb;
    }
    return a_from < b_from;
}

// ---- team reductions: a team serves one sample.  Team<W>: one wave (W = 1) or the W waves of a workgroup sharing
// `scr`; GTeam<GL> (below): GL = 16 / 32 lanes of a wave, several samples per wave.
template <int W>
struct Team {
    static constexpr bool kOneWave = W == 1;
    static constexpr uint32_t kSize = W * 64u;
    double *scr_d;      // W doubles
    int *scr_i;         // W ints
    uint32_t wave, lane;
    __device__ __forceinline__ uint32_t tl() const { return wave * 64u + lane; }
    __device__ __forceinline__ uint32_t sub() const { return lane; }          // lane inside the shuffle domain
    __device__ __forceinline__ unsigned long long ballot(bool p) const { return __ballot(p); }
    template <class T>
    __device__ __forceinline__ T shfl(T v, int src) const { return __shfl(v, src); }
    __device__ __forceinline__ void sync() const { if (W > 1) __syncthreads(); }
    __device__ __forceinline__ uint32_t sum(uint32_t v) const {
        v = wave_sum(v);
        if (W == 1) return v;
        sync();
        if (lane == 0) scr_i[wave] = (int)v;
        sync();
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) t += (uint32_t)scr_i[w];
        return t;
    }
    __device__ __forceinline__ int min_i(int v) const {
        v = wave_min_u(v);
        if (W == 1) return v;
        sync();
        if (lane == 0) scr_i[wave] = v;
        sync();
        int t = scr_i[0];
#pragma unroll
        for (int w = 1; w < W; ++w) t = scr_i[w] < t ? scr_i[w] : t;
        return t;
    }
    __device__ __forceinline__ int max_i(int v) const { return -min_i(-v); }
    // values of the team's first lane
    __device__ __forceinline__ void bcast2(uint32_t &a, uint32_t &b) const {
        a = (uint32_t)__shfl((int)a, 0);
        b = (uint32_t)__shfl((int)b, 0);
        if (W == 1) return;
        sync();
        if (wave == 0 && lane == 0) { scr_i[0] = (int)a; scr_i[1] = (int)b; }
        sync();
        a = (uint32_t)scr_i[0]; b = (uint32_t)scr_i[1];
        sync();
    }
    // lexicographic min of (t, j)
    __device__ __forceinline__ void argmin(double &t, int &j) const {
        for (int off = 32; off > 0; off >>= 1) {
            const double ot = __shfl_xor(t, off);
            const int oj = __shfl_xor(j, off);
            if (ot < t || (ot == t && oj < j)) { t = ot; j = oj; }
        }
        if (W == 1) return;
        sync();
        if (lane == 0) { scr_d[wave] = t; scr_i[wave] = j; }
        sync();
        t = scr_d[0]; j = scr_i[0];
#pragma unroll
        for (int w = 1; w < W; ++w)
            if (scr_d[w] < t || (scr_d[w] == t && scr_i[w] < j)) { t = scr_d[w]; j = scr_i[w]; }
    }
    // kd pre-order first of the team's candidates (kEmpty = none)
    __device__ __forceinline__ int first_preorder(const RunConst &rc, int v) const {
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(v, off);
            if (o != kEmpty && (v == kEmpty || (o != v && kd_preorder_less(rc, o, v)))) v = o;
        }
        if (W == 1) return v;
        sync();
        if (lane == 0) scr_i[wave] = v;
        sync();
        int t = scr_i[0];
#pragma unroll
        for (int w = 1; w < W; ++w) {
            const int o = scr_i[w];
            if (o != kEmpty && (t == kEmpty || (o != t && kd_preorder_less(rc, o, t)))) t = o;
        }
        return t;
    }
};

// GL consecutive lanes of a wave serving one sample (64 / GL samples per wave).  Control flow around every call is
// uniform per group; other groups of the wave may be masked off, so nothing here reads a lane outside the group.
template <int GL>
struct GTeam {
    static constexpr bool kOneWave = true;
    static constexpr uint32_t kSize = GL;
    uint32_t gl, base;      // lane inside the group, first lane of the group inside the wave
    __device__ __forceinline__ uint32_t tl() const { return gl; }
    __device__ __forceinline__ uint32_t sub() const { return gl; }
    __device__ __forceinline__ unsigned long long ballot(bool p) const {
        const unsigned long long m = __ballot(p) >> base;
        return GL == 64 ? m : (m & ((1ull << (GL & 63)) - 1ull));
    }
    template <class T>
    __device__ __forceinline__ T shfl(T v, int src) const { return __shfl(v, (int)base + src); }
    __device__ __forceinline__ void sync() const {}
    __device__ __forceinline__ uint32_t sum(uint32_t v) const {
        if constexpr (GL == 16 && PORRT_DPP) {
            v += dpp_u32<kDppRowRor + 8>(v); v += dpp_u32<kDppRowRor + 4>(v); v += dpp_u32<kDppRowRor + 2>(v); v += dpp_u32<kDppRowRor + 1>(v);
            return v;
        }
        for (int off = GL / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
        return v;
    }
    __device__ __forceinline__ int min_i(int v) const {
        if constexpr (GL == 16 && PORRT_DPP) {
            int o;
            o = (int)dpp_u32<kDppRowRor + 8>((uint32_t)v); v = o < v ? o : v;
            o = (int)dpp_u32<kDppRowRor + 4>((uint32_t)v); v = o < v ? o : v;
            o = (int)dpp_u32<kDppRowRor + 2>((uint32_t)v); v = o < v ? o : v;
            o = (int)dpp_u32<kDppRowRor + 1>((uint32_t)v); v = o < v ? o : v;
            return v;
        }
        for (int off = GL / 2; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o < v ? o : v; }
        return v;
    }
    __device__ __forceinline__ int max_i(int v) const { return -min_i(-v); }
    __device__ __forceinline__ void bcast2(uint32_t &a, uint32_t &b) const {
        if constexpr (GL == 16 && PORRT_DPP) { a = dpp_u32<kDppRowBcast>(a); b = dpp_u32<kDppRowBcast>(b); return; }
        a = (uint32_t)__shfl((int)a, (int)base);
        b = (uint32_t)__shfl((int)b, (int)base);
    }
    template <int N>
    __device__ __forceinline__ void argmin_step(double &t, int &j) const {
        const double ot = dpp_f64<kDppRowRor + N>(t);
        const int oj = (int)dpp_u32<kDppRowRor + N>((uint32_t)j);
        if (ot < t || (ot == t && oj < j)) { t = ot; j = oj; }
    }
    __device__ __forceinline__ void argmin(double &t, int &j) const {
        if constexpr (GL == 16 && PORRT_DPP) { argmin_step<8>(t, j); argmin_step<4>(t, j); argmin_step<2>(t, j); argmin_step<1>(t, j); return; }
        for (int off = GL / 2; off > 0; off >>= 1) {
            const double ot = __shfl_xor(t, off);
            const int oj = __shfl_xor(j, off);
            if (ot < t || (ot == t && oj < j)) { t = ot; j = oj; }
        }
    }
    __device__ __forceinline__ int first_preorder(const RunConst &rc, int v) const {
        for (int off = GL / 2; off > 0; off >>= 1) {
            const int o = __shfl_xor(v, off);
            if (o != kEmpty && (v == kEmpty || (o != v && kd_preorder_less(rc, o, v)))) v = o;
        }
        return v;
    }
};

// ---- neighbour lists of one sample, as connect_rrt_sample sees them
// GlobalList: the lists k_near wrote (ids + coordinates per step parity, values per sample); after the connect pass
// val[a] holds every neighbour's rewire candidate (or < 0) for the commit pass.
struct GlobalList {
    static constexpr bool kCompact = false;
    GPTR(const int) cid;
    GPTR(const dbl2) cxy;
    GPTR(double) cval;
    GPTR(const double) gdA;
    __device__ __forceinline__ int id(uint32_t a) const { return cid[a]; }
    __device__ __forceinline__ void xy(uint32_t a, double &x, double &y) const { const dbl2 v = cxy[a]; x = v.x; y = v.y; }
    __device__ __forceinline__ double val(uint32_t a) const { return cval[a]; }
    __device__ __forceinline__ void set_val(uint32_t a, double v) const { cval[a] = v; }
    __device__ __forceinline__ void set_dA(uint32_t, double) const {}
    __device__ __forceinline__ double dA(uint32_t, int j) const { return gdA[j]; }
};
__device__ __forceinline__ GlobalList global_list(const RunConst &rc, uint32_t b, uint32_t k) {
    GlobalList L;
    L.cid = as_global((const int *)rc.cand_id) + cand_off(rc, b, k);
    L.cxy = as_global(reinterpret_cast<const dbl2 *>(rc.cand_xy)) + cand_off(rc, b, k);
    L.cval = as_global(rc.cand_val) + cand_val_off(rc, b, k);
    L.gdA = as_global((const double *)rc.distA);
    return L;
}
// The hits of a radius search as k_conn2 keeps them: id, coordinates (x is replaced by the cost once the ray is through,
// then by nothing: the rewire candidates leave through out_*) and dist_root, which came with the coordinates from the
// region page.  LdsHits: a sample with at most kLdsHits hits, everything in LDS.  MemHits: more hits (dense starts of a
// tree, neighbourhoods of the goal point), everything in the sample's slice of the global list arrays.  Either way only
// the actual rewire candidates are left in memory for the commit pass, compacted -- for MemHits in place: a candidate's
// position is at most its entry index, and an entry is read before anything is written in its round.
#ifndef PORRT_LDS_HITS
#define PORRT_LDS_HITS 80
#endif
constexpr uint32_t kLdsHits = PORRT_LDS_HITS;        // (even: the next sample's doubles stay 8-byte aligned)
static_assert(kLdsHits % 2u == 0u, "kLdsHits");
constexpr uint32_t kHitBytes = kLdsHits * (4u + 8u + 8u + 8u);
struct LdsHits {
    static constexpr bool kCompact = true;
    int *hid;
    double *hx, *hy, *hd;
    GPTR(int) out_id;       // compact (id, candidate dist_root) list of the sample, read by commit_rrt_sample
    GPTR(double) out_val;
    GPTR(uint32_t) out_cnt;
    uint32_t out_cap;
    __device__ __forceinline__ int id(uint32_t a) const { return hid[a]; }
    __device__ __forceinline__ void xy(uint32_t a, double &x, double &y) const { x = hx[a]; y = hy[a]; }
    __device__ __forceinline__ double val(uint32_t a) const { return hx[a]; }
    __device__ __forceinline__ void set_val(uint32_t a, double v) const { hx[a] = v; }
    __device__ __forceinline__ void set_dA(uint32_t, double) const {}
    __device__ __forceinline__ double dA(uint32_t a, int) const { return hd[a]; }
};
struct MemHits {
    static constexpr bool kCompact = true;
    GPTR(int) sid;
    GPTR(dbl2) sxy;
    GPTR(double) sd;
    GPTR(int) out_id;
    GPTR(double) out_val;
    GPTR(uint32_t) out_cnt;
    uint32_t out_cap;
    __device__ __forceinline__ int id(uint32_t a) const { return sid[a]; }
    __device__ __forceinline__ void xy(uint32_t a, double &x, double &y) const { const dbl2 v = sxy[a]; x = v.x; y = v.y; }
    __device__ __forceinline__ double val(uint32_t a) const { return sxy[a].x; }
    __device__ __forceinline__ void set_val(uint32_t a, double v) const { sxy[a].x = v; }
    __device__ __forceinline__ void set_dA(uint32_t, double) const {}
    __device__ __forceinline__ double dA(uint32_t a, int) const { return sd[a]; }
};

// RRT*: validated neighbours, best parent, new node, rewire phase 1 for sample k by a team (Team<W> / GTeam<GL>) on a
// neighbour list (GlobalList / HitList).  A lane keeps its first candidate in registers through all three passes (that
// is every candidate when the sample has at most `team size` neighbours, the common case); further candidates go
// through the list's value slots.
struct NoNearestSearch { __device__ int operator()() const { return 0; } };      // k_near always names the nearest node
template <class TeamT, class ListT, class Grid, class NNF = NoNearestSearch>
__device__ void connect_rrt_sample(const RunConst &rc, const TeamT &tm, const ListT &L, const Grid &grid, uint32_t b, uint32_t k, uint32_t id,
                                   double px, double py, uint32_t cnt, uint32_t &err, const uint32_t *clone_ids = nullptr, uint32_t n_clone = 0,
                                   int clr_known = -1, NNF nearest_search = NNF(), uint32_t swap = 0) {
    // swap (the one-kernel-per-step form, k_step1_rrt): the two dist_root arrays trade places every step -- the step's snapshot is
    // distB and its rewire accumulator distA on odd steps -- so that a step's rewire commit can run beside the next step's connect
    // clone_ids: further new nodes of this step at exactly (px, py), all with ids above `id` (the copies of the goal
    // point a step adds, rrt.rs:176-181).  The reference would run the same search n_clone + 1 times on the same
    // snapshot: same neighbours, same costs, same parent, same dist_root; of the rewires only the first copy's are
    // strict improvements (rrt.rs:157).  They get their nodes (and their own deferred-tie records) from this one pass.
    const uint32_t tl = tm.tl(), TS = TeamT::kSize;
    const int goal_kind = rc.goal_kind;
    auto gdA = as_global(swap ? rc.distB : rc.distA);
    const double INF = __longlong_as_double(0x7FF0000000000000ll);

    // pass 1: raycast every neighbour, total cost through it (rrt.rs:124, 137-140).  A ray whose end pixels lie inside
    // the all-free window around the new node's pixel (clr) crosses free pixels only and is not walked.
    PORRT_T0();
    double bt = INF;
    int bj = 0x7FFFFFFF;
    uint32_t nvalid = 0, leq = 0;
    int j0 = -1;                // first candidate of this lane, register resident
    double cost0 = -1.0, tot0 = INF, dA0 = 0.0;
    uint32_t bi = 0, bjx = 0, clr_b = 0;
    if (rc.has_grid) {
        to_pixel(rc, px, py, bi, bjx);
        if (clr_known >= 0) clr_b = (uint32_t)clr_known;         // the caller fetched it beside its other loads
        else if (bi < rc.H && bjx < rc.W) clr_b = as_global(rc.clr)[bi * rc.W + bjx];
    }
    for (uint32_t a = tl; a < cnt; a += TS) {
        const int j = L.id(a);
        double ax, ay;
        L.xy(a, ax, ay);
        const double dA = L.dA(a, j);
        const double cost = sqrt(dist2(ax, ay, px, py));
        uint32_t ai = 0, aj = 0;
        if (rc.has_grid) to_pixel(rc, ax, ay, ai, aj);
        bool ok = true;
        if (rc.has_grid) ok = segment_in_clearance(ai, aj, bi, bjx, clr_b) || traversed_class_px(rc, grid, ai, aj, bi, bjx, &err) == CLS_FREE;
#if defined(PORRT_TIMING) && PORRT_TIMING == 4
        // (slot 2: rays walked, its count: hits -- of every sample, heavy or not; printed as walked / hits / 100)
        if (b >= (uint32_t)PORRT_TIMING_FROM && b < (uint32_t)PORRT_TIMING_UNTIL) {
            atomicAdd(&rc.cnt->tim[10], 1ull);
            if (rc.has_grid && !segment_in_clearance(ai, aj, bi, bjx, clr_b) && !(rc.sat != nullptr && ai < rc.H && aj < rc.W && bi < rc.H && bjx < rc.W && segment_box_free(rc, ai, aj, bi, bjx)))
                atomicAdd(&rc.cnt->tim[2], 1ull);
        }
#endif
        const double total = dA + cost;
        if (a == tl) { j0 = j; cost0 = ok ? cost : -1.0; tot0 = total; dA0 = dA; }
        else { L.set_val(a, ok ? cost : -1.0); L.set_dA(a, dA); }
        if (ok) {
            ++nvalid;
            // (leq: how many of this lane's valid candidates share its minimum -- the ties are counted here, not by a pass of their own)
            if (total < bt) { bt = total; bj = j; leq = 1u; }
            else if (total == bt) { ++leq; bj = j < bj ? j : bj; }
        }
    }
#if defined(PORRT_TIMING) && PORRT_TIMING == 4
    PORRT_TACC_S(rc, 1);
#elif defined(PORRT_TIMING)
    t__0 = wall_clock64();
#endif
    nvalid = tm.sum(nvalid);
    const double my_bt = bt;
    tm.argmin(bt, bj);
    PORRT_TACC_S(rc, 4);
    int best;
    double best_cost, dnew;
    bool deferred = false;
    if (nvalid == 0) {
        // rrt.rs:132-134: fall back to the nearest node, not collision-checked
        best = as_global(rc.q_nn)[q_off(rc, b) + k];
        if (best < 0) best = nearest_search();      // k_nn2 did not need it (the sample was not steered): searched now, by the whole team
        best_cost = sqrt(dist2(as_global(rc.nx)[best], as_global(rc.ny)[best], px, py));
        dnew = gdA[best] + best_cost;
    } else {
        // equal totals: the reference keeps the first in kd-tree pre-order (rrt.rs:143-145).  The kd structure is
        // built beside the steps and may lag: if a tied node is not in it yet, the choice is deferred (k_tie_fix) --
        // every tied parent gives the same dist_root, so nothing but parent[id] depends on it.
        uint32_t n_tie = 0;
        int tie_max = -1;
        auto each_tie = [&](auto &&f) {
            if (j0 >= 0 && cost0 >= 0.0 && tot0 == bt) f(j0);
            for (uint32_t a = tl + TS; a < cnt; a += TS) {
                const double cost = L.val(a);
                if (cost >= 0.0) {
                    const int j = L.id(a);
                    if (L.dA(a, j) + cost == bt) f(j);
                }
            }
        };
        n_tie = tm.sum(my_bt == bt ? leq : 0u);
        best = bj;
        if (n_tie > 1) {
            each_tie([&](int j) { tie_max = j > tie_max ? j : tie_max; });
            tie_max = tm.max_i(tie_max);
            uint32_t kd_done = 0, unused = 0;
            if (tl == 0) kd_done = __hip_atomic_load(&rc.cnt->kd_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            tm.bcast2(kd_done, unused);       // one answer for the whole team
            bool lazy_ok = false;
            if (rc.kd_lazy) {
                // Only the goal path is kept beside the steps (g_track_step).  Tied nodes off it that all lie at ONE place are a chain of
                // ancestors in the kd-tree (a point inserted again follows the path of its first copy and goes on below it): the lowest
                // id is first, as on the goal path.  (The goal-biased samples of a tree that has not reached its goal yet are steered to
                // the same point again and again: such clusters are the off-path ties of an ordinary run.)  Tied nodes at different
                // places off the path may need the whole structure: every tied node then counts as not yet known and the record waits
                // for the build after the steps.
                int off_min = kEmpty;
                each_tie([&](int j) { if (!(as_global(rc.kd_gexit)[j] & kOnG)) off_min = j < off_min ? j : off_min; });
                off_min = tm.min_i(off_min);
                uint32_t n_else = 0;
                if (off_min != kEmpty) {
                    const double ox = as_global(rc.nx)[off_min], oy = as_global(rc.ny)[off_min];
                    each_tie([&](int j) { if (!(as_global(rc.kd_gexit)[j] & kOnG) && (as_global(rc.nx)[j] != ox || as_global(rc.ny)[j] != oy)) ++n_else; });
                    n_else = tm.sum(n_else);
                }
                if (n_else) { kd_done = 0; if (tl == 0) { atomicAdd(&rc.cnt->n_lca, 1u); atomicMax(&rc.cnt->lca_next, b + 1u); } }
                else lazy_ok = true;
            }
            // first in pre-order among the tied nodes the kd structure already holds
            int on_min = kEmpty;        // tied nodes ON the goal path: an ancestor chain, the lowest id is first
            int off_best = kEmpty;      // pre-order-first tied node off the goal path
            uint32_t n_fresh = 0;
            each_tie([&](int j) {
                if ((uint32_t)j >= kd_done) { ++n_fresh; return; }
                if (as_global(rc.kd_gexit)[j] & kOnG) on_min = j < on_min ? j : on_min;
                else if (lazy_ok) off_best = j < off_best ? j : off_best;              // (one place: the lowest id)
                else if (off_best == kEmpty || kd_preorder_less(rc, j, off_best)) off_best = j;
            });
            on_min = tm.min_i(on_min);
            off_best = lazy_ok ? tm.min_i(off_best) : tm.first_preorder(rc, off_best);
            int known;
            if (off_best == kEmpty) known = on_min;
            else if (on_min == kEmpty) known = off_best;
            else known = kd_preorder_less(rc, on_min, off_best) ? on_min : off_best;
            const uint32_t my_fresh = n_fresh;
            n_fresh = (uint32_t)tie_max < kd_done ? 0u : tm.sum(n_fresh);
            if (n_fresh == 0) {
                best = known;
            } else {
                // record: the fresh tied ids (+ the best of the known ones) for k_tie_fix
                deferred = true;
                const uint32_t m = n_fresh + (known != kEmpty ? 1u : 0u);
                uint32_t rec = 0, base = 0;
                if (tl == 0) {
                    rec = atomicAdd(&rc.cnt->pend_cnt, 1u);
                    base = atomicAdd(&rc.cnt->pool_n, m);
                    if (rec < rc.pend_cap && base + m <= rc.pool_cap) {
                        rc.pend_new[rec] = (int)id; rc.pend_off[rec] = base; rc.pend_n[rec] = m;
                        rc.pend_cur[rec] = known != kEmpty ? 1u : 0u;
                        if (known != kEmpty) rc.pend_pool[base] = known;
                    } else {
                        err |= ERR_TIE_POOL;
                    }
                }
                tm.bcast2(rec, base);
                if (rec < rc.pend_cap && base + m <= rc.pool_cap) {
                    // the placeholder is in place before the record can be seen: k_tie_fix may settle a record the moment it is
                    // published (CAS of the placeholder), from another stream
                    if (tl == 0) as_global(rc.parent)[id] = kParentPending;
                    for (uint32_t c = tl; c < n_clone; c += TS) as_global(rc.parent)[clone_ids[c]] = kParentPending;
                    if (my_fresh) {
                        each_tie([&](int j) { if ((uint32_t)j >= kd_done) rc.pend_pool[base + atomicAdd(&rc.pend_cur[rec], 1u)] = j; });
                        __threadfence();
                    }
                    tm.sync();
                    if (tl == 0) {
                        __threadfence();
                        __hip_atomic_store(&rc.pend_state[rec], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    // the copies wait for the same tied nodes: records of their own over the same pooled ids
                    for (uint32_t c = tl; c < n_clone; c += TS) {
                        const uint32_t rc2 = atomicAdd(&rc.cnt->pend_cnt, 1u);
                        if (rc2 < rc.pend_cap) {
                            rc.pend_new[rc2] = (int)clone_ids[c]; rc.pend_off[rc2] = base; rc.pend_n[rc2] = m; rc.pend_cur[rc2] = m;
                            __threadfence();
                            __hip_atomic_store(&rc.pend_state[rc2], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            err |= ERR_TIE_POOL;
                        }
                    }
                }
            }
        }
        // cost and dist_root through the chosen parent: the lane that holds it as its first candidate has both
        const unsigned long long own = TeamT::kOneWave ? tm.ballot(j0 == best && cost0 >= 0.0) : 0ull;
        if (own) {
            const int src = (int)__builtin_ctzll(own);
            best_cost = tm.shfl(cost0, src);
            dnew = tm.shfl(tot0, src);
        } else {
            best_cost = sqrt(dist2(as_global(rc.nx)[best], as_global(rc.ny)[best], px, py));
            dnew = gdA[best] + best_cost;
        }
    }

    PORRT_TACC_S(rc, 5);
    // new node (rrt.rs:148, 30-37) and goal test (rrt.rs:165-167)
    // SquareGoal test (common.rs:336-345), lane g <-> goal g; the goal table sits in the run constants (scalar cache)
    bool fin = false;
    unsigned long long fmask = 0;
    if (goal_kind == 1) {
        for (uint32_t g0 = 0; g0 < rc.G && !fin; g0 += TeamT::kOneWave ? TeamT::kSize : 64u) {
            const uint32_t g = g0 + tm.sub();
            double goal_d = __longlong_as_double(0x7FF0000000000000ll);
            unsigned long long goal_m = 0;
            if (g < rc.G) {
                goal_d = fabs(rc.gcx[g] - px);
                goal_d += fabs(rc.gcy[g] - py);
                goal_m = rc.gmask[g];
            }
            const unsigned long long hits = tm.ballot(goal_d < rc.g_l1);      // first listed goal wins
            if (hits) { fin = true; fmask = tm.shfl(goal_m, (int)__builtin_ctzll(hits)); }
        }
    }
    if (tl == 0) {
        // everything that READS memory first (the memory counter is in order: a load's wait behind a store waits for
        // the store's round trip too), then nothing but stores
        if (goal_kind == 2) {
            GlobalGrid ggrid;
            ggrid.p = rc.cls; ggrid.W = rc.W;
            fin = goal_hit(rc, ggrid, px, py, fmask, &err);
        }
        int rep_at[kRepLevels];
#pragma unroll
        for (int l = 0; l < kRepLevels; ++l) {
            int cx, cy;
            const int G = rep_dim(l);
            rep_cell(rc, px, py, G, cx, cy);
            rep_at[l] = rep_off(l) + cy * G + cx;
        }
        auto g_rep = as_global(rc.rep);
        auto g_par = as_global(rc.parent);
        auto g_ff = as_global(rc.final_flag);
        auto g_fm = as_global(rc.final_mask);
        auto g_dB = as_global(rc.distB);
        auto g_nx = as_global(rc.nx), g_ny = as_global(rc.ny), g_dA = as_global(rc.distA);
        g_nx[id] = px;
        g_ny[id] = py;
        // the pyramid only ever bounds a search (any node named by a cell will do): its coarse levels are complete long before the
        // tree is, so only a young tree's nodes are entered there -- two scattered stores fewer per node
        g_rep[rep_at[0]] = (int)id;
        auto g_repf = as_global(rc.rep_f);
        const flt2 pf = flt2{(float)px, (float)py};
        g_repf[rep_at[0]] = pf;
        if (id < kRepCoarseUntil) {
#pragma unroll
            for (int l = 1; l < kRepLevels; ++l) g_rep[rep_at[l]] = (int)id;
            g_repf[rep_at[1]] = pf;
        }
        if (!deferred) g_par[id] = best;           // (a deferred parent: the placeholder was stored before its record was published)
        g_dA[id] = dnew;
        g_dB[id] = dnew;
        // final_flag is cleared when a grow starts (RRT*); the mask is only ever read where the flag is set
        if (fin) { g_ff[id] = 1; g_fm[id] = fmask; atomicAdd(&rc.cnt->n_final, 1u); }
    }
    if (n_clone) {
        if (goal_kind == 2) {                                            // the team's first lane made that test
            uint32_t f = fin ? 1u : 0u, unused = 0;
            tm.bcast2(f, unused);
            fin = f != 0u;
        }
        for (uint32_t c = tl; c < n_clone; c += TS) {
            const uint32_t ic = clone_ids[c];
            as_global(rc.nx)[ic] = px;
            as_global(rc.ny)[ic] = py;
            if (!deferred) as_global(rc.parent)[ic] = best;
            as_global(rc.distA)[ic] = dnew;
            as_global(rc.distB)[ic] = dnew;
            if (fin) {
                as_global(rc.final_flag)[ic] = 1;
                as_global(rc.final_mask)[ic] = goal_kind == 2 ? 1ull : fmask;
                atomicAdd(&rc.cnt->n_final, 1u);
            }
        }
    }
    PORRT_TACC_S(rc, 6);
    // rewire phase 1 (rrt.rs:152-161): dist_root candidates, min wins
    auto gdB = as_global(reinterpret_cast<unsigned long long *>(swap ? rc.distA : rc.distB));
    if constexpr (ListT::kCompact) {
        // only the actual candidates are kept for the commit pass, compacted (rounds of one candidate per lane)
        {
            uint32_t n_out = 0;
            for (uint32_t a0 = 0; a0 < cnt; a0 += TS) {
                const uint32_t a = a0 + tl;
                double via = -1.0;
                int j = -1;
                if (a < cnt) {
                    const double cost = a0 == 0 ? cost0 : L.val(a);
                    j = a0 == 0 ? j0 : L.id(a);
                    if (nvalid != 0 && cost >= 0.0 && j != best) {
                        const double v = dnew + cost;
                        if (v < (a0 == 0 ? dA0 : L.dA(a, j))) { g_atomic_min(gdB + j, f64_bits(v)); via = v; }
                    }
                }
                const unsigned long long hm = tm.ballot(via >= 0.0);
                const uint32_t pos = n_out + (uint32_t)__popcll(hm & ((1ull << tm.sub()) - 1ull));
                if (via >= 0.0) {
                    if (pos < L.out_cap) { L.out_id[pos] = j; L.out_val[pos] = via; }
                    else err |= ERR_CAND_OVERFLOW;
                }
                n_out += (uint32_t)__popcll(hm);
            }
            if (tl == 0) *L.out_cnt = n_out;
        }
        PORRT_TACC_S(rc, 7);
        return;
    } else {
    if (j0 >= 0) {
        double out = -1.0;                  // cand_val after this pass: the candidate dist_root of a rewire, or < 0
        if (nvalid != 0 && cost0 >= 0.0 && j0 != best) {
            const double via = dnew + cost0;
            if (via < dA0) {
                g_atomic_min(gdB + j0, f64_bits(via));
                out = via;
            }
        }
        L.set_val(tl, out);
    }
    for (uint32_t a = tl + TS; a < cnt; a += TS) {
        const double cost = L.val(a);
        const int j = L.id(a);
        double out = -1.0;
        if (nvalid != 0 && cost >= 0.0 && j != best) {
            const double via = dnew + cost;
            if (via < L.dA(a, j)) {
                g_atomic_min(gdB + j, f64_bits(via));
                out = via;
            }
        }
        L.set_val(a, out);
    }
    }
}

constexpr uint32_t kHeavyCand = 256;     // samples with more neighbours than this are served by the 4-wave team

// One workgroup = kConnectWaves samples.  Phase 1: every wave serves its own sample if it is light.  Phase 2:
// samples with more than kHeavyCand neighbours (the dense start of a tree, duplicates of the goal point) are
// served one after the other by the whole workgroup as a 4-wave team.
// (one workgroup's kConnectWaves samples; bx = the workgroup's index among the step's connect workgroups)
template <bool LDSGRID>
__device__ __forceinline__ void connect_block(const RunConst &rc, uint32_t b, uint32_t nb, uint32_t vwords, uint32_t bx, uint8_t *lds_tiles, double *s_d, int *s_i,
                                              uint32_t *s_heavy, uint32_t swap = 0) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t k = uni(bx * kConnectWaves + wv);
    const uint32_t qo = q_off(rc, b);
    const bool active = k < nb && as_global(rc.q_vid)[qo + (k < nb ? k : 0)] >= 0;
    const uint32_t cnt = active ? cand_count(rc, b, k) : 0u;
    const bool heavy = active && cnt > kHeavyCand;
    if (lane == 0) s_heavy[wv] = heavy ? cnt : 0u;
    const uint32_t TW = 2u * rc.tile_R + 1u;
    const uint32_t tile_bytes = (TW * TW + 15u) & ~15u;
    uint32_t err = 0;
    if (active && !heavy) {
        TileGrid grid;
        if (LDSGRID) {
            grid = load_tile(rc, lds_tiles + wv * tile_bytes, as_global(rc.q_x)[qo + k], as_global(rc.q_y)[qo + k], lane, 64u);
            __builtin_amdgcn_wave_barrier();
        } else {
            grid.lds = nullptr; grid.glob = rc.cls; grid.W = rc.W; grid.TW = 0; grid.oi = 0; grid.oj = 0;
        }
        Team<1> tm;
        tm.scr_d = nullptr; tm.scr_i = nullptr; tm.wave = 0; tm.lane = lane;
        const uint32_t id = uni(uni(as_global(rc.n_at)[b]) + rank_before(rc, b, vwords, k));            // k is wave-uniform: scalars
        const double px = uni_d(as_global(rc.q_x)[qo + k]), py = uni_d(as_global(rc.q_y)[qo + k]);
        GlobalList gl = global_list(rc, b, k);
        if (swap) gl.gdA = as_global((const double *)rc.distB);
        connect_rrt_sample(rc, tm, gl, grid, b, k, id, px, py, cnt, err, nullptr, 0, -1, NoNearestSearch(), swap);
    }
    __syncthreads();
    Team<kConnectWaves> tmh;
    tmh.scr_d = s_d; tmh.scr_i = s_i; tmh.wave = wv; tmh.lane = lane;
    for (uint32_t w = 0; w < kConnectWaves; ++w) {
        const uint32_t hc = s_heavy[w];            // workgroup-uniform
        if (!hc) continue;
        const uint32_t kh = bx * kConnectWaves + w;
        TileGrid grid;
        __syncthreads();
        if (LDSGRID) grid = load_tile(rc, lds_tiles, rc.q_x[qo + kh], rc.q_y[qo + kh], threadIdx.x, kConnectWaves * 64u);
        else { grid.lds = nullptr; grid.glob = rc.cls; grid.W = rc.W; grid.TW = 0; grid.oi = 0; grid.oj = 0; }
        __syncthreads();
        const uint32_t idh = uni(uni(as_global(rc.n_at)[b]) + rank_before(rc, b, vwords, kh));
        const double pxh = uni_d(as_global(rc.q_x)[qo + kh]), pyh = uni_d(as_global(rc.q_y)[qo + kh]);
        GlobalList glh = global_list(rc, b, kh);
        if (swap) glh.gdA = as_global((const double *)rc.distB);
        connect_rrt_sample(rc, tmh, glh, grid, b, kh, idh, pxh, pyh, hc, err, nullptr, 0, -1, NoNearestSearch(), swap);
    }
    if (err) atomicOr(&rc.cnt->err, err);
}

template <bool LDSGRID>
__global__ __launch_bounds__(kConnectWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_connect_rrt(const RunConst *__restrict__ rcp, uint32_t b, uint32_t nb,
                                                                     uint32_t vwords) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_tiles[];
    __shared__ double s_d[kConnectWaves];
    __shared__ int s_i[kConnectWaves];
    __shared__ uint32_t s_heavy[kConnectWaves];
    __shared__ __attribute__((aligned(16))) uint8_t s_ins[kInsertLds];
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row (porrt_grow_batch)
    nb = row_nb(rc, b, nb);
    if (blockIdx.x == gridDim.x - 1) { if (nb) insert_step_pages(rc, b, nb, vwords, s_ins); return; }    // the extra block
    if (blockIdx.x * kConnectWaves >= nb) return;
    connect_block<LDSGRID>(rc, b, nb, vwords, blockIdx.x, lds_tiles, s_d, s_i, s_heavy);
}

// Pipelined steps (RRT*, one wave per sample): step b is connected while step b + 1 is searched, in ONE launch -- the two
// touch disjoint data once the filing of a step's nodes has a kernel of its own (k_file_commit, between two of these):
//   k_near(0), F(0), S(0) = connect(0) + search(1), F(1) = file(1) + rewire phase 2 of (0), S(1) = connect(1) + search(2), ...
// The search of step b + 1 needs the pages, coordinates and counts of the nodes up to step b (F(b) wrote them: the
// connect pass's own stores of nx / ny carry the same bits), never dist_root or parents; its per-sample results go to the
// other half of q_* (q_stride) and the other parity of the neighbour lists.  A single query's chain of dependent kernels
// is then max(search, connect) + file per step instead of search + connect.  The connect workgroups come first (they
// take longer).
template <bool LDSGRID>
__global__ __launch_bounds__(kConnectWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_step_rrt(const RunConst *__restrict__ rcp, uint32_t b, uint32_t nb,
                                                                  uint32_t i0_next, uint32_t nb_next, uint32_t vwords) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_tiles[];
    __shared__ double s_d[kConnectWaves];
    __shared__ int s_i[kConnectWaves];
    __shared__ uint32_t s_heavy[kConnectWaves];
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row (porrt_grow_batch)
    const uint32_t cblocks = (nb + kConnectWaves - 1u) / kConnectWaves;
    if (blockIdx.x < cblocks) { connect_block<LDSGRID>(rc, b, nb, vwords, blockIdx.x, lds_tiles, s_d, s_i, s_heavy); return; }
    const uint32_t k = uni((blockIdx.x - cblocks) * 4u + (threadIdx.x >> 6));
    if (k < nb_next) near_sample<false>(rc, b + 1u, i0_next, vwords, k, threadIdx.x & 63u);
}

// ---- one kernel per step (single query, option pipeline = 4)
// The pipelined form above still has two kernels on a step's critical path, because two things wait for ALL of a phase: the filing
// of a step's nodes for the step's searches, and the next connect pass for the rewire commit.  Neither has to:
//   * the search of step b takes the nodes of steps < b - 1 from the pages and those of step b - 1 -- at most K, positions known
//     since that step's search -- from the step's own arrays (near_sample_lag), so the filing of step b - 1 runs BESIDE it;
//   * dist_root lives in two arrays that trade the roles of snapshot and rewire accumulator every step (connect_rrt_sample, swap),
//     so the rewire commit of step b - 1 runs BESIDE the connect pass of step b (commit_rrt_sample, lag).
// X(b) = connect(b) | search(b + 1) | file(b) | commit(b - 1): everything in it depends on X(b - 1) and nothing in it on anything else
// in it.  One launch gap per step instead of two.
// flat: the steered states of the step before, staged in LDS by the workgroup (one round trip for its four samples' two scans each)
__device__ __forceinline__ void near_sample_lag(const RunConst &rc, uint32_t b, uint32_t i0, uint32_t vwords, uint32_t k, uint32_t lane, uint32_t nb_prev,
                                                const dbl2 *flat) {
    const uint32_t pb = b - 1u;
    const uint32_t Np = uni(as_global(rc.n_at)[pb]);             // ids below it are in the pages (counts of parity pb & 1)
    // the step before: its valid samples become the ids Np, Np + 1, ... in sample order; lane t holds word t of the mask
    const unsigned long long word = lane < vwords ? as_global(rc.valid_mask)[(size_t)pb * vwords + lane] : 0ull;
    uint32_t incl = (uint32_t)__popcll(word);
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
    const uint32_t excl = incl - (uint32_t)__popcll(word);
    const uint32_t N = Np + uni(__shfl(incl, 63));
    const double sqx = uni_d(as_global(rc.sx)[i0 + k]), sqy = uni_d(as_global(rc.sy)[i0 + k]);
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    const double T2 = rc.rad_T2[N];                             // rrt.rs:121: the size before insertion
    double bestD = INF, bestx = 0.0, besty = 0.0, thr = INF;
    int best = 0x7FFFFFFF;
    auto take = [&](double x, double y, int id) {
        const double d2 = dist2(x, y, sqx, sqy);
        if (d2 > thr) return;
        const double D = sqrt(d2);                               // the reference compares rounded distances
        if (D < bestD || (D == bestD && id < best)) { bestD = D; best = id; bestx = x; besty = y; thr = d2 * (1.0 + 1e-15); }
    };
    auto wave_best = [&]() {
        double rd = bestD;
        int ri = best;
        for (int off = 32; off > 0; off >>= 1) {
            const double od = __shfl_xor(rd, off);
            const int oi = __shfl_xor(ri, off);
            if (od < rd || (od == rd && oi < ri)) { rd = od; ri = oi; }
        }
        const unsigned long long own = __ballot(best == ri && bestD == rd);
        const int src = own ? (int)__builtin_ctzll(own) : 0;
        thr = __shfl(thr, src); bestx = __shfl(bestx, src); besty = __shfl(besty, src);
        bestD = rd; best = ri;
    };
    // the step before, from its arrays (all loads of a round are independent)
    for (uint32_t t = 0; t * 64u < nb_prev; ++t) {
        const unsigned long long w = __shfl(word, (int)t);
        const uint32_t pre = __shfl(excl, (int)t);
        if ((w >> lane) & 1ull) { const dbl2 v = flat[t * 64u + lane]; take(v.x, v.y, (int)(Np + pre + (uint32_t)__popcll(w & ((1ull << lane) - 1ull)))); }
    }
    wave_best();
    {
        auto visit = [&](double x, double y, int id, bool ok) { if (ok) take(x, y, id); };
        const uint32_t own = uni(region_of(rc, sqx, sqy));
        scan_disc(rc, pb, sqx, sqy, 0.0, Np, lane, visit);
        wave_best();
        double m2;
        if (best != 0x7FFFFFFF) m2 = thr;
        else { m2 = nn_bound_wave<false>(rc, Np, sqx, sqy, 0u, lane); thr = m2 * (1.0 + 1e-9); }
        scan_disc(rc, pb, sqx, sqy, disc_radius(m2, sqx, sqy), Np, lane, visit, own);
        wave_best();
    }
    const int nn = best == 0x7FFFFFFF ? 0 : best;
    const double fx = best == 0x7FFFFFFF ? as_global(rc.nx)[0] : bestx, fy = best == 0x7FFFFFFF ? as_global(rc.ny)[0] : besty;
    double tx = sqx, ty = sqy;
    double step = fabs(tx - fx);                                  // common.rs:215-225
    step += fabs(ty - fy);
    if (step > rc.max_step) {
        const double lambda = rc.max_step / step;
        double ux = (tx - fx) * lambda, uy = (ty - fy) * lambda;
        tx = fx + ux;
        ty = fy + uy;
    }
    uint32_t err = 0;
    bool valid = true;
    if (rc.has_grid) {
        const int cls = state_class(rc, tx, ty, &err);
        valid = cls == CLS_FREE && !err;                           // RTTFuncs adapter (tamp_rrt.rs:40-42)
    }
    if (lane == 0) {
        const uint32_t qo = q_off(rc, b);
        as_global(rc.q_x)[qo + k] = tx;
        as_global(rc.q_y)[qo + k] = ty;
        const size_t o2 = (size_t)b * rc.part_stride + k;
        as_global(rc.kq_x)[o2] = tx; as_global(rc.kq_y)[o2] = ty; as_global(rc.kq_vid)[o2] = valid ? 0 : -1;
        as_global(rc.q_nn)[qo + k] = nn;
        as_global(rc.q_vid)[qo + k] = valid ? 0 : -1;
        if (valid) atomicOr(&rc.valid_mask[(size_t)b * vwords + (k >> 6)], 1ull << (k & 63u));
        if (err) atomicOr(&rc.cnt->err, err);
    }
    if (!valid) return;
    auto cid = as_global(rc.cand_id) + cand_off(rc, b, k);
    auto cxy = as_global(reinterpret_cast<dbl2 *>(rc.cand_xy)) + cand_off(rc, b, k);
    const uint32_t cap = rc.cand_cap;
    uint32_t tot = 0;
    bool over = false;
    auto hit = [&](double x, double y, int id, bool in) {
        const unsigned long long hm = __ballot(in);
        const uint32_t pos = tot + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
        if (in) {
            if (pos < cap) { cid[pos] = id; dbl2 v; v.x = x; v.y = y; cxy[pos] = v; }
            else over = true;
        }
        tot += (uint32_t)__popcll(hm);
    };
    scan_disc(rc, pb, tx, ty, disc_radius(T2, tx, ty), Np, lane, [&](double x, double y, int id, bool ok) { hit(x, y, id, ok && dist2(x, y, tx, ty) <= T2); });
    for (uint32_t t = 0; t * 64u < nb_prev; ++t) {
        const unsigned long long w = __shfl(word, (int)t);
        const uint32_t pre = __shfl(excl, (int)t);
        const uint32_t kk = t * 64u + lane;
        const dbl2 v = kk < nb_prev ? flat[kk] : dbl2{0.0, 0.0};
        const bool in = ((w >> lane) & 1ull) && dist2(v.x, v.y, tx, ty) <= T2;
        hit(v.x, v.y, (int)(Np + pre + (uint32_t)__popcll(w & ((1ull << lane) - 1ull))), in);
    }
    if (lane == 0) as_global(rc.cand_cnt)[cand_cnt_at(rc, b, k)] = tot;
    if (over) atomicOr(&rc.cnt->err, (uint32_t)ERR_CAND_OVERFLOW);
}

// X(b): workgroups [file(b)] [connect(b)] [search(b + 1)] [commit(cb): cnb samples, cb = b - 1 or none]
// The cell of the goal point (px, py) after one more level of its path: the level's node (wx, wy) at depth `depth` sends it left or right
// (nearest_neighbor.rs:32: `<` goes left, equal goes right), and a node follows the path through the level iff it lies on the same side.
__device__ __forceinline__ KdBox g_box_after(const KdBox &prev, double wx, double wy, uint32_t depth, double px, double py) {
    KdBox b = prev;
    if (depth & 1u) { if (py < wy) b.hiy = wy < b.hiy ? wy : b.hiy; else b.loy = wy > b.loy ? wy : b.loy; }
    else { if (px < wx) b.hix = wx < b.hix ? wx : b.hix; else b.lox = wx > b.lox ? wx : b.lox; }
    return b;
}
__device__ __forceinline__ KdBox g_box_all() {
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    KdBox b;
    b.lox = -INF; b.hix = INF; b.loy = -INF; b.hiy = INF;
    return b;
}
// (TAG: a copy of the function per kind of caller -- a function called from a kernel is compiled to the register budget that kernel ASKS for,
// and the single query's step kernels ask for eight waves per SIMD: 0 = those, 1 = the kernels that leave it 128 registers)
template <int TAG> __device__ void g_track_step(const RunConst &rc, uint32_t b, uint32_t nb, uint32_t vwords, uint8_t *lds);
constexpr uint32_t kGTrackNd = 168u;                                         // levels of the goal path whose cells g_track_step stages in LDS (36 B each)
constexpr uint32_t kGTrackLds0 = 8192u + 512u + 272u + kGTrackNd * 36u, kGTrackLds = kGTrackLds0 + 256u * 16u + 160u * 16u;        // bytes of LDS g_track_step needs
static_assert(kGTrackLds <= kFileLds, "g_track_step uses the filing scratch of k_step1_rrt");
// (what k_step1_rrt asks for: the kernel itself never fitted the 64 registers of eight waves per SIMD -- it takes 83 -- but a function it calls
// is compiled to the budget it ASKS for, and the goal path's workgroup spilled in 64; at five waves the function gets 88 registers and a
// single query takes 3.94 instead of 4.23 ms)
#ifndef PORRT_STEP1_WAVES
#define PORRT_STEP1_WAVES 5
#endif
template <bool LDSGRID>
__global__ __launch_bounds__(kConnectWaves * 64) __attribute__((amdgpu_waves_per_eu(PORRT_STEP1_WAVES, 8))) void k_step1_rrt(const RunConst *__restrict__ rcp, uint32_t b, uint32_t nb,
                                                                   uint32_t i0_next, uint32_t nb_next, uint32_t vwords, uint32_t cb, uint32_t cnb, uint32_t lazy) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_tiles[];
    __shared__ double s_d[kConnectWaves];
    __shared__ int s_i[kConnectWaves];
    __shared__ uint32_t s_heavy[kConnectWaves];
    __shared__ __attribute__((aligned(16))) uint8_t s_ins[kFileLds];
    const RunConst &rc = rcp[blockIdx.y];
    // in launch order: the filing (one workgroup's chain of phases: it must start first), the connect pass (the longest waves),
    // the searches, the rewire commit (short)
    const uint32_t cblocks = (nb + kConnectWaves - 1u) / kConnectWaves, sblocks = (nb_next + 3u) / 4u;
    uint32_t bx = blockIdx.x;
    if (bx == 0) { file_step_fast<kConnectWaves * 64u>(rc, b, nb, vwords, s_ins); return; }
    bx -= 1u;
    if (lazy) {                                  // the goal path of the kd order (kd_lazy): one workgroup, like the filing
        if (bx == 0) { g_track_step<0>(rc, b, nb, vwords, s_ins); return; }
        bx -= 1u;
    }
    if (bx < cblocks) { connect_block<LDSGRID>(rc, b, nb, vwords, bx, lds_tiles, s_d, s_i, s_heavy, b & 1u); return; }
    bx -= cblocks;
    if (bx < sblocks) {
        // the step's own steered states, once per workgroup (the filing scratch is free in a search workgroup; nb <= 1024)
        dbl2 *flat = reinterpret_cast<dbl2 *>(s_ins);
        {
            const uint32_t qp = q_off(rc, b);
            auto pqx = as_global(rc.q_x) + qp, pqy = as_global(rc.q_y) + qp;
            double fx[4], fy[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t kk = u * 256u + threadIdx.x;
                fx[u] = kk < nb ? pqx[kk] : 0.0;
                fy[u] = kk < nb ? pqy[kk] : 0.0;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) { dbl2 v; v.x = fx[u]; v.y = fy[u]; flat[u * 256u + threadIdx.x] = v; }
        }
        __syncthreads();
        const uint32_t k = uni(bx * 4u + (threadIdx.x >> 6));
        if (k < nb_next) near_sample_lag(rc, b + 1u, i0_next, vwords, k, threadIdx.x & 63u, nb, flat);
        return;
    }
    bx -= sblocks;
    const uint32_t ck = uni(bx * 4u + (threadIdx.x >> 6));
    if (ck < cnb) commit_rrt_sample(rc, cb, vwords, ck, threadIdx.x & 63u, 64u, true);
}

// F(bf): block 0 files step bf's nodes (positions and validity are final since its search), the others run the rewire phase
// 2 of step cb (cnb samples; cnb = 0: none)
// (1024 threads: the filing is one workgroup's chain of dependent phases, a thread per sample keeps each phase to one pass)
__global__ __launch_bounds__(1024) void k_file_commit(const RunConst *__restrict__ rcp, uint32_t bf, uint32_t nbf, uint32_t cb, uint32_t cnb, uint32_t vwords) {
    __shared__ __attribute__((aligned(16))) uint8_t s_ins[kFileLds];
    const RunConst &rc = rcp[blockIdx.y];
    if (blockIdx.x == 0) { file_step_fast<1024u>(rc, bf, nbf, vwords, s_ins); return; }
    const uint32_t ck = uni((blockIdx.x - 1u) * 16u + (threadIdx.x >> 6));
    if (ck < cnb) commit_rrt_sample(rc, cb, vwords, ck, threadIdx.x & 63u);
}

// ---- the persistent step loop of a single query (option pipeline = 2)
// All steps of a grow in ONE launch: the same phases as k_step_rrt / k_file_commit, separated by a barrier over the grid instead
// of a kernel boundary (two launch gaps per step are most of what a single query's 110-step chain has left to lose).  The grid is
// small enough to be resident at once -- the launch is a cooperative one, which refuses instead of deadlocking when it is not --
// and every workgroup takes the phase's work items in turn.  Per step b:
//     S(b):  connect(b) in the first items, search(b + 1) in the rest          | barrier
//     F(b):  workgroup 0 files step b + 1's nodes, the others run step b's rewire phase 2     | barrier, coop_filed = b + 2
// The kd structure (tie order) is built beside it on the side stream, by kernels launched ahead that wait for coop_filed.
// The barrier: arrivals are counted in one monotone counter; whoever waits longer than two seconds gives up for everybody.
__device__ __forceinline__ bool coop_grid_sync(const RunConst &rc, uint32_t nblocks, uint32_t &phase) {
    __shared__ uint32_t s_go;
    __syncthreads();
    if (threadIdx.x == 0) {
        ++phase;
        const uint32_t target = phase * nblocks;
        __threadfence();
        __hip_atomic_fetch_add(&rc.cnt->coop_bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        uint32_t go = 1u;
        while (__hip_atomic_load(&rc.cnt->coop_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {      // (relaxed: an acquire per poll would flush the CU's L1 under the workgroups still at work)
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > 200000000ull || __hip_atomic_load(&rc.cnt->coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                atomicOr(&rc.cnt->coop_abort, 1u); go = 0u; break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_go = go;
    }
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();            // scalar loads of what other workgroups wrote before the barrier
    return s_go != 0u;
}

template <bool LDSGRID>
__global__ __launch_bounds__(kConnectWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_coop_rrt(const RunConst *__restrict__ rcp, uint32_t n_steps, uint32_t K,
                                                                  uint32_t n_iter, uint32_t vwords) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_tiles[];
    __shared__ double s_d[kConnectWaves];
    __shared__ int s_i[kConnectWaves];
    __shared__ uint32_t s_heavy[kConnectWaves];
    __shared__ __attribute__((aligned(16))) uint8_t s_ins[kFileLds];
    const RunConst &rc = rcp[0];
    const uint32_t bid = blockIdx.x, nblk = gridDim.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t phase = 0;
    auto nb_of = [&](uint32_t b) { const uint32_t i0 = b * K; return b < n_steps ? (n_iter - i0 < K ? n_iter - i0 : K) : 0u; };
    // search(0), file(0)
    {
        const uint32_t nb0 = nb_of(0);
        for (uint32_t it = bid; it * 4u < nb0; it += nblk) {
            const uint32_t k = uni(it * 4u + wv);
            if (k < nb0) near_sample<false>(rc, 0u, 0u, vwords, k, lane);
        }
        if (!coop_grid_sync(rc, nblk, phase)) return;
        if (bid == 0) file_step_fast<kConnectWaves * 64u>(rc, 0u, nb0, vwords, s_ins);
        if (!coop_grid_sync(rc, nblk, phase)) return;
        if (bid == 0 && threadIdx.x == 0) __hip_atomic_store(&rc.cnt->coop_filed, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (uint32_t b = 0; b < n_steps; ++b) {
        const uint32_t nb = nb_of(b), nbn = nb_of(b + 1u), i0n = (b + 1u) * K;
        const uint32_t cb4 = (nb + kConnectWaves - 1u) / kConnectWaves, items = cb4 + (nbn + 3u) / 4u;
        for (uint32_t it = bid; it < items; it += nblk) {
            if (it < cb4) {
                connect_block<LDSGRID>(rc, b, nb, vwords, it, lds_tiles, s_d, s_i, s_heavy);
                __syncthreads();                   // (the tiles and the team scratch are reused by the workgroup's next item)
            } else {
                const uint32_t k = uni((it - cb4) * 4u + wv);
                if (k < nbn) near_sample<false>(rc, b + 1u, i0n, vwords, k, lane);
            }
        }
        if (!coop_grid_sync(rc, nblk, phase)) return;
        // F: the filing of step b + 1 by the first workgroup, step b's rewire phase 2 by the others (by all when nothing is filed)
        const uint32_t first_commit = nbn ? 1u : 0u;
        if (nbn && bid == 0) file_step_fast<kConnectWaves * 64u>(rc, b + 1u, nbn, vwords, s_ins);
        if (bid >= first_commit) {
            for (uint32_t it = bid - first_commit; it * 4u < nb; it += nblk - first_commit) {
                const uint32_t ck = uni(it * 4u + wv);
                if (ck < nb) commit_rrt_sample(rc, b, vwords, ck, lane);
            }
        }
        if (!coop_grid_sync(rc, nblk, phase)) return;
        if (bid == 0 && threadIdx.x == 0) __hip_atomic_store(&rc.cnt->coop_filed, b + 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// RRT*: rewire phase 2 for sample k of step b (one wave).  A pair wins iff its candidate equals the accumulated
// minimum; among equal candidates the lowest new id wins (sequential order of the reference, strict `<`, rrt.rs:157).
// lag (the one-kernel-per-step form): the step's accumulator and its snapshot are distB / distA on even steps and the other way
// round on odd ones, the commit runs beside the next step's connect pass -- which reads the accumulator as ITS snapshot and
// accumulates into the other array -- and carries the winners' values over with an atomic minimum (commutes with that pass).
__device__ void commit_rrt_sample(const RunConst &rc, uint32_t b, uint32_t vwords, uint32_t k, uint32_t lane, uint32_t stride, bool lag) {
    // everything the sample's own first trip to memory can fetch is asked for before the first thing is looked at (the valid bit
    // decides whether there is anything to do, and a test in front of the other loads is a trip of its own)
    const unsigned long long vword = as_global(rc.valid_mask)[(size_t)b * vwords + (k >> 6)];
    const uint32_t N = as_global(rc.n_at)[b];
    const uint32_t rank = stride < 64u ? rank_before_lanes(rc, b, vwords, k, lane, stride) : rank_before(rc, b, vwords, k);
    const uint32_t cnt = cand_count(rc, b, k);
    auto cid = as_global(rc.cand_id) + cand_off(rc, b, k);
    auto cval = as_global(rc.cand_val) + cand_val_off(rc, b, k);
    if (!((vword >> (k & 63u)) & 1ull)) return;
    const int id = (int)(N + rank);
    const bool odd = lag && (b & 1u);
    auto gdB = as_global(odd ? rc.distA : rc.distB);                                       // what the step's connect pass accumulated into
    auto gother = as_global(reinterpret_cast<unsigned long long *>(odd ? rc.distB : rc.distA));
    auto gpd = as_global(rc.pg_d);
    auto gso = as_global(rc.slot_of);
    // the new node's own dist_root goes to its page slot (filed by insert_step_pages beside the connect pass)
    if (lane == 0) gpd[gso[id]] = as_global(rc.distA)[id];
    // four candidates per lane in flight: the list of a sample in a dense neighbourhood has hundreds, and each is a chain of
    // dependent loads (candidate -> distB[j] -> parent[j])
    constexpr int U = 4;
    for (uint32_t a0 = lane; a0 < cnt; a0 += stride * U) {
        double via[U], dB[U];
        int j[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t a = a0 + (uint32_t)u * stride;
            via[u] = a < cnt ? cval[a] : -1.0;
            j[u] = a < cnt ? cid[a] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) dB[u] = via[u] >= 0.0 ? gdB[j[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!(via[u] >= 0.0) || f64_bits(via[u]) != f64_bits(dB[u])) continue;
            auto gpar = as_global(rc.parent) + j[u];
            int old = *gpar;
            while (old < (int)N || id < old) {      // parents from before this step are always < N
                int expect = old;
                if (__hip_atomic_compare_exchange_strong(gpar, &expect, id, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                old = expect;
            }
            if (lag) g_atomic_min(gother + j[u], f64_bits(via[u]));
            else as_global(rc.distA)[j[u]] = via[u];
            gpd[gso[j[u]]] = via[u];
        }
    }
}

// stand-alone form (last step of a launch sequence)
__global__ __launch_bounds__(256) void k_commit_rrt(const RunConst *__restrict__ rcp, uint32_t b, uint32_t nb, uint32_t vwords, uint32_t lag) {
    const uint32_t k = uni((blockIdx.x * 256u + threadIdx.x) >> 6);
    if (k < nb && k < row_nb(rcp[blockIdx.y], b, nb)) commit_rrt_sample(rcp[blockIdx.y], b, vwords, k, threadIdx.x & 63u, 64u, lag != 0u);
}

// ---- kd_lazy: the tie order without the kd-tree.
// kd_preorder_less settles every pair of tied nodes from where their root paths leave the goal path G (kd_gexit) -- except two
// nodes off G that leave it at the same node.  And where a new node leaves G depends on G alone: the levels that are not copies
// of the goal point are real tests (g_nd*), the copies reduce to two comparisons (k_kd_locate's long way).  So beside the steps
// only G is kept: one workgroup per row and step (it rides in k_conn2) gives every new node its exit level, and lets the nodes
// that stay on G to its end extend it, in id order.  A tie between two nodes off G is left deferred and counted (n_lca); the host
// then builds the whole structure after the steps, as with option kd_after, and k_tie_fix settles those records.  Equal costs
// through different parents off G take coordinates made to collide; the copies of the goal point, the ties of every run that
// reaches its goal, are all on G.
template <int TAG> __device__ void g_track_step(const RunConst &rc, uint32_t b, uint32_t nb, uint32_t vwords, uint8_t *lds) {
#ifdef PORRT_GTRACK_TIMING
    const unsigned long long gt0 = wall_clock64();
#define GT_MARK(slot) do { if (threadIdx.x == 0) { atomicAdd(&rc.cnt->tim[slot], wall_clock64() - gt0); atomicAdd(&rc.cnt->tim[(slot) + 8], 1ull); } } while (0)
#else
#define GT_MARK(slot) do {} while (0)
#endif
    constexpr uint32_t kNd = kGTrackNd;                                      // non-duplicate levels whose cells are staged in LDS (more: read from memory -- a trip per probe)
    uint16_t *s_k = reinterpret_cast<uint16_t *>(lds);                       // [4096] sample of the t-th new node
    unsigned long long *s_cand = reinterpret_cast<unsigned long long *>(lds + 8192);      // [64] new nodes on G to its end
    uint32_t *s_wpre = reinterpret_cast<uint32_t *>(lds + 8192 + 512);       // [65] valid samples before each mask word
    double *s_blx = reinterpret_cast<double *>(lds + 8192 + 512 + 272);      // [kNd] the goal point's cell after each level: lo x, hi x, lo y, hi y
    double *s_bhx = s_blx + kNd, *s_bly = s_bhx + kNd, *s_bhy = s_bly + kNd;
    uint32_t *s_ndi = reinterpret_cast<uint32_t *>(s_bhy + kNd);             // [kNd] the level's depth
    constexpr uint32_t kApp = 256, kCand = 160;
    double *s_ax = reinterpret_cast<double *>(lds + kGTrackLds0), *s_ay = s_ax + kApp;      // the levels this step adds to G
    double *s_cx = s_ay + kApp, *s_cy = s_cx + kCand;                         // the nodes on G to its end, in id order
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    auto vm = as_global(rc.valid_mask) + (size_t)b * vwords;
    const uint32_t N = as_global(rc.n_at)[b];
    const uint32_t glen0 = rc.cnt->g_len, n_nd = rc.cnt->g_nd_len, d0 = rc.cnt->g_first_dup[0], d1 = rc.cnt->g_first_dup[1];
    if (tid < 64u) s_cand[tid] = 0ull;
    if (tid < 64u) {                             // valid samples before each word of the mask: a word per lane, a prefix sum over the wave
        const uint32_t c = tid < vwords ? (uint32_t)__popcll(vm[tid]) : 0u;
        uint32_t inc = c;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, off); if ((int)lane >= off) inc += o; }
        if (tid < vwords) s_wpre[tid] = inc - c;
        if (tid == 63u) s_wpre[vwords] = inc;
    }
    for (uint32_t t2 = tid; t2 < n_nd && t2 < kNd; t2 += blockDim.x) {
        const KdBox bx = rc.g_nd_box[t2];
        s_ndi[t2] = as_global(rc.g_nd)[t2]; s_blx[t2] = bx.lox; s_bhx[t2] = bx.hix; s_bly[t2] = bx.loy; s_bhy[t2] = bx.hiy;
    }
    __syncthreads();
    const uint32_t n_new = s_wpre[vwords];
    for (uint32_t k = tid; k < nb; k += blockDim.x) {
        const unsigned long long w = vm[k >> 6];
        if ((w >> (k & 63u)) & 1ull) s_k[s_wpre[k >> 6] + (uint32_t)__popcll(w & ((1ull << (k & 63u)) - 1ull))] = (uint16_t)k;
    }
    __syncthreads();
    const double px = rc.gp_x, py = rc.gp_y;
    auto gex = as_global(rc.kd_gexit);
    const uint32_t qo = q_off(rc, b);
    GT_MARK(0);
    // every new node against G as it stands before the step: four nodes per thread, their coordinates fetched together.  A node follows G
    // through a level iff it lies in the goal point's cell after that level, and the cells are nested: the first level it fails is found by
    // bisection over the levels' cells (the first kNd of them in LDS) -- a node by the goal point used to walk a hundred levels one by one.
    for (uint32_t t0 = tid; t0 < ((n_new + blockDim.x - 1u) / blockDim.x) * blockDim.x; t0 += 4u * blockDim.x) {
        double cvx[4], cvy[4];
        uint32_t cE[4], lo[4], hi[4];
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t t = t0 + u * blockDim.x;
            const uint32_t k = t < n_new ? s_k[t] : 0u;
            cvx[u] = t < n_new ? as_global(rc.q_x)[qo + k] : 0.0;
            cvy[u] = t < n_new ? as_global(rc.q_y)[qo + k] : 0.0;
            cE[u] = 0xFFFFFFFFu;
            const bool open = t < n_new && !(cvx[u] == px && cvy[u] == py);   // (a copy of the goal point passes every level)
            lo[u] = 0u; hi[u] = open ? n_nd : 0u;                             // the first failed level lies in [lo, hi]; hi == n_nd: none fails
        }
        for (;;) {
            bool more = false;
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) more = more || lo[u] < hi[u];
            if (!more) break;
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                if (lo[u] >= hi[u]) continue;
                const uint32_t mid = (lo[u] + hi[u]) >> 1;
                double blx, bhx, bly, bhy;
                if (mid < kNd) { blx = s_blx[mid]; bhx = s_bhx[mid]; bly = s_bly[mid]; bhy = s_bhy[mid]; }
                else { const KdBox bx = rc.g_nd_box[mid]; blx = bx.lox; bhx = bx.hix; bly = bx.loy; bhy = bx.hiy; }
                const bool in = !(cvx[u] < blx) && cvx[u] < bhx && !(cvy[u] < bly) && cvy[u] < bhy;
                if (in) lo[u] = mid + 1u; else hi[u] = mid;
            }
        }
        GT_MARK(4);
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t t = t0 + u * blockDim.x;
            if (t < n_new && !(cvx[u] == px && cvy[u] == py) && lo[u] < n_nd) cE[u] = lo[u] < kNd ? s_ndi[lo[u]] : as_global(rc.g_nd)[lo[u]];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {
            const uint32_t t = t0 + u * blockDim.x;
            if (t >= n_new) continue;
            uint32_t E = cE[u];
            if (cvx[u] < px && d0 < E) E = d0;
            if (cvy[u] < py && d1 < E) E = d1;
            if (E != 0xFFFFFFFFu) gex[N + t] = E;
            else atomicOr(&s_cand[t >> 6], 1ull << (t & 63u));
        }
    }
    __syncthreads();
    GT_MARK(5);
    // the coordinates of the nodes that stay on G to its end, in id order (the first kCand of them), for the wave below
    if (tid < 64u) {
        const uint32_t c = (uint32_t)__popcll(s_cand[tid]);
        uint32_t inc = c;
        for (int off = 1; off < 64; off <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, off); if ((int)lane >= off) inc += o; }
        s_wpre[tid] = inc - c;                                               // (the sample prefix is no longer needed)
    }
    __syncthreads();
    for (uint32_t t = tid; t < n_new; t += blockDim.x) {
        const unsigned long long w = s_cand[t >> 6];
        if (!((w >> (t & 63u)) & 1ull)) continue;
        const uint32_t rank = s_wpre[t >> 6] + (uint32_t)__popcll(w & ((1ull << (t & 63u)) - 1ull));
        if (rank < kCand) { const uint32_t k = s_k[t]; s_cx[rank] = as_global(rc.q_x)[qo + k]; s_cy[rank] = as_global(rc.q_y)[qo + k]; }
    }
    __syncthreads();
    GT_MARK(1);
    // the nodes that follow G to its end, in id order: the first extends it, the next ones are tested against the levels added
    // before them and extend it in their turn if they pass them all.  One wave; their coordinates are in LDS (the first kCand of
    // them), the levels this step adds are kept there too (the first kApp): in an ordinary step -- a handful of copies of the goal
    // point -- nothing in the loop waits for memory.
    if (tid < 64u) {
        uint32_t len = glen0, nd_len = n_nd, fd0 = d0, fd1 = d1;
        KdBox cell = n_nd ? rc.g_nd_box[n_nd - 1u] : g_box_all();  // the goal point's cell at the end of G
        auto gx = as_global(reinterpret_cast<unsigned long long *>(rc.g_x)), gy = as_global(reinterpret_cast<unsigned long long *>(rc.g_y));
        uint32_t c_at = 0;
        for (uint32_t w = 0; w < (n_new + 63u) / 64u; ++w) {
            for (unsigned long long m = s_cand[w]; m;) {
                const uint32_t t = w * 64u + (uint32_t)__builtin_ctzll(m);
                m &= m - 1ull;
                // its coordinates: from the table (or, past the table's end, from memory)
                double vx, vy;
                if (c_at < kCand) { vx = s_cx[c_at]; vy = s_cy[c_at]; }
                else { const uint32_t k = s_k[t]; vx = as_global(rc.q_x)[qo + k]; vy = as_global(rc.q_y)[qo + k]; }
                ++c_at;
                uint32_t E = 0xFFFFFFFFu;
                if (!(vx == px && vy == py)) {                               // (a copy of the goal point passes every level)
                    for (uint32_t l0 = glen0; l0 < len && E == 0xFFFFFFFFu; l0 += 64u) {
                        const uint32_t lvl = l0 + lane;
                        bool out = false;
                        if (lvl < len) {
                            double wx, wy;
                            if (lvl - glen0 < kApp) { wx = s_ax[lvl - glen0]; wy = s_ay[lvl - glen0]; }
                            else {
                                __builtin_amdgcn_s_waitcnt(0);               // (more levels in one step than LDS keeps: from memory, once the stores are through)
                                wx = __longlong_as_double((long long)__hip_atomic_load(gx + lvl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                wy = __longlong_as_double((long long)__hip_atomic_load(gy + lvl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                            }
                            out = kd_left(vx, vy, wx, wy, lvl) != kd_left(px, py, wx, wy, lvl);
                        }
                        const unsigned long long ob = __ballot(out);
                        if (ob) E = l0 + (uint32_t)__builtin_ctzll(ob);
                    }
                }
                if (lane == 0) {
                    if (E != 0xFFFFFFFFu) {
                        gex[N + t] = E;
                    } else {
                        if (len + 8u < rc.g_cap) {
                            as_global(rc.g_id)[len] = (int)(N + t);
                            __hip_atomic_store(gx + len, (unsigned long long)__double_as_longlong(vx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(gy + len, (unsigned long long)__double_as_longlong(vy), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            atomicOr(&rc.cnt->err, (uint32_t)ERR_GPATH_OVERFLOW);
                        }
                        if (len - glen0 < kApp) { s_ax[len - glen0] = vx; s_ay[len - glen0] = vy; }
                        gex[N + t] = len | kOnG;
                        if (!(vx == px && vy == py)) {
                            as_global(rc.g_nd)[nd_len] = len; as_global(rc.g_nd_x)[nd_len] = vx; as_global(rc.g_nd_y)[nd_len] = vy;
                            cell = g_box_after(cell, vx, vy, len, px, py);
                            rc.g_nd_box[nd_len] = cell;
                        }
                    }
                }
                if (E == 0xFFFFFFFFu) {
                    if (vx == px && vy == py) { if (len & 1u) fd1 = len < fd1 ? len : fd1; else fd0 = len < fd0 ? len : fd0; }
                    else ++nd_len;
                    ++len;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      // (the level is in LDS before the next node reads it)
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
            }
        }
        GT_MARK(2);
        if (lane == 0) {
            rc.cnt->g_len = len; rc.cnt->g_nd_len = nd_len; rc.cnt->g_first_dup[0] = fd0; rc.cnt->g_first_dup[1] = fd1;
            __threadfence();
            __hip_atomic_store(&rc.cnt->kd_done, N + n_new, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        GT_MARK(3);
    }
}

// The goal path's workgroup as a kernel of its own: a single query launches it on its side stream beside the step kernel (step b's new
// nodes are known since the step kernel before; the step kernel after waits for it), one workgroup per row.
__global__ __launch_bounds__(256) void k_gtrack(const RunConst *__restrict__ rcp, uint32_t b, uint32_t nb, uint32_t vwords) {
    __shared__ __attribute__((aligned(16))) uint8_t s_lds[kGTrackLds];
    const RunConst &rc = rcp[blockIdx.y];
    nb = row_nb(rc, b, nb);
    if (nb == 0) return;
    g_track_step<1>(rc, b, nb, vwords, s_lds);
}

// the kd state of a grow's start (k_init_root), for the full build after lazily tracked steps; grid (64, rows) x 256
__global__ __launch_bounds__(256) void k_kd_reset(const RunConst *__restrict__ rcp) {
    const RunConst &rc = rcp[blockIdx.y];
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < (uint32_t)(128 * 128); i += gridDim.x * 256u) rc.kd_hint[i] = 0ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double x = rc.nx[0], y = rc.ny[0];
        rc.cnt->g_len = 1;
        rc.cnt->g_first_dup[0] = 0xFFFFFFFFu;
        rc.cnt->g_first_dup[1] = 0xFFFFFFFFu;
        if (x == rc.gp_x && y == rc.gp_y) { rc.cnt->g_first_dup[0] = 0; rc.cnt->g_nd_len = 0; }
        else { rc.g_nd[0] = 0; rc.g_nd_x[0] = x; rc.g_nd_y[0] = y; rc.g_nd_box[0] = g_box_after(g_box_all(), x, y, 0u, rc.gp_x, rc.gp_y); rc.cnt->g_nd_len = 1; }
        rc.kd_rec[0].child[0] = kEmpty; rc.kd_rec[0].child[1] = kEmpty;
        rc.cnt->kd_done = 1;
        rc.cnt->kd_snap = 0;
        rc.cnt->n_losers = 0;
        rc.g_snap[0] = 1; rc.g_snap[1] = rc.cnt->g_nd_len; rc.g_snap[2] = rc.cnt->g_first_dup[0]; rc.g_snap[3] = rc.cnt->g_first_dup[1];
    }
}

// Insert this step's nodes into the reference's kd-tree in id order (KdTree::add, nearest_neighbor.rs:29-46;
// rrt.rs:163) -- only the STRUCTURE is kept (child / parent / depth / cell), searches never use it.  It exists to
// reproduce the order in which the reference resolves equal-cost parents (kd pre-order).  Two kernels:
//   k_kd_locate  one wave per new node: find the empty slot of the tree as it stood before the step.  The descent
//                does not start at the root: a point's root path passes a node iff the point lies in the cell the
//                node was inserted into (kd_box), so it starts at the deepest node whose cell covers the whole
//                square of a 128 x 128 hint grid the point falls into (kd_hint) -- a few levels above the slot
//                however deep and lopsided the tree is (RRT insertion order gives depths of 100+).  Only when that node lies
//                on the goal path G (the path of the point every 100th iteration re-adds: one exact duplicate deeper
//                each time, thousands of levels late in a run) is the exit from G computed the long way: duplicate
//                levels collapse to two comparisons, only the non-duplicate levels are tested.
//   k_kd_hint    one wave per new node: raise the hint of every grid square its cell covers completely
//   k_kd_claim   one workgroup: nodes that reached the same empty slot are ordered by rounds -- the lowest id
//                takes the slot (atomicMin), the others step below it.  Contenders of one slot always arrive in
//                the same round because they share the whole path above it, so this equals sequential insertion.
enum : uint32_t { LOC_SIDE = 1u, LOC_ONPATH = 2u };
// Beside a persistent step loop (k_coop_rrt) there are no stream events between the steps: the first kernel of a kd group is
// launched ahead and waits here until the steps it inserts have been filed (one thread per workgroup polls, asleep in between).
// Gives up after two seconds of wall clock and says so (coop_abort): a kernel that can wait must not be able to hang the GPU.
constexpr uint32_t kWaitFiled = 0x80000000u;
__device__ __forceinline__ bool coop_wait_filed(const RunConst &rc, uint32_t need) {
    __shared__ uint32_t s_ok;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        uint32_t ok = 1u;
        while (__hip_atomic_load(&rc.cnt->coop_filed, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(64);
            if (wall_clock64() - t0 > 200000000ull || __hip_atomic_load(&rc.cnt->coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                atomicOr(&rc.cnt->coop_abort, 1u); ok = 0u; break;
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0u;
}
constexpr int kHG = 128;             // hint grid squares per axis
constexpr uint32_t kHintMaxSquares = 256;    // (hint_max of the rows of a batch)    // cells covering more squares leave the hints alone (kd_hint_block)

__device__ __forceinline__ KdBox load_box(const KdBox *p, size_t i) {
    auto g = as_global(p);
    KdBox b;
    b.lox = g[i].lox; b.hix = g[i].hix; b.loy = g[i].loy; b.hiy = g[i].hiy;
    return b;
}
__device__ __forceinline__ bool box_holds(const KdBox &bx, double x, double y) {
    return bx.lox <= x && x < bx.hix && bx.loy <= y && y < bx.hiy;
}
// cell of the child slot `side` of a node at (wx, wy), depth d, whose own cell is bx
__device__ __forceinline__ void box_cut(KdBox &bx, double wx, double wy, uint32_t depth, uint32_t side) {
    if (depth & 1u) { if (side) bx.loy = wy; else bx.hiy = wy; }
    else { if (side) bx.lox = wx; else bx.hix = wx; }
}

// descend from node `cur` (depth dcur, cell bx, taking `side`) to an empty slot of the old tree; on return bx is
// the cell of that slot.  One dependent 24-byte load per level.
__device__ __forceinline__ void kd_descend(const RunConst &rc, uint32_t Nsnap, double vx, double vy, int &cur, uint32_t &dcur, uint32_t &side,
                                           KdBox &bx) {
    auto grec = as_global(rc.kd_rec);
    KdRec rec;
    rec.x = grec[cur].x; rec.y = grec[cur].y; rec.child[0] = grec[cur].child[0]; rec.child[1] = grec[cur].child[1];
    for (;;) {
        box_cut(bx, rec.x, rec.y, dcur, side);
        const int c = side ? rec.child[1] : rec.child[0];
        if ((uint32_t)c >= Nsnap) break;              // empty (kEmpty) or newer than the snapshot
        rec.x = grec[c].x; rec.y = grec[c].y; rec.child[0] = grec[c].child[0]; rec.child[1] = grec[c].child[1];
        cur = c;
        dcur += 1;
        side = kd_left(vx, vy, rec.x, rec.y, dcur) ? 0u : 1u;
    }
}

// One launch serves the new nodes of `nsteps` consecutive steps starting at step b0 (the structure is built beside the
// steps and may lag them, so several steps' nodes are inserted together): wave -> (step, sample).
// The descent sees the tree as far as k_kd_claim has published it when the kernel starts (cnt->kd_snap: nodes below
// n_at[kd_snap], G as recorded in g_snap[kd_snap]); newer nodes are treated as absent and k_kd_link / k_kd_claim
// finish the descent through them (with one side stream the published state is always the whole tree before the group).
// LPN = lanes per node: 64 (one wave per node: shortest latency, used when a launch has few nodes) or 1 (one thread
// per node, the non-duplicate levels of G staged in LDS: 64x fewer waves, used when many contexts are grown together
// and the GPU is short of wave slots, not of time).
constexpr uint32_t kTieParts = 8;        // workgroups that look at the deferred ties when they ride in another kernel
__device__ __forceinline__ void kd_hint_block(const RunConst &rc, uint32_t b0, uint32_t nsteps, uint32_t vwords, uint32_t bx);
template <int T>
__device__ __forceinline__ void tie_fix_block(const RunConst &rc, uint32_t part, uint32_t nparts);
// hb0, hns (hns > 0: a single query's side chain): the workgroups after the locating ones raise the hints of the group BEFORE this
// one (steps hb0 .. hb0 + hns - 1; a descent may start at any node whose cell holds the point, so hints that change under
// the descents are as good as the old ones) and look at the deferred ties -- two kernels less on the chain locate -> link -> claim.
template <int LPN>
__global__ __launch_bounds__(256) void k_kd_locate(const RunConst *__restrict__ rcp, uint32_t b0, uint32_t nsteps, uint32_t K, uint32_t nb_last,
                                                    uint32_t vwords, uint32_t lpar, uint32_t hb0 = 0, uint32_t hns = 0) {
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row (porrt_grow_batch)
    if (rc.kd_lazy && rc.cnt->lca_next <= b0) return;      // (the build after lazily tracked steps: only the rows and steps a tie asked for)
    if (hns) {
        const uint32_t lb = LPN == 64 ? (nsteps * K * 64u + 255u) / 256u : (nsteps * K + 255u) / 256u;
        if (blockIdx.x >= lb) {
            if (blockIdx.x + kTieParts >= gridDim.x) tie_fix_block<256>(rc, blockIdx.x + kTieParts - gridDim.x, kTieParts);
            else kd_hint_block(rc, hb0, hns, vwords, blockIdx.x - lb);
            return;
        }
    }
    if (lpar & kWaitFiled) { if (!coop_wait_filed(rc, b0 + nsteps)) return; lpar &= ~kWaitFiled; }
    const uint32_t lane = LPN == 64 ? (threadIdx.x & 63u) : 0u;
    const uint32_t wid = LPN == 64 ? blockIdx.x * 4u + (threadIdx.x >> 6) : blockIdx.x * 256u + threadIdx.x;
    const uint32_t st = wid / K, ks = wid - st * K;
    bool active = st < nsteps && ks < (st + 1 == nsteps ? nb_last : K);
    const uint32_t b = b0 + (active ? st : 0u);
    if (active && rc.sched_nb) active = ks < row_nb(rc, b, K);          // (the row's own step sizes: perm[] holds that many entries)
    // the step's samples in the spatial order of k_sort_samples: neighbouring threads descend through the same nodes, so a
    // wave's loads of a level fall into a few cache lines instead of 64
    const uint32_t k = active ? (uint32_t)as_global(rc.perm)[(size_t)b * rc.part_stride + ks] : 0u;
    // The new nodes are the valid samples (positions known since k_near), id = n_at[b] + rank in their step.
    const size_t o2 = (size_t)b * rc.part_stride + (active ? k : 0u);
    if (active && as_global(rc.kq_vid)[o2] < 0) active = false;
    PORRT_T0();
    const uint32_t bsnap = __hip_atomic_load(&rc.cnt->kd_snap, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t glen0 = as_global(rc.g_snap)[4 * bsnap + 0], n_nd = as_global(rc.g_snap)[4 * bsnap + 1];
    // (a goal path has a few dozen non-duplicate levels; the LDS of a side-stream workgroup is LDS the step kernels beside it
    // cannot have, and their occupancy hangs on it)
    constexpr uint32_t kNdLds = 128;
    __shared__ double s_ndx[LPN == 1 ? kNdLds : 1], s_ndy[LPN == 1 ? kNdLds : 1];
    __shared__ uint32_t s_ndi[LPN == 1 ? kNdLds : 1];
    if (LPN == 1) {
        for (uint32_t t2 = threadIdx.x; t2 < n_nd && t2 < kNdLds; t2 += 256u) {
            s_ndi[t2] = as_global(rc.g_nd)[t2]; s_ndx[t2] = as_global(rc.g_nd_x)[t2]; s_ndy[t2] = as_global(rc.g_nd_y)[t2];
        }
        __syncthreads();
    }
    if (!active) return;
    const uint32_t Nsnap = as_global(rc.n_at)[bsnap], N = as_global(rc.n_at)[b0];
    const uint32_t t = as_global(rc.n_at)[b] - N + rank_before(rc, b, vwords, k);
    const double px = rc.gp_x, py = rc.gp_y;
    const double vx = as_global(rc.kq_x)[o2], vy = as_global(rc.kq_y)[o2];
    if (lane == 0) {   // the node's kd record exists from here on (k_kd_link only links it)
        KdRec rec;
        rec.x = vx; rec.y = vy; rec.child[0] = kEmpty; rec.child[1] = kEmpty;
        rc.kd_rec[N + t] = rec;
    }
    // start of the descent: the deepest node whose cell covers the point's whole hint-grid square (k_kd_hint)
    int bd = -1, bu = -1;
    {
        int hx, hy;
        rep_cell(rc, vx, vy, kHG, hx, hy);
        const unsigned long long hv = __hip_atomic_load(&rc.kd_hint[hy * kHG + hx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bu = (int)(uint32_t)hv;
        while ((uint32_t)bu >= Nsnap) bu = __hip_atomic_load(&rc.kd_up[bu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // newer than the snapshot
        bd = (int)as_global(rc.kd_depth)[bu];
        if (!box_holds(load_box(rc.kd_box, (size_t)bu), vx, vy)) bu = -1;     // cannot happen (see k_kd_hint); the long way is always right
    }
    PORRT_TACC_C(rc, 0);
    int cur;
    uint32_t dcur, side, gex = 0, flags = 0;
    KdBox bx;
    bool long_way = bu < 0;
    if (!long_way) {
        const uint32_t ge = as_global(rc.kd_gexit)[bu];
        if (ge & kOnG) long_way = true;
        else {
            cur = bu; dcur = (uint32_t)bd; gex = ge;
            bx = load_box(rc.kd_box, (size_t)bu);
            side = kd_left(vx, vy, as_global(rc.kd_rec)[bu].x, as_global(rc.kd_rec)[bu].y, dcur) ? 0u : 1u;
            kd_descend(rc, Nsnap, vx, vy, cur, dcur, side, bx);
        }
    }
    PORRT_TACC_C(rc, 1);
#if defined(PORRT_TIMING) && PORRT_TIMING == 3
    if (LPN == 1) { const unsigned long long lw = __ballot(long_way); if ((threadIdx.x & 63u) == 0u) { atomicAdd(&rc.cnt->tim[6], (unsigned long long)__popcll(lw)); atomicAdd(&rc.cnt->tim[14], lw ? 1ull : 0ull); } }
#endif
    if (long_way) {
        // Where does this node's descent leave the goal path G?  At the first level whose test it fails.  A level
        // held by an exact duplicate of the goal point tests `x < p.x` (even depth) or `y < p.y` (odd depth), and
        // the goal point itself always goes right there, so all duplicate levels reduce to two comparisons against
        // the first duplicate of each parity; only the few non-duplicate levels (g_nd*) are real tests.
        uint32_t E = 0xFFFFFFFFu, leftE = 0;
        for (uint32_t s0 = lane; s0 < n_nd; s0 += (uint32_t)LPN) {
            uint32_t ii;
            double wx, wy;
            if (LPN == 1 && s0 < kNdLds) { ii = s_ndi[s0]; wx = s_ndx[s0]; wy = s_ndy[s0]; }
            else { ii = as_global(rc.g_nd)[s0]; wx = as_global(rc.g_nd_x)[s0]; wy = as_global(rc.g_nd_y)[s0]; }
            const bool gl = kd_left(px, py, wx, wy, ii), vl = kd_left(vx, vy, wx, wy, ii);
            if (vl != gl && ii < E) { E = ii; leftE = vl ? 1u : 0u; }
        }
        if (LPN == 64) {
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t oe = __shfl_xor(E, off), ol = __shfl_xor(leftE, off);
                if (oe < E) { E = oe; leftE = ol; }
            }
        }
        const uint32_t d0 = as_global(rc.g_snap)[4 * bsnap + 2], d1 = as_global(rc.g_snap)[4 * bsnap + 3];
        if (vx < px && d0 < E) { E = d0; leftE = 1u; }
        if (vy < py && d1 < E) { E = d1; leftE = 1u; }
        if (E == 0xFFFFFFFFu) {             // on G to its end: below the last node, on the goal point's side
            dcur = glen0 - 1;
            cur = as_global(rc.g_id)[dcur];
            const double gxx = as_global(rc.g_x)[dcur], gyy = as_global(rc.g_y)[dcur];
            side = kd_left(px, py, gxx, gyy, dcur) ? 0u : 1u;
            flags = LOC_ONPATH;
            bx = load_box(rc.kd_box, (size_t)cur);
            box_cut(bx, gxx, gyy, dcur, side);
        } else {
            dcur = E;
            gex = E;
            cur = as_global(rc.g_id)[E];
            side = leftE ? 0u : 1u;
            bx = load_box(rc.kd_box, (size_t)cur);
            kd_descend(rc, Nsnap, vx, vy, cur, dcur, side, bx);
        }
    }
    PORRT_TACC_C(rc, 2);
    if (lane == 0) {
        const uint32_t lo = lpar * rc.loc_stride + t;
        rc.loc_cur[lo] = cur;
        rc.loc_dcur[lo] = dcur;
        rc.loc_gex[lo] = gex;
        rc.loc_flags[lo] = flags | (side ? LOC_SIDE : 0u);
        rc.loc_box[lo] = bx;
        atomicMin(&rc.kd_rec[cur].child[side], (int)(N + t));     // first round's bid for the slot (k_kd_link)
    }
}

// Linking the located nodes into the tree.  Nodes that stopped at the same empty slot must be ordered as sequential
// insertion in id order would order them: the lowest id takes the slot, the others step below it and contend for its
// child slots, and so on ("rounds"; contenders of one slot always arrive in the same round because they share the
// whole path above it).
//   k_kd_locate  ends with the first round's bid: atomicMin of the id on the slot it found.
//   k_kd_link    one thread per node, whole GPU: a node that holds its slot is published (parent, depth, cell, goal
//                path bookkeeping); a loser steps below the winner and is parked in the loser list.
//   k_kd_claim   one workgroup: plays the remaining rounds for the (few) losers -- every slot they can still reach
//                belongs to a node of this launch, so the slots live in LDS.  While many nodes are still moving the
//                whole workgroup plays a round; the last <= 64 are handed to ONE wave, whose rounds need no barrier,
//                and contenders that are all the same point (the step's copies of the goal point, re-added every
//                100th iteration) are settled as a chain at once.  Then it publishes how far the structure is complete.
constexpr uint32_t kClaimMax = 4096;

// publish one node: parent link, depth, cell, and the goal path bookkeeping
__device__ __forceinline__ void kd_publish(const RunConst &rc, uint32_t N, const KdMove &m, int parent_id) {
    const int w = (int)(N + m.t);
    const uint32_t dw = m.dcur + 1u;
    as_global(rc.kd_rec)[parent_id].child[m.side] = w;
    as_global(rc.kd_up)[w] = parent_id;
    as_global(rc.kd_depth)[w] = dw;
    rc.kd_box[w] = m.box;
    if (m.onpath) {
        if (dw + 8 < rc.g_cap) { as_global(rc.g_id)[dw] = w; as_global(rc.g_x)[dw] = m.vx; as_global(rc.g_y)[dw] = m.vy; }
        else atomicOr(&rc.cnt->err, (uint32_t)ERR_GPATH_OVERFLOW);
        as_global(rc.kd_gexit)[w] = dw | kOnG;
        atomicMax(&rc.cnt->g_len, dw + 1);
        if (m.vx == rc.gp_x && m.vy == rc.gp_y) atomicMin(&rc.cnt->g_first_dup[dw & 1u], dw);
        else {
            const uint32_t sl = atomicAdd(&rc.cnt->g_nd_len, 1u);
            as_global(rc.g_nd)[sl] = dw; as_global(rc.g_nd_x)[sl] = m.vx; as_global(rc.g_nd_y)[sl] = m.vy;
        }
    } else {
        as_global(rc.kd_gexit)[w] = m.gex;
    }
}

// step below the winner, a node of this launch with index tw at (wx, wy)
__device__ __forceinline__ void kd_step_below(const RunConst &rc, KdMove &m, uint32_t tw, double wx, double wy) {
    const uint32_t dw = m.dcur + 1u;
    const bool vl = kd_left(m.vx, m.vy, wx, wy, dw);
    if (m.onpath && vl != kd_left(rc.gp_x, rc.gp_y, wx, wy, dw)) { m.onpath = false; m.gex = dw; }
    m.cur = (int)tw; m.dcur = dw; m.side = vl ? 0u : 1u;
    box_cut(m.box, wx, wy, dw, m.side);
}

// new nodes of the steps [b0, b0 + nsteps): called by whole waves (every lane gets the sum).  A word of the masks per lane and a
// sum over the wave: one trip to memory (it used to be one thread's loop over up to 128 words through a pointer of unknown
// address space -- a drained round trip per word, at the start of three kernels of every group).
__device__ __forceinline__ uint32_t kd_group_size(const RunConst &rc, uint32_t b0, uint32_t nsteps, uint32_t vwords) {
    auto vm = as_global(rc.valid_mask) + (size_t)b0 * vwords;
    const uint32_t lane = threadIdx.x & 63u, nw = vwords * nsteps;
    uint32_t n_new = 0;
    for (uint32_t w = lane; w < nw; w += 64u) n_new += (uint32_t)__popcll(vm[w]);
    return wave_sum(n_new);
}

__global__ __launch_bounds__(256) void k_kd_link(const RunConst *__restrict__ rcp, uint32_t b0, uint32_t nsteps, uint32_t vwords, uint32_t lpar) {
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row (porrt_grow_batch)
    if (rc.kd_lazy && rc.cnt->lca_next <= b0) return;
    const uint32_t N = as_global(rc.n_at)[b0];
    const uint32_t n_new = kd_group_size(rc, b0, nsteps, vwords);
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_new || n_new > kClaimMax) return;
    const uint32_t lo = lpar * rc.loc_stride + t;
    KdMove m;
    m.t = t;
    const uint32_t fl = as_global(rc.loc_flags)[lo];
    m.onpath = fl & LOC_ONPATH;
    m.side = (fl & LOC_SIDE) ? 1u : 0u;
    m.cur = as_global(rc.loc_cur)[lo];
    m.dcur = as_global(rc.loc_dcur)[lo];
    m.gex = as_global(rc.loc_gex)[lo];
    m.box = rc.loc_box[lo];
    m.vx = as_global(rc.kd_rec)[N + t].x;
    m.vy = as_global(rc.kd_rec)[N + t].y;
    const int w = as_global(rc.kd_rec)[m.cur].child[m.side];           // the bids were placed by k_kd_locate
    if (w == (int)(N + t)) { kd_publish(rc, N, m, m.cur); return; }
    if ((uint32_t)w < N) { atomicOr(&rc.cnt->err, (uint32_t)ERR_GPATH_OVERFLOW); return; }    // cannot happen: the slot was empty in the old tree
    kd_step_below(rc, m, (uint32_t)w - N, as_global(rc.kd_rec)[w].x, as_global(rc.kd_rec)[w].y);
    // (at most n_new <= kClaimMax losers per group, and the claim kernel takes the count back to zero; the bound is checked all the
    // same -- an append past the array would be a fault of the whole GPU, an error code is not)
    const uint32_t li = atomicAdd(&rc.cnt->n_losers, 1u);
    if (li < kClaimMax) rc.kd_losers[li] = m;
    else atomicOr(&rc.cnt->err, (uint32_t)ERR_GPATH_OVERFLOW);
}

// A workgroup barrier that waits for the LDS traffic only: the rounds below talk through LDS alone, and a barrier that also
// drains the global stores of kd_publish (what __syncthreads does) costs a round trip to memory per round.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");      // orders (and waits for) the LDS accesses only
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// CAP: the most nodes a launch may hold (the slots in LDS, the losers in registers); the engine takes the smaller form when the
// group's steps cannot hold more -- beside the step kernels every KB of LDS counts.
// TPB: threads of the workgroup (CAP / TPB losers per thread; the losers are the first n_l entries, so with few of them only the
// first one or two per thread exist).  A workgroup must find all its wave slots free at once: beside the step kernels of a batch a
// 16-wave workgroup waits for them several times longer than its rounds take, a 4-wave one does not.
// LDSXY (a single query: the GPU's LDS is idle): the new nodes' coordinates are staged in LDS, so that a round's "step below the
// winner" reads them there instead of through a dependent load from memory -- the rounds are this kernel's whole run time.
template <uint32_t CAP, bool LDSXY = false, uint32_t TPB = 1024>
__global__ __launch_bounds__(TPB) void k_kd_claim(const RunConst *__restrict__ rcp, uint32_t b0, uint32_t nsteps, uint32_t vwords) {
    static_assert(CAP % TPB == 0 && TPB % 64u == 0 && TPB <= 1024u && CAP <= kClaimMax, "k_kd_claim: CAP, TPB");
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row (porrt_grow_batch)
    if (rc.kd_lazy && rc.cnt->lca_next <= b0) return;
    __shared__ int s_ch[CAP][2];
    __shared__ dbl2 s_xy[LDSXY ? CAP : 1];
    __shared__ uint32_t s_nact;
    __shared__ KdMove s_tail[64];
    const uint32_t N = as_global(rc.n_at)[b0], b = b0 + nsteps - 1u;
    const uint32_t n_new = kd_group_size(rc, b0, nsteps, vwords);
    if (n_new > CAP) { if (threadIdx.x == 0) atomicOr(&rc.cnt->err, (uint32_t)ERR_GPATH_OVERFLOW); return; }
    const uint32_t n_l = rc.cnt->n_losers < kClaimMax ? rc.cnt->n_losers : kClaimMax;      // (k_kd_link stores no loser past the array)
    auto grec = as_global(rc.kd_rec);
#ifdef PORRT_CLAIM_PROBE
    const unsigned long long pt0 = wall_clock64();
    uint32_t p_rounds = 0, p_tail = 0;
#endif
    if (n_l) {
        for (uint32_t t = threadIdx.x; t < n_new; t += TPB) {
            s_ch[t][0] = kEmpty; s_ch[t][1] = kEmpty;
            if (LDSXY) { dbl2 v; v.x = grec[N + t].x; v.y = grec[N + t].y; s_xy[t] = v; }
        }
        constexpr int kPer = CAP / TPB;
        bool todo[kPer];
        KdMove mv[kPer];
#pragma unroll
        for (int r = 0; r < kPer; ++r) {
            const uint32_t q = threadIdx.x + r * TPB;
            todo[r] = q < n_l;
            mv[r] = rc.kd_losers[todo[r] ? q : 0u];
        }
        __syncthreads();
        // Losers are few next to the threads (a few hundred of a group's 2048 nodes): a wave none of whose lanes holds one
        // has nothing to do in any round and leaves now -- the barriers below count the waves that are still there -- so that
        // the step kernels running beside this workgroup get its wave slots and registers back.  (Wave 0 holds the first
        // losers and does the closing part.)
        if (threadIdx.x >= 64u && !__ballot(todo[0])) return;
        auto settle = [&](KdMove &m, bool &td) {            // after the bids of a round
            const int w = s_ch[m.cur][m.side];
            if (w == (int)m.t) { kd_publish(rc, N, m, (int)(N + (uint32_t)m.cur)); td = false; }
            else if (LDSXY) { const dbl2 v = s_xy[w]; kd_step_below(rc, m, (uint32_t)w, v.x, v.y); }
            else kd_step_below(rc, m, (uint32_t)w, grec[N + (uint32_t)w].x, grec[N + (uint32_t)w].y);
        };
        for (;;) {
            uint32_t mine = 0;
#pragma unroll
            for (int r = 0; r < kPer; ++r) mine += todo[r] ? 1u : 0u;
            if (threadIdx.x == 0) s_nact = 0;
            lds_barrier();
            if (mine) atomicAdd(&s_nact, mine);
            lds_barrier();
            if (s_nact <= 64u) break;
#ifdef PORRT_CLAIM_PROBE
            ++p_rounds;
#endif
#pragma unroll
            for (int r = 0; r < kPer; ++r)
                if (todo[r]) atomicMin(&s_ch[mv[r].cur][mv[r].side], (int)mv[r].t);
            lds_barrier();
#pragma unroll
            for (int r = 0; r < kPer; ++r)
                if (todo[r]) settle(mv[r], todo[r]);
        }
        if (s_nact) {
            lds_barrier();
            if (threadIdx.x == 0) s_nact = 0;
            lds_barrier();
#pragma unroll
            for (int r = 0; r < kPer; ++r)
                if (todo[r]) s_tail[atomicAdd(&s_nact, 1u)] = mv[r];
            lds_barrier();
            if (threadIdx.x < 64u) {
                bool td = threadIdx.x < s_nact;
                KdMove m = s_tail[td ? threadIdx.x : 0u];
                while (__ballot(td)) {
#ifdef PORRT_CLAIM_PROBE
                    ++p_tail;
#endif
                    {   // contenders of one slot that are all the same point form a chain in id order: settle it at once
                        const uint32_t slot = ((uint32_t)m.cur << 1) | m.side;
                        unsigned long long rem = __ballot(td);
                        while (rem) {
                            const int l0 = (int)__builtin_ctzll(rem);
                            const uint32_t slot0 = (uint32_t)__builtin_amdgcn_readlane((int)slot, l0);
                            const double x0 = __shfl(m.vx, l0), y0 = __shfl(m.vy, l0);
                            const unsigned long long grp = __ballot(td && slot == slot0);
                            const unsigned long long same = __ballot(td && slot == slot0 && m.vx == x0 && m.vy == y0);
                            rem &= ~grp;
                            if (same != grp || __popcll(grp) < 2) continue;
                            const bool in = (grp >> threadIdx.x) & 1ull;
                            uint32_t rank = 0, par = 0;
                            for (unsigned long long mm = grp; mm;) {
                                const int l = (int)__builtin_ctzll(mm);
                                mm &= mm - 1;
                                const uint32_t tl = (uint32_t)__builtin_amdgcn_readlane((int)m.t, l);
                                if (in && tl < m.t) { ++rank; par = (rank == 1 || tl > par) ? tl : par; }
                            }
                            if (in) {
                                if (rank == 0) {
                                    s_ch[m.cur][m.side] = (int)m.t;
                                    kd_publish(rc, N, m, (int)(N + (uint32_t)m.cur));
                                } else {
                                    for (uint32_t i = 0; i < rank; ++i) kd_step_below(rc, m, par, m.vx, m.vy);   // same point at every level
                                    s_ch[par][m.side] = (int)m.t;
                                    kd_publish(rc, N, m, (int)(N + par));
                                }
                                td = false;
                            }
                        }
                    }
                    if (td) atomicMin(&s_ch[m.cur][m.side], (int)m.t);
                    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0) only: the LDS atomics, not the stores of kd_publish
                    __builtin_amdgcn_wave_barrier();
                    if (td) settle(m, td);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    // every record of the group is written: connect kernels running beside us may now trust ids < N + n_new
    __threadfence();
    __syncthreads();
#ifdef PORRT_CLAIM_PROBE
    if (threadIdx.x == 0) {      // developer statistics: losers (max, sum), launches, rounds, tail rounds, time
        atomicMax(&rc.cnt->dbg[0], n_l); atomicAdd(&rc.cnt->dbg[1], n_l); atomicAdd(&rc.cnt->dbg[2], 1u); atomicAdd(&rc.cnt->dbg[3], p_rounds);
        { const unsigned long long dt = wall_clock64() - pt0; atomicAdd(&rc.cnt->tim[0], dt); atomicMax(&rc.cnt->tim[2], dt); rc.cnt->tim[10] = 1ull; } atomicAdd(&rc.cnt->tim[8], 1ull); atomicAdd(&rc.cnt->tim[1], (unsigned long long)p_tail * 100ull); atomicAdd(&rc.cnt->tim[9], 1ull);
    }
#endif
    if (threadIdx.x == 0) {
        rc.cnt->n_losers = 0;
        // G as the next k_kd_locate may see it
        as_global(rc.g_snap)[4 * (b + 1) + 0] = __hip_atomic_load(&rc.cnt->g_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        as_global(rc.g_snap)[4 * (b + 1) + 1] = __hip_atomic_load(&rc.cnt->g_nd_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        as_global(rc.g_snap)[4 * (b + 1) + 2] = __hip_atomic_load(&rc.cnt->g_first_dup[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        as_global(rc.g_snap)[4 * (b + 1) + 3] = __hip_atomic_load(&rc.cnt->g_first_dup[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __hip_atomic_store(&rc.cnt->kd_snap, b + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&rc.cnt->kd_done, N + n_new, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// After k_kd_claim: a new node whose cell covers a hint-grid square completely is an ancestor of every point that
// will ever fall into that square; the deepest such node is the best place to start a descent.  The squares between
// cell(lo) and cell(hi), both excluded, lie inside [lo, hi) because the cell function is monotone; an infinite
// bound includes the clamped border square.  hint = max over (depth, id), a commutative update.
__device__ __forceinline__ void kd_hint_block(const RunConst &rc, uint32_t b0, uint32_t nsteps, uint32_t vwords, uint32_t bx) {
    // One thread per new node: late in a run a node's cell covers no whole square, or a few, and the thread raises
    // them itself; the rare big cells (young tree) are parked in LDS and shared out to the workgroup's waves.
    __shared__ uint32_t s_nbig, s_big_id[256];
    __shared__ int s_big_r[256][4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t t = bx * 256u + threadIdx.x;
    if (rc.kd_lazy && rc.cnt->lca_next <= b0) return;
    const uint32_t n_new = kd_group_size(rc, b0, nsteps, vwords);
    if (threadIdx.x == 0) s_nbig = 0;
    __syncthreads();
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    auto gh = as_global(rc.kd_hint);
    const uint32_t N = as_global(rc.n_at)[b0];
    if (t < n_new) {
        const uint32_t id = N + t;
        const KdBox bx = rc.kd_box[id];
        int lx, ly, ux, uy;
        rep_cell(rc, bx.lox, bx.loy, kHG, lx, ly);
        rep_cell(rc, bx.hix, bx.hiy, kHG, ux, uy);
        const int ix0 = bx.lox == -INF ? 0 : lx + 1, ix1 = bx.hix == INF ? kHG - 1 : ux - 1;
        const int iy0 = bx.loy == -INF ? 0 : ly + 1, iy1 = bx.hiy == INF ? kHG - 1 : uy - 1;
        if (ix0 <= ix1 && iy0 <= iy1) {
            const uint32_t w = (uint32_t)(ix1 - ix0 + 1), n = w * (uint32_t)(iy1 - iy0 + 1);
            if (n > rc.hint_max) {
                // (a young tree's shallow nodes in a batch of many rows: a hint only shortens a descent, and a square they cover is soon
                // covered by deeper nodes; raising thousands of squares per node -- with a thousand young trees grown together, tens
                // of millions of atomics per group -- costs more than the levels it saves.  A single query keeps them: its kd chain
                // is on its critical path and the atomics are few)
            } else if (n <= 8u) {
                const unsigned long long val = ((unsigned long long)as_global(rc.kd_depth)[id] << 32) | id;
                for (uint32_t i = 0; i < n; ++i) {
                    const uint32_t ry = i / w;
                    __hip_atomic_fetch_max(gh + (size_t)(iy0 + (int)ry) * kHG + ix0 + (int)(i - ry * w), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                const uint32_t q = atomicAdd(&s_nbig, 1u);
                s_big_id[q] = id; s_big_r[q][0] = ix0; s_big_r[q][1] = ix1; s_big_r[q][2] = iy0; s_big_r[q][3] = iy1;
            }
        }
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x >> 6; q < s_nbig; q += 4u) {
        const uint32_t id = s_big_id[q];
        const int ix0 = s_big_r[q][0], ix1 = s_big_r[q][1], iy0 = s_big_r[q][2], iy1 = s_big_r[q][3];
        const uint32_t w = (uint32_t)(ix1 - ix0 + 1), n = w * (uint32_t)(iy1 - iy0 + 1);
        const unsigned long long val = ((unsigned long long)as_global(rc.kd_depth)[id] << 32) | id;
        for (uint32_t i = lane; i < n; i += 64u) {
            const uint32_t ry = i / w;
            __hip_atomic_fetch_max(gh + (size_t)(iy0 + (int)ry) * kHG + ix0 + (int)(i - ry * w), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Resolve deferred equal-cost parents whose tied nodes are all in the kd structure by now.  One workgroup, one
// wave per record, starting at the first record not yet settled; runs on the kd stream after each k_kd_claim and
// once more at the end of a run.  The parent is only installed if no rewire replaced the placeholder in the
// meantime (a rewire is final, rrt.rs:152-161).
template <int T>
__device__ __forceinline__ void tie_fix_block(const RunConst &rc, uint32_t part, uint32_t nparts) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t n = __hip_atomic_load(&rc.cnt->pend_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    n = n < rc.pend_cap ? n : rc.pend_cap;
    const uint32_t lo = rc.cnt->pend_lo;
    const uint32_t kd_done = __hip_atomic_load(&rc.cnt->kd_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    Team<1> tm;
    tm.scr_d = nullptr; tm.scr_i = nullptr; tm.wave = 0; tm.lane = lane;
    for (uint32_t p = lo + part * (uint32_t)(T / 64) + wv; p < n; p += nparts * (uint32_t)(T / 64)) {        // (records striped over the workgroups that look at them)
        if (__hip_atomic_load(&rc.pend_state[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1u) continue;
        const uint32_t base = as_global(rc.pend_off)[p], m = as_global(rc.pend_n)[p];
        int mx = -1;
        for (uint32_t a = lane; a < m; a += 64u) { const int j = as_global(rc.pend_pool)[base + a]; mx = j > mx ? j : mx; }
        mx = tm.max_i(mx);
        if ((uint32_t)mx >= kd_done) continue;
        int on_min = kEmpty, off_best = kEmpty;
        for (uint32_t a = lane; a < m; a += 64u) {
            const int j = as_global(rc.pend_pool)[base + a];
            if (rc.kd_gexit[j] & kOnG) on_min = j < on_min ? j : on_min;
            else if (off_best == kEmpty || kd_preorder_less(rc, j, off_best)) off_best = j;
        }
        on_min = tm.min_i(on_min);
        off_best = tm.first_preorder(rc, off_best);
        int best;
        if (off_best == kEmpty) best = on_min;
        else if (on_min == kEmpty) best = off_best;
        else best = kd_preorder_less(rc, on_min, off_best) ? on_min : off_best;
        if (lane == 0) {
            atomicCAS(&rc.parent[as_global(rc.pend_new)[p]], kParentPending, best);
            as_global(rc.pend_state)[p] = 2u;
            atomicAdd(&rc.cnt->n_deferred, 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && part == 0) {       // records settle roughly in order: skip the settled prefix next time
        uint32_t l = lo;
        while (l < n && __hip_atomic_load(&rc.pend_state[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u) ++l;
        rc.cnt->pend_lo = l;
    }
}
template <int T>
__global__ __launch_bounds__(T) void k_tie_fix(const RunConst *__restrict__ rcp) {
    tie_fix_block<T>(rcp[blockIdx.y], 0u, 1u);      // one context per grid row (porrt_grow_batch)
}
__global__ __launch_bounds__(256) void k_kd_hint(const RunConst *__restrict__ rcp, uint32_t b0, uint32_t nsteps, uint32_t vwords) {
    kd_hint_block(rcp[blockIdx.y], b0, nsteps, vwords, blockIdx.x);      // one context per grid row (porrt_grow_batch)
}
// the same with the deferred ties looked at by one more workgroup (a single query's side chain: one kernel less on it)
__global__ __launch_bounds__(256) void k_kd_hint_fix(const RunConst *__restrict__ rcp, uint32_t b0, uint32_t nsteps, uint32_t vwords) {
    if (blockIdx.x + kTieParts >= gridDim.x) { tie_fix_block<256>(rcp[blockIdx.y], blockIdx.x + kTieParts - gridDim.x, kTieParts); return; }
    kd_hint_block(rcp[blockIdx.y], b0, nsteps, vwords, blockIdx.x);
}

// PTO: edges to every neighbour with a valid transition, reachability phase 1 and 2 (pto.rs:95-124)
template <bool LDSGRID>
__global__ __launch_bounds__(kConnectWaves * 64) void k_connect_pto(const RunConst *__restrict__ rcp, uint32_t b, uint32_t nb,
                                                                     uint32_t vwords) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_tiles[];
    __shared__ __attribute__((aligned(16))) uint8_t s_ins[kInsertLds];
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row (porrt_grow_batch)
    nb = row_nb(rc, b, nb);
    if (blockIdx.x == gridDim.x - 1) { if (nb) insert_step_pages(rc, b, nb, vwords, s_ins); return; }    // the extra block
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = uni(blockIdx.x * kConnectWaves + (threadIdx.x >> 6));
    if (k >= nb || as_global(rc.q_vid)[k] < 0) return;
    const double px = as_global(rc.q_x)[k], py = as_global(rc.q_y)[k];
    GlobalGrid ggrid;
    ggrid.p = rc.cls; ggrid.W = rc.W;
    TileGrid grid;
    if (LDSGRID) {
        const uint32_t TW = 2u * rc.tile_R + 1u;
        grid = load_tile(rc, lds_tiles + (threadIdx.x >> 6) * ((TW * TW + 15u) & ~15u), px, py, lane, 64u);
        __builtin_amdgcn_wave_barrier();
    } else {
        grid.lds = nullptr; grid.glob = rc.cls; grid.W = rc.W; grid.TW = 0; grid.oi = 0; grid.oj = 0;
    }
    const uint32_t N = as_global(rc.n_at)[b];
    const uint32_t id = N + rank_before(rc, b, vwords, k);
    uint32_t cnt = cand_count(rc, b, k);
    int *cid = rc.cand_id + cand_off(rc, b, k);
    double *cval = rc.cand_val + cand_val_off(rc, b, k);
    dbl2 *cxy = reinterpret_cast<dbl2 *>(rc.cand_xy) + cand_off(rc, b, k);
    if (cnt == 0) {                       // pto.rs:99: nobody in range -> the nearest node
        if (lane == 0) {                  // lane 0 is also the only reader of slot 0
            const int nn = as_global(rc.q_nn)[k];
            dbl2 v;
            v.x = as_global(rc.nx)[nn]; v.y = as_global(rc.ny)[nn];
            cid[0] = nn; cxy[0] = v; as_global(rc.cand_cnt)[cand_cnt_at(rc, b, k)] = 1;
        }
        cnt = 1;
    }
    uint32_t err = 0;
    unsigned long long r_new = 0;
    uint32_t n_edges = 0;
    for (uint32_t a = lane; a < cnt; a += 64) {
        const int j = cid[a];
        const dbl2 axy = cxy[a];
        const int cls = traversed_class(rc, grid, axy.x, axy.y, px, py, &err);
        const int tv = class_to_validity(rc, cls);
        cval[a] = (double)tv;
        if (tv >= 0) {
            r_new |= as_global(rc.reachA)[j] & rc.validities[tv];      // pto.rs:111-114, pto_reachability.rs:42-52
            ++n_edges;
        }
    }
    r_new = wave_or(r_new);
    // edge slots: one atomic per wave, lanes take consecutive slots
    uint32_t lane_off = n_edges;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(lane_off, off);
        if ((int)lane >= off) lane_off += v;
    }
    const uint32_t total = __shfl(lane_off, 63);
    uint32_t base = 0;
    if (lane == 0 && total) base = atomicAdd(&rc.cnt->n_edges, total);
    base = __shfl(base, 0);
    uint32_t slot = base + lane_off - n_edges;
    for (uint32_t a = lane; a < cnt; a += 64) {
        const int tv = (int)cval[a];
        const int j = cid[a];
        if (tv >= 0) {
            if (slot < rc.e_cap) {
                as_global(rc.e_from)[slot] = (uint32_t)j;
                as_global(rc.e_to)[slot] = id;
                as_global(rc.e_tv)[slot] = (uint32_t)tv;
            } else {
                err |= ERR_EDGE_OVERFLOW;
            }
            ++slot;
            atomicOr(&rc.reachB[j], r_new & rc.validities[tv]);   // pto.rs:117-120
        }
    }
    if (lane == 0) {
        as_global(rc.nx)[id] = px;
        as_global(rc.ny)[id] = py;
        rep_insert(rc, px, py, (int)id);
        as_global(rc.parent)[id] = -1;
        as_global(rc.distA)[id] = 0.0;
        as_global(rc.distB)[id] = 0.0;
        as_global(rc.vid)[id] = (uint8_t)as_global(rc.q_vid)[k];
        as_global(rc.reachA)[id] = r_new;
        as_global(rc.reachB)[id] = r_new;
        unsigned long long mask = 0;
        const bool fin = goal_hit(rc, ggrid, px, py, mask, &err);
        as_global(rc.final_flag)[id] = fin ? 1 : 0;
        as_global(rc.final_mask)[id] = fin ? mask : 0ull;
        if (fin) {
            atomicAdd(&rc.cnt->n_final, 1u);
            atomicOr(&rc.cnt->finality, r_new & mask);
        }
    }
    if (err) atomicOr(&rc.cnt->err, err);
}

// PTO: publish the reach masks touched in this step and refresh the finality (pto_reachability.rs:92-101)
__global__ __launch_bounds__(256) void k_commit_pto(const RunConst *__restrict__ rcp, uint32_t b, uint32_t nb, uint32_t vwords) {
    const RunConst &rc = rcp[blockIdx.y];      // one context per grid row (porrt_grow_batch)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t k = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t N = as_global(rc.n_at)[b];
    nb = row_nb(rc, b, nb);
    if (!nb) return;                               // (a row that has stopped: k_row_sched carried n_at on)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t add = 0;
        for (uint32_t w = 0; w < vwords; ++w) add += __popcll(rc.valid_mask[(size_t)b * vwords + w]);
        as_global(rc.n_at)[b + 1] = N + add;
    }
    if (k >= nb || as_global(rc.q_vid)[k] < 0) return;
    const uint32_t cnt = cand_count(rc, b, k);
    const int *cid = rc.cand_id + cand_off(rc, b, k);
    const double *cval = rc.cand_val + cand_val_off(rc, b, k);
    for (uint32_t a = lane; a < cnt; a += 64) {
        if ((int)cval[a] < 0) continue;
        const int j = cid[a];
        const unsigned long long r = as_global(rc.reachB)[j];
        as_global(rc.reachA)[j] = r;
        if (rc.final_flag[j]) atomicOr(&rc.cnt->finality, r & as_global(rc.final_mask)[j]);
    }
}

// The cost of the best solution, on the device (rrt.rs:183-193, 48-61): for every final node the path to the root is
// written into scratch (leaf to root), then summed from the root outwards exactly as the reference folds it, and the
// workgroup keeps the first minimum in final-node order (ascending id, strict `<`).  One workgroup per context; the
// caller falls back to the host walk if the scratch is too small.
__global__ __launch_bounds__(1024) void k_best_cost(const RunConst *__restrict__ rcp, uint32_t nsteps) {
    const RunConst &rc = rcp[blockIdx.y];
    const uint32_t n_nodes = as_global(rc.n_at)[nsteps], cap = rc.bc_cap;
    int *scratch = rc.bc_scratch;
    uint32_t *cursor = rc.bc_cursor;
    __shared__ unsigned long long s_cost[1024];
    __shared__ uint32_t s_id[1024], s_len[1024];
    auto gpar = as_global(rc.parent);
    auto gx = as_global(rc.nx), gy = as_global(rc.ny);
    auto gff = as_global(rc.final_flag);
    const unsigned long long INF_BITS = 0x7FF0000000000000ull;
    unsigned long long bc = INF_BITS;
    uint32_t bid = 0xFFFFFFFFu, blen = 0, over = 0;
    for (uint32_t f = threadIdx.x; f < n_nodes; f += 1024u) {          // ascending ids per thread
        if (!gff[f]) continue;
        uint32_t L = 0;
        for (int p = (int)f; p >= 0; p = gpar[p]) { ++L; if (L > n_nodes) break; }
        const uint32_t off = atomicAdd(cursor, L);
        if (L > n_nodes || off + L > cap) { over = 1; continue; }
        uint32_t q = off + L;
        for (int p = (int)f; p >= 0; p = gpar[p]) scratch[--q] = p;     // scratch[off] = root ... scratch[off + L - 1] = f
        double sum = 0.0;
        for (uint32_t a = 0; a + 1 < L; ++a) {
            const int u = scratch[off + a], v = scratch[off + a + 1];
            sum += sqrt(dist2(gx[u], gy[u], gx[v], gy[v]));
        }
        const unsigned long long sb = f64_bits(sum);
        if (sb < bc) { bc = sb; bid = f; blen = L; }                    // costs are >= 0: the bit patterns order like the values
    }
    s_cost[threadIdx.x] = bc; s_id[threadIdx.x] = bid; s_len[threadIdx.x] = blen;
    const int any_over = __syncthreads_or((int)over);
    for (uint32_t st = 512; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            const unsigned long long oc = s_cost[threadIdx.x + st];
            const uint32_t oi = s_id[threadIdx.x + st];
            if (oc < s_cost[threadIdx.x] || (oc == s_cost[threadIdx.x] && oi < s_id[threadIdx.x])) {
                s_cost[threadIdx.x] = oc; s_id[threadIdx.x] = oi; s_len[threadIdx.x] = s_len[threadIdx.x + st];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        BestCost r;
        r.cost_bits = s_cost[0]; r.final_id = s_id[0]; r.path_len = s_len[0]; r.overflow = any_over ? 1u : 0u; r.pad = 0;
        *rc.bc_out = r;
    }
}

// ---- the rows' own step schedules (porrt_grow_batch with per-member n_iter_min / n_iter_max and the reference's loop condition)
// The plan of a row, a function of (n_iter_min, n_iter_max, K) alone: steps of K iterations up to n_iter_min (the last one
// shorter), where the condition `i < n_iter_min || (no solution && i < n_iter_max)` (rrt.rs:109, pto.rs:67) is first looked at,
// then steps of K up to n_iter_max with the condition looked at after each -- the batched contract of DESIGN.md section 3.
__global__ __launch_bounds__(256) void k_sched_init(const RunConst *__restrict__ rcp) {
    const RunConst &rc = rcp[blockIdx.y];
    if (!rc.sched_nb) return;
    const uint32_t K = rc.sched_K, mn = rc.sched_min, mx = rc.sched_max, s1 = (mn + K - 1u) / K;
    for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b < rc.sched_steps; b += gridDim.x * 256u) {
        uint32_t i0, nb;
        if (b < s1) { i0 = b * K; nb = mn - i0 < K ? mn - i0 : K; }
        else {
            const unsigned long long at = (unsigned long long)mn + (unsigned long long)(b - s1) * K;
            i0 = at < mx ? (uint32_t)at : mx;
            nb = i0 < mx ? (mx - i0 < K ? mx - i0 : K) : 0u;
        }
        rc.sched_i0[b] = i0; rc.sched_nb[b] = nb;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { rc.cnt->sched_stop = 0xFFFFFFFFu; rc.cnt->sched_iter = 0; }
}

// Before step b: does row q still run?  One thread per row.  A row that has ended keeps sched_nb[b] = 0 for every later step (all
// the step's kernels then leave it alone) and its tree size is carried along n_at, so that the launch's last step names every
// row's final size.  active[b] counts the rows that run step b (the host stops launching when it reads 0).
__global__ __launch_bounds__(64) void k_row_sched(const RunConst *__restrict__ rcp, uint32_t Q, uint32_t b, uint32_t *__restrict__ active) {
    const uint32_t q = blockIdx.x * 64u + threadIdx.x;
    bool runs = false;
    if (q < Q) {
        const RunConst &rc = rcp[q];
        if (rc.sched_nb) {
            Counters *c = rc.cnt;
            bool stop = c->sched_stop != 0xFFFFFFFFu;
            if (!stop) {
                const uint32_t i = rc.sched_i0[b], nbp = rc.sched_nb[b];
                const bool solved = rc.mode == 1 ? (c->n_final > 0 && (c->finality & rc.all_worlds) == rc.all_worlds) : c->n_final > 0;
                stop = nbp == 0u || (i >= rc.sched_min && solved);              // i >= n_iter_max  ||  (i >= n_iter_min && solved)
                if (stop) { c->sched_stop = b; c->sched_iter = i; }
            }
            if (stop) {
                rc.sched_nb[b] = 0;
                rc.n_at[b + 1] = rc.n_at[b];
            }
            runs = !stop;
        } else {
            runs = true;
        }
    }
    const unsigned long long m = __ballot(runs);
    if (threadIdx.x == 0 && m) atomicAdd(&active[b], (uint32_t)__popcll(m));
}

// porrt_grow_batch: every member's counters and final tree size into one array, so that the host needs one copy
// A batch whose rows end at different steps (k_row_sched): the rows that still have work at step b, in their order, then rows that
// have none as padding up to `slots` -- one workgroup.  A row has work at step b while it runs, and at the step after its last one
// (sched_stop == b: its rewire commit rides in that step's search kernel).  k_rows_gather copies the rows' run constants into
// the array the step kernels are launched on from then on, so that their grids shrink with the batch.
__global__ __launch_bounds__(1024) void k_rows_compact(const RunConst *__restrict__ rcp, uint32_t Q, uint32_t b, uint32_t slots, uint32_t *__restrict__ live_idx) {
    __shared__ uint32_t s_wsum[16], s_base, s_dead;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_base = 0; s_dead = 0xFFFFFFFFu; }
    __syncthreads();
    for (uint32_t q0 = 0; q0 < Q; q0 += 1024u) {
        const uint32_t q = q0 + threadIdx.x;
        bool keep = false;
        if (q < Q) {
            const uint32_t stop = rcp[q].cnt->sched_stop;
            keep = stop == 0xFFFFFFFFu || stop >= b;
            if (!keep) atomicMin(&s_dead, q);
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wsum[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t off = s_base;
        for (uint32_t w = 0; w < wv; ++w) off += s_wsum[w];
        if (keep) { const uint32_t pos = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); if (pos < slots) live_idx[pos] = q; }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t w = 0; w < 16u; ++w) t += s_wsum[w]; s_base += t; }
        __syncthreads();
    }
    // padding: a row without work (there is one whenever fewer rows are kept than the batch holds); a full array needs none
    const uint32_t kept = s_base, dead = s_dead;
    // (the host sizes `slots` from a count of running rows that is two steps old, and rows only end: more rows with work than slots
    // cannot happen -- if it did, a row would silently miss its steps, so it is an error of the batch instead)
    if (kept > slots && threadIdx.x == 0) atomicOr(&rcp[0].cnt->err, (uint32_t)ERR_PAGE_OVERFLOW);
    for (uint32_t i = kept + threadIdx.x; i < slots; i += 1024u) live_idx[i] = dead != 0xFFFFFFFFu ? dead : 0u;
}
__global__ __launch_bounds__(256) void k_rows_gather(const RunConst *__restrict__ rcp, const uint32_t *__restrict__ live_idx, RunConst *__restrict__ out) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(rcp + live_idx[blockIdx.x]);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + blockIdx.x);
    static_assert(sizeof(RunConst) % 4 == 0, "copied by words");
    for (uint32_t w = threadIdx.x; w < sizeof(RunConst) / 4u; w += 256u) dst[w] = src[w];
}

// porrt_get_trees into arrays the caller has pinned (porrt_host_pin): one descriptor per tree, grid.y = trees; the kernel reads the
// node arrays and writes the caller's layout over the link -- xy interleaved, the parent widened to 64 bits, dist_root (32 B per node,
// every store a full line of a wave).
struct TreeOut {
    const double *nx, *ny, *dist;
    const int *parent;
    double *oxy, *odist;       // host addresses as the device sees them (or null)
    long long *oparent;
    unsigned long long n;
};
__global__ __launch_bounds__(256) void k_trees_out(const TreeOut *__restrict__ trees) {
    const TreeOut t = trees[blockIdx.y];
    for (unsigned long long j = (unsigned long long)blockIdx.x * 256u + threadIdx.x; j < t.n; j += (unsigned long long)gridDim.x * 256u) {
        if (t.oxy) { dbl2 v; v.x = as_global(t.nx)[j]; v.y = as_global(t.ny)[j]; reinterpret_cast<dbl2 *>(t.oxy)[j] = v; }
        if (t.oparent) t.oparent[j] = (long long)as_global(t.parent)[j];
        if (t.odist) t.odist[j] = as_global(t.dist)[j];
    }
}

struct BatchOut { Counters cnt; uint32_t nodes, pad; };
__global__ __launch_bounds__(64) void k_batch_gather(const RunConst *__restrict__ rcp, uint32_t steps, BatchOut *__restrict__ out) {
    if (threadIdx.x) return;
    const RunConst &rc = rcp[blockIdx.x];
    BatchOut o;
    o.cnt = *rc.cnt;
    o.nodes = rc.n_at[steps];
    o.pad = 0;
    out[blockIdx.x] = o;
}

} // namespace porrt
