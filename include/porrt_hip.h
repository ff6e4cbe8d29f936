/*
 * porrt_hip.h -- C ABI of the MI355X-native batched belief-space RRT expansion
 * engine (libporrt_hip.so).  Plain pointers and sizes only; no torch / HIP types.
 *
 * It is the drop-in for ONE path of cambyse/po-rrt: the grow/extend loop
 *     RRT::grow_tree   src/rrt.rs:102-174  (RRT* with best-parent + rewire)
 *     PTO::grow_graph  src/pto.rs:55-139   (belief-space RRG + Reachability)
 * instantiated for the reference's grid-backed domains
 *     MapShelfDomain   src/map_shelves_io.rs:65-203,459-488 (RTTFuncs adapter
 *                      src/map_shelves_tamp_rrt.rs:35-47)
 *     Map (doors)      src/map_io.rs:67-241,482-513
 * and its declarative goals (SquareGoal src/common.rs:304-350, ObservationGoal
 * src/rrt.rs:325-341).  The reference's trait callbacks (RTTFuncs rrt.rs:64-76,
 * PTOFuncs pto_graph.rs:121-168, GoalFuncs common.rs:294-302) are opaque host
 * closures, so the engine takes their DATA (grid, zones, goal table, sampler box
 * and seed) instead of their code.  The style follows the reference's own C
 * boundary (src/pto_c.rs:63-270: opaque handle, setters, plan, getters) without
 * callbacks and without taking ownership of caller memory.
 *
 * Conventions: every call returns 0 on success or a negative PORRT_ERR_* code
 * (porrt_last_error() gives the text); nothing panics across the boundary; the
 * caller owns all host buffers; one context drives one GPU and is not
 * thread-safe (it mirrors `&mut self` of RRT::plan, rrt.rs:88); sampler state
 * persists across porrt_grow calls on one context (the reference reuses one RRT
 * object for many plans, src/map_shelves_tamp_rrt.rs:196-232).
 *
 * Limits: states are 2-D (every grid-backed domain of the reference is); at most 64 worlds (one
 * u64 mask per node; the reference's largest case has 16) and 64 goals; batch_K in 1..4096; rasters
 * up to 2^32 pixels; iteration counts below 2^31.  Capacities inside (neighbour lists, edge and
 * deferred-tie pools) grow on demand: the run is replayed from the saved sampler state, results do
 * not depend on it.
 */
#ifndef PORRT_HIP_H
#define PORRT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct porrt_ctx porrt_ctx;

enum {
    PORRT_OK = 0,
    PORRT_INCOMPLETE = 1,           /* PTO: "final nodes are not reached for each world" (pto.rs:137) */
    PORRT_ERR_INVALID = -1,         /* bad argument / call order */
    PORRT_ERR_INVALID_START = -2,   /* pto.rs:61 expect("Start from a valid state!") */
    PORRT_ERR_RASTER = -3,          /* pixel access outside the map, door pixel without zone id, or a
                                       segment crossing two zones: the reference panics there
                                       (image get_pixel; map_io.rs:233) */
    PORRT_ERR_DEVICE = -4,          /* HIP runtime error */
    PORRT_ERR_CAPACITY = -5,        /* a neighbour list outgrew its capacity even after regrowth */
    PORRT_ERR_NO_DEVICE = -6,
    PORRT_ERR_IO = -7,              /* a file could not be opened / written (the reference panics: "Impossible to open image") */
    PORRT_ERR_PEER = -8,            /* porrt_exchange_best: another rank of the collective failed; no rank went on */
    PORRT_ERR_NOMEM = -9            /* a host allocation failed (std::bad_alloc never crosses the boundary) */
};

enum { PORRT_DOMAIN_SHELF = 0, PORRT_DOMAIN_DOOR = 1 };   /* MapShelfDomain / Map */
enum { PORRT_MODE_RRT = 0, PORRT_MODE_PTO = 1, PORRT_MODE_PRM = 2 };   /* RRT::grow_tree / PTO::grow_graph / PRM::grow_graph (results only) */

/* ---- lifetime.  Replaces: RRT::new (rrt.rs:84-86) / PTO::new (pto.rs:37-53) --------- */
porrt_ctx  *porrt_create(int device);      /* NULL when no HIP device is usable */
void        porrt_destroy(porrt_ctx *ctx);
const char *porrt_last_error(const porrt_ctx *ctx);

/* ---- domain data.  Replaces: MapShelfDomain::open/build (map_shelves_io.rs:80-94),
 * Map::open/build (map_io.rs:82-96): ppm = W / (up[0] - low[0]); raster row-major,
 * occ[i*W + j] = img.get_pixel(j, i). */
int porrt_set_grid(porrt_ctx *ctx, const uint8_t *occ, uint32_t W, uint32_t H,
                   const double low[2], const double up[2], int domain);
/* Replaces: add_zones (map_shelves_io.rs:106-148, map_io.rs:113-161): zone raster
 * (255 = none), zone centroids, worlds and world validities are derived inside. */
int porrt_set_zones(porrt_ctx *ctx, const uint8_t *zone_ids, double visibility);

/* ---- samplers.  Replaces: ContinuousSampler::new + DiscreteSampler::new
 * (sample_space.rs:13-21,45-49; the reference hard-codes seed 0).  The stream is
 * rand_pcg Pcg64::seed_from_u64(seed) with rand 0.8 gen_range, generated on the GPU. */
int porrt_set_sampler(porrt_ctx *ctx, const double low[2], const double up[2], uint64_t seed);
int porrt_set_discrete_seed(porrt_ctx *ctx, uint64_t seed);
/* Inject the stream instead: xy[2*i..] is the i-th value ContinuousSampler::sample()
 * would return (goal-biased iterations consume none, rrt.rs:176-181); worlds[i] is
 * the i-th DiscreteSampler::sample() value (one per iteration, pto.rs:142). */
int porrt_set_samples(porrt_ctx *ctx, const double *xy, size_t n);
int porrt_set_worlds(porrt_ctx *ctx, const uint32_t *worlds, size_t n);

/* ---- goals.  Replaces: SquareGoal::new (common.rs:310-333); bit w of masks[g] is
 * world w of the goal's validity.  ObservationGoal (rrt.rs:325-341). */
int porrt_set_square_goal(porrt_ctx *ctx, const double *centers /* G*2 */, const uint64_t *masks,
                          uint32_t G, double l1_radius);
int porrt_set_observation_goal(porrt_ctx *ctx, uint32_t zone_id);

/* ---- the hot path.  Replaces: RRT::grow_tree (rrt.rs:102-174) when mode is
 * PORRT_MODE_RRT and PTO::grow_graph (pto.rs:55-139) when PORRT_MODE_PTO.
 * batch_K samples are expanded per step against the tree as it stands at the start
 * of the step; batch_K = 1 is the reference's loop.  Returns PORRT_OK,
 * PORRT_INCOMPLETE (PTO only) or an error. */
int porrt_grow(porrt_ctx *ctx, const double start[2], double max_step, double search_radius,
               uint64_t n_iter_min, uint64_t n_iter_max, uint32_t batch_K, int mode);

/* The same growth for several contexts of ONE device at once -- the form in which the reference's many-query caller uses the path:
 * the TAMP search runs rrt.plan(.., n_iter_min 2500, n_iter_max 10000) twice per search edge (map_shelves_tamp_rrt.rs:224,232,355,
 * 367,492,500; main.rs:532), thousands of independent queries.  Their dependent-load chains overlap inside each kernel launch (one
 * grid row per context) instead of queueing behind each other.  Every context runs the loop of rrt.rs:109 / pto.rs:67,
 *     while i < n_iter_min || (no solution yet && i < n_iter_max),
 * on its own: a member whose loop has ended drops out of the later launches (a row mask kept on the device), the others go on.
 * starts = n_ctx x 2.  Every context keeps its own map, goal, sampler state and results, exactly as after n_ctx porrt_grow calls
 * with the same arguments; the getters are per context.  Contexts must not be used concurrently elsewhere during the call.
 * Returns the worst member code (PORRT_INCOMPLETE if a PTO member's final set is incomplete).
 * porrt_grow_batch_each: the same with n_iter_min[q], n_iter_max[q] per context. */
int porrt_grow_batch(porrt_ctx *const *ctxs, uint32_t n_ctx, const double *starts, double max_step,
                     double search_radius, uint64_t n_iter_min, uint64_t n_iter_max, uint32_t batch_K, int mode);
int porrt_grow_batch_each(porrt_ctx *const *ctxs, uint32_t n_ctx, const double *starts, double max_step,
                          double search_radius, const uint64_t *n_iter_min, const uint64_t *n_iter_max, uint32_t batch_K, int mode);

/* ---- results, copied into caller-owned buffers (query the sizes first).
 * Replaces: RRTTree{nodes: Vec<RRTNode{state,parent_id,dist_from_root}>} (rrt.rs:14-22)
 * and the final ids returned by grow_tree (rrt.rs:103,165-167). */
uint64_t porrt_num_nodes(const porrt_ctx *ctx);
uint64_t porrt_num_iterations(const porrt_ctx *ctx);
int      porrt_get_tree(const porrt_ctx *ctx, double *xy /* N*2 */, int64_t *parent /* -1 = root */,
                        double *dist_root);
/* The same for n contexts of one device at once (the trees of a porrt_grow_batch): worker threads with pinned staging
 * and copy streams of their own overlap the device-to-host copies with laying the trees out in the caller's arrays
 * (xy[q]: N_q*2 doubles, parent[q]: N_q, dist_root[q]: N_q; an array of pointers, or single entries, may be NULL). */
int      porrt_get_trees(porrt_ctx *const *ctxs, uint32_t n_ctx, double *const *xy, int64_t *const *parent,
                         double *const *dist_root);
/* A caller that fetches trees again and again into the same arrays can hand those arrays to the device once: porrt_host_pin page-locks
 * [p, p + bytes) and maps it for the GPU (hipHostRegister), and a porrt_get_trees whose every output array lies inside pinned ranges
 * lets ONE kernel write all the trees straight into the caller's arrays in their final layout -- no staging copies, no host threads
 * re-packing them.  The range must stay allocated until porrt_host_unpin (a freed and re-used address would be written through the
 * old mapping).  Arrays that are not pinned take the staged path above; the results are the same. */
int      porrt_host_pin(void *p, size_t bytes);
int      porrt_host_unpin(void *p);
uint64_t porrt_num_final(const porrt_ctx *ctx);
int      porrt_get_final_ids(const porrt_ctx *ctx, uint64_t *ids);
int      porrt_get_final_masks(const porrt_ctx *ctx, uint64_t *masks);
/* PTO mode.  Replaces: Reachability::reachability (pto_reachability.rs:54-56),
 * PTONode.validity_id and the PTOGraph edges (pto_graph.rs:171-207): forward edges
 * (neighbour -> new node) in the order the reference adds them (pto.rs:103-114): new nodes
 * ascending, and for one new node its neighbours in the order KdTree::nearest_neighbors lists
 * them (kd pre-order, nearest_neighbor.rs:101-117) -- so that adjacency lists rebuilt from it
 * equal the reference's element for element.  The reference also stores each reverse edge with
 * the same validity id (pto.rs:117-120).  The order is restored on the device when the edges are
 * first asked for (the edges are dealt into one bucket per node and each bucket is ordered by one
 * wave, by the nodes' kd pre-order ranks, which the host supplies from a kd-tree over the node
 * coordinates); the growth itself does not need it. */
int      porrt_get_reach(const porrt_ctx *ctx, uint64_t *masks /* N */);
int      porrt_get_node_validity(const porrt_ctx *ctx, uint32_t *validity_ids /* N */);
uint64_t porrt_num_edges(const porrt_ctx *ctx);
int      porrt_get_edges(const porrt_ctx *ctx, uint32_t *from, uint32_t *to, uint32_t *validity_id);
int      porrt_is_final_set_complete(const porrt_ctx *ctx);   /* pto_reachability.rs:81-90 */
/* derived domain data (map_shelves_io.rs:113,132-148; map_io.rs:121-126) */
int      porrt_n_worlds(const porrt_ctx *ctx);
int      porrt_get_validities(const porrt_ctx *ctx, uint64_t *masks /* <= 65 */);
int      porrt_get_zone_positions(const porrt_ctx *ctx, double *xy /* <= 64*2 */);

/* Host side of RRT::plan (get_best_solution rrt.rs:183-193, get_path_to 48-61,
 * get_path_cost 223-227): length of the best path (0 = "No solution found");
 * path_xy may be NULL to query the length. */
uint64_t porrt_best_solution(const porrt_ctx *ctx, double *path_xy, uint64_t cap, double *cost);
/* The cost (and final node) of that path alone, evaluated on the device without fetching the tree: same
 * arithmetic, same first-minimum rule, bit-identical cost.  Lets a caller that holds many trees on the GPU
 * (porrt_grow_batch) pick the one worth downloading.  Returns 1, or 0 for "No solution found". */
int      porrt_best_cost(const porrt_ctx *ctx, double *cost, uint64_t *final_id);
/* The same for the n contexts of the last porrt_grow_batch, in one launch (one workgroup per context);
 * costs[q] = +inf where context q has no solution.  Other sets of contexts are evaluated one after the other. */
int      porrt_best_cost_batch(porrt_ctx *const *ctxs, uint32_t n_ctx, double *costs);

/* ---- PRM* roadmap growth: PRM::init(start) + PRM::grow_graph(max_step, search_radius, n_iter) (src/prm.rs:33-109) on a
 * grid-backed domain.  Every sample becomes a node (the reference neither steers nor checks the state; validity id 0),
 * connected both ways to the earlier nodes within heuristic_radius(graph size) whose transition the domain's
 * transition_validator accepts.  The roadmap depends on the sample stream only, so the whole of it is evaluated at once
 * with the reference's sequential semantics (no batch size).  Samples: the context's sampler (its state moves on by
 * n_iter draws) or the injected stream.  Results through the getters of a PTO graph: porrt_num_nodes (n_iter + 1, node 0
 * = start), porrt_get_tree (coordinates; parents -1), porrt_num_edges / porrt_get_edges (the forward edges neighbour ->
 * new node in the reference's adjacency order; PTOGraph gets add_edge(from, to, 0) and add_edge(to, from, 0), the third
 * array is what transition_validator returned). */
int      porrt_grow_prm(porrt_ctx *ctx, const double start[2], double max_step, double search_radius, uint64_t n_iter);
/* PRM::plan_path (src/prm.rs:111-123) on that roadmap: the kd-tree's nearest nodes of start and goal
 * (nearest_neighbor.rs:48-91), dijkstra from the goal (src/pto_graph.rs:275-303; run as device sweeps to the same
 * fixpoint), extract_path (pto_graph.rs:305-326).  Returns the number of states of the path (0 = start and goal are not
 * connected: the reference returns an empty Vec); path_xy receives min(cap, that number) states. */
int64_t  porrt_prm_plan_path(porrt_ctx *ctx, const double start[2], const double goal[2], double *path_xy, uint64_t cap);

/* ---- belief-space expansion: PTO::build_belief_graph (src/pto.rs:185-259) on the graph of the last
 * porrt_grow(mode PORRT_MODE_PTO) of this context, with PTOFuncs::reachable_belief_states (map_io.rs:515-546,
 * map_shelves_io.rs:490-520), PTOFuncs::observe (map_io.rs:281-300, map_shelves_io.rs:242-265) and
 * compute_compatibility (common.rs:266-276).  start_belief: one probability per world, summing to 1
 * (assert_belief_state_validity, common.rs:279-281).
 * Belief node id = graph node * n_beliefs + belief id -- the order of BeliefGraph::add_node at pto.rs:198-201;
 * node_to_belief_nodes[node][belief] is Some(that id) iff the belief is compatible with the node's validity, i.e. iff
 * the pair can have edges.  Children / parents come back as CSR lists in the reference's Vec::push order.
 * The lists stay on the device for the rows that follow; the getters copy them out.
 * Errors: the panics of the path ("no id corresponding to this belief state", hash collisions, raster faults in the
 * visibility raycast) become negative return codes. */
int      porrt_build_belief_graph(porrt_ctx *ctx, const double *start_belief, uint32_t n_worlds);
uint64_t porrt_bg_num_beliefs(const porrt_ctx *ctx);                 /* reachable_belief_states().len() */
uint64_t porrt_bg_num_nodes(const porrt_ctx *ctx);                   /* n graph nodes * n beliefs */
uint64_t porrt_bg_num_edges(const porrt_ctx *ctx);
int      porrt_bg_get_beliefs(const porrt_ctx *ctx, double *out /* n_beliefs * n_worlds */);
int      porrt_bg_get_observable_zones(const porrt_ctx *ctx, uint64_t *masks /* per graph node: bit z = zone z seen */);
int      porrt_bg_get_node_types(const porrt_ctx *ctx, uint8_t *types /* 0 Unknown, 1 Action, 2 Observation (belief_graph.rs:13-17) */);
int      porrt_bg_get_children(const porrt_ctx *ctx, uint64_t *off /* n_nodes + 1 */, uint32_t *ids /* n_edges, may be NULL */);
int      porrt_bg_get_parents(const porrt_ctx *ctx, uint64_t *off, uint32_t *ids);
/* seconds of the last build: [0] total, [1] device kernels (HIP events), [2] host tables, [3] reachable beliefs,
 * [4] observation fold table, [5] adjacency lists, [6] device allocation + uploads, [7] fetching the PTO edges */
int      porrt_bg_get_seconds(const porrt_ctx *ctx, double *out, uint32_t n);

/* ---- expected costs to the goals over the belief graph: PTO::compute_expected_costs_to_goals (src/pto.rs:261-275) =
 * the final belief nodes (final graph nodes x beliefs compatible with the node and with its finality), then
 * conditional_dijkstra (src/belief_graph.rs:89-175) with cost_evaluator = norm2 (the PTOFuncs default,
 * pto_graph.rs:150-152).  Runs on the belief graph of the last porrt_build_belief_graph, on the device; the costs stay
 * there.  Bit-identical to the reference's queue-driven loop (the relaxation is monotone in f64: every order ends in
 * the same fixpoint).  +inf = no policy from that belief node. */
int      porrt_bg_compute_expected_costs(porrt_ctx *ctx);
int      porrt_bg_get_expected_costs(const porrt_ctx *ctx, double *out /* porrt_bg_num_nodes() */);
int      porrt_bg_expected_cost_of(const porrt_ctx *ctx, uint64_t belief_node, double *out);   /* [0] = policy.expected_costs */
int      porrt_bg_get_dp_info(const porrt_ctx *ctx, double *total_s, double *device_s, uint32_t *sweeps);
/* belief nodes the sweeps of the last run passed over, summed over the sweeps (a sweep reads and writes a level's change flags and the
 * costs of the rows that changed: the unit of that row's roofline in bench.py) */
uint64_t porrt_bg_get_dp_sweep_rows(const porrt_ctx *ctx);
/* PTO::extract_policy (src/pto.rs:277-283; extract_policy / get_best_expected_children src/belief_graph.rs:177-263) from
 * belief node 0.  Policy node k (in Policy::add_node order; node 0 is the root) has original_node_id original_ids[k],
 * parent parents[k] (-1 for the root) and is a leaf (expected cost 0) iff is_leaf[k]; its state and belief follow from
 * the id (graph node id / n_beliefs, belief id % n_beliefs).  Returns the number of policy nodes; the arrays are filled
 * when cap holds them (call with cap 0 to size them).  *expected_costs = policy.expected_costs.  The walk is
 * sequential and small: host code, reading one row of the device graph per step.  Error when the root has no finite
 * expected cost (the reference would not terminate). */
int64_t  porrt_bg_extract_policy(porrt_ctx *ctx, uint64_t *original_ids, int64_t *parents, uint8_t *is_leaf, uint64_t cap, double *expected_costs);
/* conditional_dijkstra on an explicit belief graph given as host arrays (the form of the reference's own tests,
 * belief_graph.rs:502-567): node i has state xy[2i..], belief vector beliefs[belief_row[i]], type types[i]
 * (1 Action, 2 Observation), children / parents as CSR in add_edge order.  No context needed. */
int      porrt_conditional_dijkstra(int device, uint64_t n, const double *xy, const uint32_t *belief_row, const double *beliefs,
                                    uint32_t n_belief_rows, uint32_t n_worlds, const uint8_t *types,
                                    const uint64_t *child_off, const uint32_t *child_ids, const uint64_t *parent_off, const uint32_t *parent_ids,
                                    const uint64_t *finals, uint64_t n_final, double *dist);

/* ---- measurement */
typedef struct {
    uint64_t n_iter;          /* iterations run */
    uint64_t n_nodes;         /* tree size incl. root */
    uint64_t n_steps;         /* batched steps launched */
    uint64_t n_tie_fallbacks; /* equal-cost parent ties not covered by the goal-path rule (expected 0) */
    double   total_s;         /* wall time of porrt_grow */
    double   setup_s;         /* part of total_s: host tables + uploads (radius table, worlds) */
    double   device_s;        /* HIP-event time from first to last kernel of the growth loop */
    double   scan_s;          /* HIP-event time summed over the near-search kernel (NN + steer + radius search;
                                 only filled when profiling is enabled) */
    uint64_t scan_launches;
    double   scan_pairs;      /* sample x node pairs those searches answer (2 * K * N_b per step) */
    double   scan_bytes;      /* algorithmic bytes of those searches */
    double   connect_s;       /* HIP-event time summed over the connect kernel (profiling only) */
} porrt_metrics;
int porrt_get_metrics(const porrt_ctx *ctx, porrt_metrics *out);
/* options: "profile" (0/1 per-kernel HIP events), "cand_cap" (initial neighbour-list
 * capacity per sample), "graph" (0/1 replay the growth loop as a hipGraph), "kd_group" (steps whose
 * new nodes enter the tie-order structure together; 0 = chosen from batch_K), "group_lanes" (RRT*
 * step kernels: 16 / 32 / 64 lanes per sample, 0 = one wave per sample, -1 = chosen from the number
 * of contexts advanced together), "batch_streams" (on the FIRST context of a porrt_grow_batch: the
 * sub-batches it advances side by side, each on its own streams; 0 = 2 from 32 contexts on, else 1),
 * "pipeline" (RRT* steps of the one-wave-per-sample kernels, i.e. of a single query: 0 = search, connect, commit one after the
 * other; 1 = step b + 1 is searched while step b is connected, in one launch, the filing and the rewire commit in a second;
 * 4 (default) = ONE launch per step: the filing of a step's nodes and its rewire commit run beside the next step's kernels
 * (batch_K <= 1024, else as 1); 2 = all steps in one persistent cooperative launch with barriers over the grid (measured: far
 * slower on eight XCDs, kept for the record), "kd_after" (1 = the tie-order structure is built after the last step instead of
 * beside the steps: measured slower), "kd_lazy" (RRT*, batches and the default form of a single query: 1 (default; 2 is accepted and means the same) =
 * beside the steps only the goal path of the reference's kd-tree is kept -- it orders every tie between copies of the goal point and
 * their parent -- and the whole structure is built after the steps in the rare run where two other nodes tie; 0 = the whole structure
 * beside the steps on a second stream), "kd_claim_threads", "kd_ride", "kd_inline", "early_wave_steps", "dp_sweeps", "compact_rows" (1, default: a
 * porrt_grow_batch whose members end at different steps launches its later steps on the members that still have work), "gtrack_side" (1 = a single query's goal-path workgroup as a kernel
 * of its own on the side stream; measured slower, 0 is the default: a workgroup of the step kernel), "box_table"
 * (1, default: the group and roadmap kernels answer "is this segment free" from a summed-area table of the raster when the bounding
 * box of its end pixels holds free pixels only, and walk it otherwise; 0 = always walk).  None of them changes a result. */
int porrt_set_option(porrt_ctx *ctx, const char *name, int64_t value);
/* what was in force: "launch_mode" (the last porrt_grow_batch led by this context: 0 = one launch sequence, G = G sequences side by
 * side on streams chosen by measurement, -G = G sequences on the contexts' own streams -- the probe found no parallel set, e.g. under a
 * profiler that serialises kernels), "pipeline", "group_lanes", "kd_lazy", "kd_built_after" (1: a tie of the last grow -- of the batch
 * this context led -- needed the whole kd structure, which was built after its steps), "kd_lca_steps" (this context's own need: 1 + the
 * last step with a tie that took that structure, 0 = none), "compactions" (how often the last batch this context led gathered the
 * members still running) */
int porrt_get_option(const porrt_ctx *ctx, const char *name, int64_t *value);

/* Device arithmetic self-test: sqrt and divide of n doubles on the GPU versus the host's correctly
 * rounded results; both mismatch counts must be 0 for bit-exact parity (rrt.rs costs, common.rs:218). */
int porrt_selftest(porrt_ctx *ctx, uint64_t n, uint64_t *sqrt_mismatch, uint64_t *div_mismatch);

/* ---- multi-modal PRM growth: MapShelfDomainTampPRM::grow_mm_prm (src/map_shelves_tamp_prm.rs:328-393; ModeTree :135-283,
 * sample_observation_of_zone :482-493) on a shelf domain with zones.  One PRM* roadmap per mode (belief); the reference's loop
 * decides on the host which point goes to which mode in which order (the context's discrete sampler moves on; every mode
 * clones the continuous sampler's current state, which is NOT advanced -- the reference's behaviour; the zone sampler is a
 * fresh seed-0 stream per call), then each mode's roadmap is built at once on the GPU from its ordered points.  Results:
 * per mode its belief, reaching probability, nodes (add_sample order), forward edges (neighbour -> new node, the reference's
 * adjacency order; the reverse edge is implied, prm.rs:96-103) and final node ids; per transition the observed zone, the two
 * modes, the observation flag as the reference stores it and the [node in from-mode, node in to-mode] pairs.  After this
 * call the single-graph getters (porrt_get_tree ...) have no results. */
int      porrt_grow_mm_prm(porrt_ctx *ctx, const double start[2], const double *initial_belief, uint32_t n_worlds, double max_step,
                           double search_radius, uint64_t n_iter_per_belief);
uint64_t porrt_mm_num_modes(const porrt_ctx *ctx);
uint64_t porrt_mm_num_transitions(const porrt_ctx *ctx);
uint64_t porrt_mm_num_beliefs(const porrt_ctx *ctx);          /* reachable belief states of the prior (the sample budget's factor) */
int      porrt_mm_get_mode(const porrt_ctx *ctx, uint64_t mode, double *belief /* n_worlds */, double *reaching_probability, uint64_t *n_nodes,
                           uint64_t *n_edges, uint64_t *n_final);
int      porrt_mm_get_mode_graph(const porrt_ctx *ctx, uint64_t mode, double *xy, uint32_t *edge_from, uint32_t *edge_to, uint64_t *final_ids);
int      porrt_mm_get_transition(const porrt_ctx *ctx, uint64_t t, uint32_t *zone, uint32_t *from_mode, uint32_t *to_mode, int *observation, uint64_t *n_pairs);
int      porrt_mm_get_transition_pairs(const porrt_ctx *ctx, uint64_t t, uint64_t *pairs /* 2 * n_pairs */);
int      porrt_mm_get_seconds(const porrt_ctx *ctx, double *host_s, double *roadmap_s, double *device_s);

/* ---- on-disk formats either side of the path (host code; no GPU needed).
 * porrt_read_pgm: the raster MapShelfDomain::open / Map::open load (image::open -> ImageLuma8,
 * map_shelves_io.rs:88-103, map_io.rs:90-105): P2 / P5 (P1 / P4 as 0 / 255), '#' comments in the header, samples as
 * stored (no rescaling by maxval), row-major from the top -- what porrt_set_grid / porrt_set_zones take.  Files the
 * reference rejects with "Wrong image format!" (maxval > 255 -> 16-bit gray, colour) give PORRT_ERR_INVALID.  Call with
 * out = NULL for the size. */
int porrt_read_pgm(const char *path, uint8_t *out /* W*H or NULL */, uint32_t *W, uint32_t *H);
int porrt_read_pgm_mem(const uint8_t *bytes, size_t n, uint8_t *out, uint32_t *W, uint32_t *H);
/* PTOGraph JSON (pto_graph.rs:22-118 save / load): {"nodes": [{"state", "validity_id", "parents": [{"id",
 * "validity_id"}], "children": [..]}], "validities": [[bool]]} in serde_json's pretty form.  The writer takes CSR
 * adjacency; porrt_graph_save_json writes the graph of the context's last PTO grow / PRM roadmap with the lists as the
 * reference holds them.  The reader hands the file back as the same arrays (query the sizes, then porrt_graph_file_get
 * with caller buffers; any pointer may be NULL). */
typedef struct porrt_graph_file porrt_graph_file;
int porrt_graph_write_json(const char *path, uint64_t n_nodes, const double *xy, const uint64_t *node_validity, const uint64_t *child_off,
                           const uint64_t *child_id, const uint64_t *child_validity, const uint64_t *parent_off, const uint64_t *parent_id,
                           const uint64_t *parent_validity, uint64_t n_validities, uint64_t n_worlds, const uint8_t *validities /* [n_validities][n_worlds] 0/1 */);
int porrt_graph_save_json(const porrt_ctx *ctx, const char *path);
porrt_graph_file *porrt_graph_load_json(const char *path, char *err /* may be NULL */, size_t err_cap);   /* NULL on failure */
void     porrt_graph_file_free(porrt_graph_file *g);
uint64_t porrt_graph_file_num_nodes(const porrt_graph_file *g);
uint64_t porrt_graph_file_num_children(const porrt_graph_file *g);
uint64_t porrt_graph_file_num_parents(const porrt_graph_file *g);
uint64_t porrt_graph_file_num_validities(const porrt_graph_file *g);
uint64_t porrt_graph_file_num_worlds(const porrt_graph_file *g);
int      porrt_graph_file_get(const porrt_graph_file *g, double *xy, uint64_t *node_validity, uint64_t *child_off, uint64_t *child_id,
                              uint64_t *child_validity, uint64_t *parent_off, uint64_t *parent_id, uint64_t *parent_validity, uint8_t *validities);

/* ---- the one exchange of a query-sharded job (the reference is one process and has no
 * counterpart).  Queries are independent: query q runs on rank q mod world, nothing is communicated
 * while trees grow.  At the end, per map: ncclAllGather of one 16-byte entry per rank, the first minimum of
 * (cost, rank) wins, and the winner's node arrays are broadcast device to device (RCCL over xGMI) into
 * buffers the communicator owns on every rank.  Rendezvous is the caller's: rank 0 makes the id, its 128
 * bytes travel by whatever channel the host program has (MPI, a file, torch.distributed ...). */
#define PORRT_UNIQUE_ID_BYTES 128
typedef struct porrt_comm porrt_comm;
typedef struct {
    double  cost;            /* best path cost (RRT::get_best_solution, rrt.rs:183-193); +inf = no solution */
    int32_t rank;            /* owner; -1 in a result = nobody solved this map */
    int32_t n_nodes;         /* size of that tree */
} porrt_best_entry;
int         porrt_comm_unique_id(uint8_t id[PORRT_UNIQUE_ID_BYTES]);
porrt_comm *porrt_comm_create(int device, int rank, int world, const uint8_t id[PORRT_UNIQUE_ID_BYTES]);   /* NULL on failure */
void        porrt_comm_destroy(porrt_comm *comm);
const char *porrt_comm_last_error(const porrt_comm *comm);
/* ctxs[q] (contexts of mode PORRT_MODE_RRT) planned on map map_ids[q] (< n_maps); every rank passes the same n_maps.
 * winners[m] is the same on every rank afterwards.  Collective: all ranks of the communicator must call it -- and a rank
 * whose own part fails (bad argument, a context without a tree, allocation) still takes part: before each data step the
 * ranks all-gather a status word, and either all go on or all return -- the failing rank its own code, the others
 * PORRT_ERR_PEER (porrt_comm_last_error names the rank); ranks called with different n_maps all get PORRT_ERR_INVALID. */
int      porrt_exchange_best(porrt_comm *comm, porrt_ctx *const *ctxs, uint32_t n_ctx, const uint32_t *map_ids,
                             uint32_t n_maps, porrt_best_entry *winners);
/* The collective part alone, for a caller that keeps its own best tree per map: mine[m] = (cost, n_nodes) of this rank's best tree of map m
 * (n_nodes 0 = none; the rank field is filled in), views[m] = where its arrays live (device pointers); the same protocol, results and
 * getters as porrt_exchange_best, which is this after evaluating the contexts. */
struct porrt_tree_device_view_s;
int      porrt_exchange_tables(porrt_comm *comm, const porrt_best_entry *mine, const struct porrt_tree_device_view_s *views,
                               uint32_t n_maps, porrt_best_entry *winners);
/* No rank leaves the sequence alone: once the first agreement is through, a rank whose HIP or RCCL call fails, whose RCCL reports an
 * asynchronous error, or whose step is not finished after the time-out (default 120 s) ABORTS the communicator (ncclCommAbort) before
 * it returns, so that the peers' pending collectives fail instead of waiting; every wait polls the stream and RCCL's error state,
 * none blocks for good.  The communicator is unusable afterwards (every call returns PORRT_ERR_DEVICE): the job makes a new one. */
int      porrt_comm_usable(const porrt_comm *comm);            /* 1, or 0 once a collective step failed on this rank */
int      porrt_comm_set_timeout_ms(porrt_comm *comm, int ms);
/* for the CPU tests of that protocol (no RCCL, no device): a stand-in communicator, and the function every failure of the real
 * path goes through -- stage 0 = a local failure before the first agreement (a status word; returns `code`, the communicator stays
 * usable), stage >= 1 = at or after it (the communicator is aborted); porrt_comm_test_aborts counts the aborts */
porrt_comm *porrt_comm_test_new(int rank, int world);
/* ... and a communicator over a transport the test brings: the exchange's whole sequence (agreements, all-gather, decision, the
 * broadcasts from whichever rank wins, the getters) then runs between CPU processes on host memory -- tests/test_sharding_gloo.py drives
 * it with two ranks over gloo.  Every function returns 0 or an error; "device" buffers are what alloc returns. */
typedef struct {
    void *self;
    int   (*all_gather)(void *self, const void *send, void *recv, size_t bytes_per_rank);     /* recv: world x bytes, rank order */
    int   (*broadcast)(void *self, const void *send, void *recv, size_t bytes, int root);     /* send is read on the root only */
    void *(*alloc)(void *self, size_t bytes);
    void  (*release)(void *self, void *p);
    int   (*fetch)(void *self, void *host_dst, const void *src, size_t bytes);
    int   (*abort)(void *self);                                                               /* may be NULL */
} porrt_comm_ops;
porrt_comm *porrt_comm_test_new_ops(int rank, int world, const porrt_comm_ops *ops);          /* ops must outlive the communicator */
int      porrt_comm_test_fail(porrt_comm *comm, int stage, int code /* < 0 */);
int      porrt_comm_test_aborts(const porrt_comm *comm);
uint64_t porrt_exchange_num_nodes(const porrt_comm *comm, uint32_t map);
int      porrt_exchange_get_tree(const porrt_comm *comm, uint32_t map, double *xy, int64_t *parent, double *dist_root);
/* step 2 of the exchange on its own (pure host code): all[r * n_maps + m] -> win_rank[m] */
int      porrt_exchange_decide(const porrt_best_entry *all, uint32_t world, uint32_t n_maps, int32_t *win_rank);
/* the status decision on its own (pure host code): words[2r] = code of rank r (0 or negative), words[2r+1] = its n_maps;
 * returns what rank my_rank must return (above), *bad_rank = the first failing / disagreeing rank or -1 */
int      porrt_exchange_agree(const int32_t *words /* world x 2 */, uint32_t world, uint32_t my_rank, int32_t *bad_rank);

/* The grown tree where it lives: device pointers into the context's arena (valid until the context's
 * next grow or its destruction).  n_nodes = 0 when there are no results. */
typedef struct porrt_tree_device_view_s {
    const double *nx, *ny, *dist_root;
    const int32_t *parent;       /* -1 = root */
    uint64_t n_nodes;
} porrt_tree_device_view;
porrt_tree_device_view porrt_tree_device(const porrt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
