/*
 * porrt_oracle.h -- CPU ORACLE for the po-rrt grow/extend hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check
 * in __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The
 * product path (po_rrt_amd/, include/porrt_hip.h) never links or calls it.
 *
 * It is a plain-C restatement of the reference algorithm (cambyse/po-rrt, Rust).
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference).  The Rust reference cannot be built in this environment (no
 * rustc/cargo, crates not vendored) and its maps are Git-LFS pointers, so the
 * oracle is pinned by the reference's data-free unit tests restated as
 * known-answer tests (tests/test_oracle_kat.py):
 *   - kd-tree shape / NN / radius / filtered NN  src/nearest_neighbor.rs:142-311
 *   - Reachability propagation and completeness    src/pto_reachability.rs:109-230
 *   - SquareGoal membership / goal_example         src/common.rs:401-411
 * Two third-party pieces are restated from their published algorithms and are
 * PARITY UNPINNED (nothing under /root/reference pins them):
 *   - rand 0.8 / rand_pcg 0.3 Pcg64 stream and gen_range  (pcg.c)
 *   - line_drawing 0.8 Bresenham pixel walk               (domain.c)
 *
 * Two growth algorithms are provided for both planners:
 *   algo 0 "ref_seq"     literal sequential restatement incl. the Box-linked
 *                        kd-tree and its traversal order (rrt.rs:102-174,
 *                        pto.rs:55-139).
 *   algo 1 "ref_batched" the batched contract the GPU engine implements: K
 *                        samples per step evaluated against the tree snapshot at
 *                        step start, brute-force neighbour search with
 *                        lowest-id tie-break, commit in sample order.
 *                        ref_batched(K=1) == ref_seq bit for bit (tested).
 */
#ifndef PORRT_ORACLE_H
#define PORRT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ pcg.c */
typedef struct {
    unsigned __int128 state;
    unsigned __int128 inc;
} orc_pcg64;

void     orc_pcg64_new(orc_pcg64 *r, unsigned __int128 state, unsigned __int128 stream);
void     orc_pcg64_from_seed(orc_pcg64 *r, const uint8_t seed[32]);
void     orc_pcg64_seed_from_u64(orc_pcg64 *r, uint64_t seed);
uint64_t orc_pcg64_next_u64(orc_pcg64 *r);
void     orc_pcg64_advance(orc_pcg64 *r, unsigned __int128 delta);
double   orc_gen_range_f64(orc_pcg64 *r, double low, double high);
uint64_t orc_gen_range_usize(orc_pcg64 *r, uint64_t low, uint64_t high);
/* flat helpers for ctypes (no __int128 in the signature) */
void     orc_pcg64_new_u64(orc_pcg64 *r, uint64_t state_lo, uint64_t state_hi, uint64_t stream_lo, uint64_t stream_hi);
void     orc_pcg64_get_state(const orc_pcg64 *r, uint64_t out4[4]);

/* --------------------------------------------------------------- domain.c */
double orc_norm1(const double a[2], const double b[2]);
double orc_norm2(const double a[2], const double b[2]);
void   orc_steer(const double from[2], double to[2], double max_step);
double orc_heuristic_radius(uint64_t n_nodes, double max_step, double search_radius, uint64_t dim);
uint32_t orc_f64_as_u32(double v); /* Rust `as u32` */

/* Bresenham as line_drawing 0.8: fills out_xy with (x,y) pairs, returns count
 * (or the needed count if cap is too small). */
size_t orc_bresenham(int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t *out_xy, size_t cap);

enum { ORC_DOMAIN_SHELF = 0, ORC_DOMAIN_DOOR = 1 };
/* traversed-space classes shared by both domains */
enum {
    ORC_FREE = 0,
    ORC_LOW_OBSTACLE = 1,   /* shelf domain only */
    ORC_HIGH_OBSTACLE = 2,  /* shelf HighObstacle / door Obstacle */
    ORC_ZONE_BASE = 16      /* door domain: ORC_ZONE_BASE + zone index */
};

typedef struct orc_ctx orc_ctx;

orc_ctx *orc_create(void);
void     orc_destroy(orc_ctx *c);
const char *orc_last_error(const orc_ctx *c);

int orc_set_grid(orc_ctx *c, const uint8_t *occ, uint32_t W, uint32_t H, const double low[2], const double up[2], int domain);
int orc_set_zones(orc_ctx *c, const uint8_t *zone_ids, double visibility);
int orc_set_sampler(orc_ctx *c, const double low[2], const double up[2], uint64_t seed);
int orc_set_discrete_seed(orc_ctx *c, uint64_t seed);
int orc_set_samples(orc_ctx *c, const double *xy, size_t n);
int orc_set_worlds(orc_ctx *c, const uint32_t *worlds, size_t n);
int orc_set_square_goal(orc_ctx *c, const double *centers, const uint64_t *masks, uint32_t G, double l1_radius);
int orc_set_observation_goal(orc_ctx *c, uint32_t zone_id);

/* domain queries (for tests) */
int  orc_to_pixel(const orc_ctx *c, const double xy[2], uint32_t ij[2]);
int  orc_state_class(const orc_ctx *c, const double xy[2]);            /* ORC_* class, -1 if out of raster */
int  orc_traversed_class(const orc_ctx *c, const double a[2], const double b[2]);
int  orc_n_zones(const orc_ctx *c);
int  orc_n_worlds(const orc_ctx *c);
int  orc_n_validities(const orc_ctx *c);
int  orc_get_validities(const orc_ctx *c, uint64_t *out);
int  orc_get_zone_positions(const orc_ctx *c, double *xy);
int  orc_goal(const orc_ctx *c, const double xy[2], uint64_t *mask);   /* 1 if goal, mask filled */
int  orc_goal_example(const orc_ctx *c, uint32_t world, double xy[2]);
int  orc_sample(orc_ctx *c, double xy[2]);                            /* ContinuousSampler::sample */
uint64_t orc_sample_discrete(orc_ctx *c, uint64_t n);

/* ----------------------------------------------------------------- grow.c */
enum { ORC_MODE_RRT = 0, ORC_MODE_PTO = 1 };
enum { ORC_ALGO_SEQ = 0, ORC_ALGO_BATCHED = 1, ORC_ALGO_BATCHED_KD = 2 };

/* returns 0 on success; PTO mode returns 1 when the final set is incomplete
 * (pto.rs:137) -- outputs are still valid.  Negative = error (see last_error). */
int orc_grow(orc_ctx *c, const double start[2], double max_step, double search_radius,
             uint64_t n_iter_min, uint64_t n_iter_max, uint32_t batch_K, int mode, int algo);

uint64_t orc_num_nodes(const orc_ctx *c);
uint64_t orc_num_iterations(const orc_ctx *c);
int      orc_get_tree(const orc_ctx *c, double *xy, int64_t *parent, double *dist_root);
uint64_t orc_num_final(const orc_ctx *c);
int      orc_get_final_ids(const orc_ctx *c, uint64_t *ids);
int      orc_get_final_masks(const orc_ctx *c, uint64_t *masks);
int      orc_get_reach(const orc_ctx *c, uint64_t *masks);
int      orc_get_node_validity(const orc_ctx *c, uint32_t *validity_ids);
uint64_t orc_num_edges(const orc_ctx *c);           /* forward edges (nbr -> new) in creation order */
int      orc_get_edges(const orc_ctx *c, uint32_t *from, uint32_t *to, uint32_t *validity_id);
int      orc_is_final_set_complete(const orc_ctx *c);
/* host side of RRT::plan (rrt.rs:183-193, 48-61, 223-227): best path to a final
 * node; returns path length (number of states) or 0 when there is no solution;
 * path_xy may be NULL to query the length. */
uint64_t orc_best_solution(const orc_ctx *c, double *path_xy, uint64_t cap, double *cost);
/* rrt.rs:229-246 */
uint64_t orc_firstly_final_ids(const orc_ctx *c, uint64_t *ids, uint64_t cap);

/* --------------------------------------------------------------- kdtree.c */
typedef struct orc_kdtree orc_kdtree;
orc_kdtree *orc_kd_new(const double root[2], uint64_t root_id);
void        orc_kd_free(orc_kdtree *t);
void        orc_kd_add(orc_kdtree *t, const double s[2], uint64_t id);
/* filter: NULL = accept all, else accept iff bit `world` of reach[id] is set */
uint64_t    orc_kd_nearest(const orc_kdtree *t, const double q[2], const uint64_t *reach, uint32_t world);
/* excluded-id-list variant used by the KATs (nearest_neighbor.rs:267-311) */
uint64_t    orc_kd_nearest_excluding(const orc_kdtree *t, const double q[2], const uint64_t *excl, size_t n_excl);
size_t      orc_kd_radius(const orc_kdtree *t, const double q[2], double radius, uint64_t *out_ids, size_t cap);
int         orc_kd_preorder_less(const orc_kdtree *t, uint64_t id_u, uint64_t id_v);
/* structure probes for the KATs: child id or -1 */
int64_t     orc_kd_child(const orc_kdtree *t, uint64_t node_id, int right);
int         orc_kd_state(const orc_kdtree *t, uint64_t node_id, double s[2]);

/* ---------------------------------------------------------------- reach.c */
typedef struct orc_reach orc_reach;
orc_reach *orc_reach_new(void);
void       orc_reach_free(orc_reach *r);
void       orc_reach_set_root(orc_reach *r, uint64_t validity, uint32_t n_worlds);
void       orc_reach_add_node(orc_reach *r, uint64_t validity);
void       orc_reach_add_final_node(orc_reach *r, uint64_t id, uint64_t finality);
void       orc_reach_add_edge(orc_reach *r, uint64_t from, uint64_t to, uint64_t edge_validity);
uint64_t   orc_reach_get(const orc_reach *r, uint64_t id);
int        orc_reach_is_final_set_complete(orc_reach *r);
size_t     orc_reach_final_nodes_for_world(const orc_reach *r, uint32_t world, uint64_t *out, size_t cap);

/* PRM::init + PRM::grow_graph (prm.rs:33-109); results through the PTO getters (nodes, forward edges) */
int orc_prm_grow(orc_ctx *c, const double start[2], double max_step, double search_radius, uint64_t n_iter);
/* PRM::plan_path (prm.rs:111-123; dijkstra / extract_path pto_graph.rs:275-326) on that roadmap */
int64_t orc_prm_plan_path(orc_ctx *c, const double start[2], const double goal[2], double *path_xy, uint64_t cap);

/* ------------------------------------------------------------------ belief.c
 * PTO::build_belief_graph (pto.rs:185-259) and the belief-state functions it calls */
uint64_t orc_belief_hash(const double *bs, uint32_t n);                                     /* common.rs:352-355 */
int      orc_belief_successors(const orc_ctx *c, const double *belief, uint32_t zone, double *out /* 2*n_worlds */);
int      orc_zone_observable(const orc_ctx *c, const double xy[2], uint32_t zone);
int64_t  orc_observe(const orc_ctx *c, const double xy[2], const double *belief, double *out, size_t cap_vectors);
int64_t  orc_reachable_beliefs(const orc_ctx *c, const double *start, double *out /* may be NULL */, size_t cap_vectors);
int      orc_build_belief_graph(orc_ctx *c, const double *start_belief);
void     orc_bg_release(orc_ctx *c);
uint64_t orc_bg_num_beliefs(const orc_ctx *c);
uint64_t orc_bg_num_nodes(const orc_ctx *c);      /* n_nodes * n_beliefs; belief node id = node * n_beliefs + belief */
uint64_t orc_bg_num_edges(const orc_ctx *c);
int      orc_bg_get_beliefs(const orc_ctx *c, double *out);
int      orc_bg_get_types(const orc_ctx *c, uint8_t *out);      /* 0 Unknown, 1 Action, 2 Observation */
int      orc_bg_get_children(const orc_ctx *c, uint64_t *off, uint32_t *ids);
int      orc_bg_get_parents(const orc_ctx *c, uint64_t *off, uint32_t *ids);

/* ------------------------------------------------------------------ dp.c
 * conditional_dijkstra / extract_policy (belief_graph.rs:89-263) on explicit graphs and on the context's belief graph */
int      orc_conditional_dijkstra(uint64_t n, const double *xy, const uint32_t *belief_vec /* row of `beliefs` per node */, const double *beliefs, uint32_t nw,
                                  const uint8_t *types, const uint64_t *coff, const uint32_t *cid, const uint64_t *poff, const uint32_t *pid,
                                  const uint64_t *finals, uint64_t n_final, double *dist);
int64_t  orc_extract_policy(uint64_t n, const double *xy, const uint32_t *belief_id, const uint32_t *belief_vec, const double *beliefs, uint32_t nw,
                            const uint64_t *coff, const uint32_t *cid, const double *dist,
                            uint64_t *original_id, int64_t *parent, uint8_t *is_leaf, uint64_t cap);
int      orc_bg_expected_costs(const orc_ctx *c, double *dist);           /* PTO::compute_expected_costs_to_goals pto.rs:261-275 */
int64_t  orc_bg_extract_policy(const orc_ctx *c, const double *dist, uint64_t *original_id, int64_t *parent, uint8_t *is_leaf, uint64_t cap);

#ifdef __cplusplus
}
#endif
/* ---- multi-modal PRM growth (mmprm.c): MapShelfDomainTampPRM::grow_mm_prm, src/map_shelves_tamp_prm.rs:328-393.  Uses the
 * context's continuous sampler state (cloned into every mode, not advanced) and advances its discrete sampler. */
typedef struct orc_mm orc_mm;
orc_mm  *orc_mm_prm_grow(orc_ctx *c, const double start[2], const double *initial_belief, double max_step, double search_radius, uint64_t n_iter_per_belief);
void     orc_mm_free(orc_mm *g);
uint64_t orc_mm_num_modes(const orc_mm *g);
uint64_t orc_mm_num_transitions(const orc_mm *g);
uint64_t orc_mm_num_beliefs(const orc_mm *g);
void     orc_mm_mode_info(const orc_mm *g, uint64_t m, double *belief, double *reach_p, uint64_t *n_nodes, uint64_t *n_edges, uint64_t *n_final);
void     orc_mm_mode_graph(const orc_mm *g, uint64_t m, double *xy, uint64_t *efrom, uint64_t *eto, uint64_t *finals);
void     orc_mm_transition(const orc_mm *g, uint64_t t, uint32_t *zone, uint32_t *from, uint32_t *to, int *observation, uint64_t *n_pairs);
void     orc_mm_transition_pairs(const orc_mm *g, uint64_t t, uint64_t *pairs);

#endif
