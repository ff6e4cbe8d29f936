/*
 * po_rrt_c.h -- the C symbols of the reference's own C boundary (cambyse/po-rrt src/pto_c.rs:63-270, built there as the
 * `po_rrt` dylib), served by the MI355X engine (libpo_rrt.so next to libporrt_hip.so).  An existing C caller of libpo_rrt
 * links this library instead and keeps its calls -- with ONE change in kind: the reference takes the planning domain as
 * host callbacks (state / transition validity, observer, goal), and opaque host closures cannot run on a GPU.  The
 * callback setters are kept so that such a caller still links, but they record an error (and return it: the reference
 * declares them void, a caller that ignores the value sees plan() fail instead); the domain is declared by DATA through
 * the po_rrt_* functions at the end, for the reference's two grid-backed domains (MapShelfDomain src/map_shelves_io.rs,
 * Map src/map_io.rs) and its SquareGoal (src/common.rs:304-350).
 *
 * Differences a caller must know (each deliberate):
 *   - plan() does NOT take ownership of `start` (the reference rebuilds a Vec from the raw pointer and frees it,
 *     pto_c.rs:231 -- a double free for any C caller that owns its buffer); nor of any other pointer handed in.
 *   - no panics across the boundary: every function returns 0 or a negative PORRT_ERR_* code (porrt_hip.h), and
 *     po_rrt_last_error() has the text.  The reference aborts ("graph not grown up to solution", pto_c.rs:215).
 *   - samplers: the reference seeds from OS entropy (new_true_random, pto_c.rs:213); so does this library unless
 *     po_rrt_set_seed() is called.
 *   - policy refinement (PTOPolicyRefiner::refine_solution, pto_c.rs:217-218) is outside the accelerated path (SURVEY
 *     section 2: OUT OF SCOPE): the paths are the extracted policy's, refinement_s reads 0, refine_iterations is ignored.
 *   - state_dim must be 2 (all grid-backed domains of the reference are 2-D; the reference also offers 3, 7, 9 for
 *     callback domains), n_worlds at most 64.
 */
#ifndef PO_RRT_C_H
#define PO_RRT_C_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct CPlanningProblem CPlanningProblem;        /* pto_c.rs:28-61, opaque here */

/* callback types of the reference (pto_c.rs:17-23) */
typedef int64_t (*StateValidityCallbackType)(const double *, size_t);
typedef int64_t (*TransitionValidityCallbackType)(const double *, size_t, const double *, size_t);
typedef double (*CostEvaluatorCallbackType)(const double *, size_t, const double *, size_t);
typedef void (*ObserverCallbackType)(const double *, size_t, const double *, size_t, size_t ***, size_t *);
typedef bool (*GoalCallbackType)(const double *, size_t, bool *, size_t);
typedef void (*GoalExampleCallbackType)(size_t, double *, size_t);

CPlanningProblem *new_planning_problem(void);                                                        /* pto_c.rs:63-100 */
void delete_planning_problem(CPlanningProblem *);                                                    /* :102-103 */
int set_problem_dimensions(CPlanningProblem *, size_t state_dim, size_t n_worlds);                   /* :105-111 */
int set_lower_sampling_bound(CPlanningProblem *, double *low, size_t n);                             /* :113-120 */
int set_upper_sampling_bound(CPlanningProblem *, double *up, size_t n);                              /* :122-129 */
int set_world_validities(CPlanningProblem *, size_t **validities, size_t n);                         /* :131-136; derived from the zone raster here, the argument is not read */
int set_state_validity_callback(CPlanningProblem *, StateValidityCallbackType);                      /* :138-143  -> PORRT_ERR_INVALID */
int set_transition_validity_callback(CPlanningProblem *, TransitionValidityCallbackType);            /* :145-150  -> PORRT_ERR_INVALID */
int set_cost_evaluator_callback(CPlanningProblem *, CostEvaluatorCallbackType);                      /* :152-157  -> PORRT_ERR_INVALID */
int set_observer_callback(CPlanningProblem *, ObserverCallbackType);                                 /* :160-165  -> PORRT_ERR_INVALID */
int set_start_belief_state(CPlanningProblem *, double *start_belief, size_t n_worlds, double **reachable_belief_states,
                           size_t n_reachable);                                                      /* :167-175; the reachable beliefs are computed by the library */
int set_goal_callback(CPlanningProblem *, GoalCallbackType);                                         /* :177-182  -> PORRT_ERR_INVALID */
int set_goal_example_callback(CPlanningProblem *, GoalExampleCallbackType);                          /* :184-189  -> PORRT_ERR_INVALID */
int set_search_parameters(CPlanningProblem *, size_t n_iterations_min, size_t n_iterations_max, double max_step,
                          double search_radius);                                                     /* :191-199 */
int set_refine_parameters(CPlanningProblem *, size_t refine_iterations);                             /* :201-206 */
int plan(CPlanningProblem *, double *start, size_t n);                                               /* :226-241 */
int get_planning_metrics(CPlanningProblem *, size_t *n_iterations, double *graph_growth_s, double *belief_space_expansion_s,
                         double *dynamic_programming_s, double *refinement_s, double *total_s);      /* :243-253 */
int get_paths_info(CPlanningProblem *, size_t *number_of_paths, size_t **path_lengths, double *expected_cost);   /* :255-262 */
int get_paths_variable(CPlanningProblem *, size_t path_id, size_t state_id, double **state, size_t *state_size);   /* :264-270 */

/* ---- the domain as data (no counterpart in pto_c.rs: there it is code behind the callbacks) */
/* occ: W*H gray raster (porrt_read_pgm), row-major from the top, covering [low, up) of the sampling bounds set above;
 * domain 0 = MapShelfDomain, 1 = Map (doors); zone_ids (W*H, 255 = none) and visibility as add_zones
 * (map_shelves_io.rs:105-117, map_io.rs:113-128); NULL zone_ids = no zones. */
int po_rrt_set_grid_domain(CPlanningProblem *, const uint8_t *occ, uint32_t W, uint32_t H, int domain, const uint8_t *zone_ids, double visibility);
/* SquareGoal::new(centers with world masks, max_dist) (common.rs:311-334) */
int po_rrt_set_square_goals(CPlanningProblem *, const double *centers /* G*2 */, const uint64_t *world_masks /* G */, uint32_t G, double l1_radius);
int po_rrt_set_seed(CPlanningProblem *, uint64_t continuous_seed, uint64_t discrete_seed);     /* reproducible runs (the reference: OS entropy) */
int po_rrt_set_device(CPlanningProblem *, int device, uint32_t batch_K);                      /* GPU and samples per grow step (default 0, 256) */
const char *po_rrt_last_error(const CPlanningProblem *);

#ifdef __cplusplus
}
#endif
#endif
