// pto_c_shim.cpp -- libpo_rrt.so: the symbols of the reference's C boundary (cambyse/po-rrt src/pto_c.rs:63-270) on top of
// the engine's public C ABI (include/porrt_hip.h; nothing else of the engine is used).  See include/po_rrt_c.h for the
// contract and the deliberate differences (no callbacks, no ownership of caller memory, error codes instead of panics).
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../include/po_rrt_c.h"
#include "../../include/porrt_hip.h"

struct CPlanningProblem {
    // input (pto_c.rs:30-49)
    size_t state_dim = 0, n_worlds = 0;
    std::vector<double> low, up, start_belief;
    std::vector<uint8_t> occ, zones;
    uint32_t W = 0, H = 0;
    int domain = 0;
    bool has_grid = false, has_zones = false, has_goal = false, callbacks_given = false;
    double visibility = 0.0;
    std::vector<double> goal_centers;
    std::vector<uint64_t> goal_masks;
    double goal_l1 = 0.0;
    size_t n_iterations_min = 0, n_iterations_max = 0, refine_iterations = 0;
    double max_step = 0.0, search_radius = 0.0;
    bool seeded = false;
    uint64_t cseed = 0, dseed = 0;
    int device = 0;
    uint32_t batch_K = 256;
    // output (pto_c.rs:51-60)
    std::vector<std::vector<std::array<double, 2>>> paths;
    std::vector<size_t> paths_lengths;
    double expected_costs = 0.0;
    size_t n_iterations = 0;
    double graph_growth_s = 0, belief_space_expansion_s = 0, dynamic_programming_s = 0, refinement_s = 0, total_s = 0;
    std::string err;
    int fail(int code, const std::string &m) { err = m; return code; }
};

static const char *kNoCallbacks =
    "callbacks cannot run on the GPU: declare the domain with po_rrt_set_grid_domain / po_rrt_set_square_goals (include/po_rrt_c.h)";

extern "C" {

CPlanningProblem *new_planning_problem(void) { return new CPlanningProblem(); }
void delete_planning_problem(CPlanningProblem *p) { delete p; }
const char *po_rrt_last_error(const CPlanningProblem *p) { return p ? p->err.c_str() : "null planning problem"; }

int set_problem_dimensions(CPlanningProblem *p, size_t state_dim, size_t n_worlds) {
    if (!p) return PORRT_ERR_INVALID;
    if (state_dim != 2) return p->fail(PORRT_ERR_INVALID, "state_dim must be 2 (the grid-backed domains are 2-D)");
    if (n_worlds == 0 || n_worlds > 64) return p->fail(PORRT_ERR_INVALID, "n_worlds must be in 1..64");
    p->state_dim = state_dim; p->n_worlds = n_worlds;
    return PORRT_OK;
}
int set_lower_sampling_bound(CPlanningProblem *p, double *low, size_t n) {
    if (!p || !low) return PORRT_ERR_INVALID;
    if (n != p->state_dim) return p->fail(PORRT_ERR_INVALID, "lower bound: size differs from state_dim (the reference asserts)");
    p->low.assign(low, low + n);
    return PORRT_OK;
}
int set_upper_sampling_bound(CPlanningProblem *p, double *up, size_t n) {
    if (!p || !up) return PORRT_ERR_INVALID;
    if (n != p->state_dim) return p->fail(PORRT_ERR_INVALID, "upper bound: size differs from state_dim (the reference asserts)");
    p->up.assign(up, up + n);
    return PORRT_OK;
}
int set_world_validities(CPlanningProblem *p, size_t **, size_t) { return p ? PORRT_OK : PORRT_ERR_INVALID; }     // derived from the zone raster
int set_state_validity_callback(CPlanningProblem *p, StateValidityCallbackType) { if (!p) return PORRT_ERR_INVALID; p->callbacks_given = true; return p->fail(PORRT_ERR_INVALID, kNoCallbacks); }
int set_transition_validity_callback(CPlanningProblem *p, TransitionValidityCallbackType) { if (!p) return PORRT_ERR_INVALID; p->callbacks_given = true; return p->fail(PORRT_ERR_INVALID, kNoCallbacks); }
int set_cost_evaluator_callback(CPlanningProblem *p, CostEvaluatorCallbackType) { if (!p) return PORRT_ERR_INVALID; p->callbacks_given = true; return p->fail(PORRT_ERR_INVALID, kNoCallbacks); }
int set_observer_callback(CPlanningProblem *p, ObserverCallbackType) { if (!p) return PORRT_ERR_INVALID; p->callbacks_given = true; return p->fail(PORRT_ERR_INVALID, kNoCallbacks); }
int set_goal_callback(CPlanningProblem *p, GoalCallbackType) { if (!p) return PORRT_ERR_INVALID; p->callbacks_given = true; return p->fail(PORRT_ERR_INVALID, kNoCallbacks); }
int set_goal_example_callback(CPlanningProblem *p, GoalExampleCallbackType) { if (!p) return PORRT_ERR_INVALID; p->callbacks_given = true; return p->fail(PORRT_ERR_INVALID, kNoCallbacks); }
int set_start_belief_state(CPlanningProblem *p, double *b, size_t n, double **, size_t) {
    if (!p || !b) return PORRT_ERR_INVALID;
    if (n != p->n_worlds) return p->fail(PORRT_ERR_INVALID, "start belief: size differs from n_worlds (the reference asserts)");
    p->start_belief.assign(b, b + n);
    return PORRT_OK;
}
int set_search_parameters(CPlanningProblem *p, size_t nmin, size_t nmax, double max_step, double search_radius) {
    if (!p) return PORRT_ERR_INVALID;
    p->n_iterations_min = nmin; p->n_iterations_max = nmax; p->max_step = max_step; p->search_radius = search_radius;
    return PORRT_OK;
}
int set_refine_parameters(CPlanningProblem *p, size_t n) { if (!p) return PORRT_ERR_INVALID; p->refine_iterations = n; return PORRT_OK; }

int po_rrt_set_grid_domain(CPlanningProblem *p, const uint8_t *occ, uint32_t W, uint32_t H, int domain, const uint8_t *zone_ids, double visibility) {
    if (!p || !occ || !W || !H) return PORRT_ERR_INVALID;
    if (domain != PORRT_DOMAIN_SHELF && domain != PORRT_DOMAIN_DOOR) return p->fail(PORRT_ERR_INVALID, "domain: 0 = MapShelfDomain, 1 = Map");
    p->occ.assign(occ, occ + (size_t)W * H);
    p->has_zones = zone_ids != nullptr;
    if (zone_ids) p->zones.assign(zone_ids, zone_ids + (size_t)W * H);
    p->W = W; p->H = H; p->domain = domain; p->visibility = visibility; p->has_grid = true;
    return PORRT_OK;
}
int po_rrt_set_square_goals(CPlanningProblem *p, const double *centers, const uint64_t *masks, uint32_t G, double l1) {
    if (!p || !centers || !masks || !G || G > 64) return PORRT_ERR_INVALID;
    p->goal_centers.assign(centers, centers + 2 * (size_t)G);
    p->goal_masks.assign(masks, masks + G);
    p->goal_l1 = l1; p->has_goal = true;
    return PORRT_OK;
}
int po_rrt_set_seed(CPlanningProblem *p, uint64_t c, uint64_t d) { if (!p) return PORRT_ERR_INVALID; p->seeded = true; p->cseed = c; p->dseed = d; return PORRT_OK; }
int po_rrt_set_device(CPlanningProblem *p, int device, uint32_t K) { if (!p || !K) return PORRT_ERR_INVALID; p->device = device; p->batch_K = K; return PORRT_OK; }

// plan_inner! (pto_c.rs:208-224): PTO::new, grow_graph, plan_belief_space (build_belief_graph, expected costs, policy),
// save_planning_metrics, save_paths -- without the refiner.
int plan(CPlanningProblem *p, double *start, size_t n) {
    if (!p || !start) return PORRT_ERR_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    p->paths.clear(); p->paths_lengths.clear();
    if (p->callbacks_given) return p->fail(PORRT_ERR_INVALID, kNoCallbacks);
    if (n != p->state_dim || n != 2) return p->fail(PORRT_ERR_INVALID, "start: size differs from state_dim (the reference asserts)");
    if (!p->has_grid || !p->has_goal) return p->fail(PORRT_ERR_INVALID, "no domain: call po_rrt_set_grid_domain and po_rrt_set_square_goals");
    if (p->low.size() != 2 || p->up.size() != 2 || p->start_belief.size() != p->n_worlds) return p->fail(PORRT_ERR_INVALID, "sampling bounds / start belief not set");
    porrt_ctx *c = porrt_create(p->device);
    if (!c) return p->fail(PORRT_ERR_NO_DEVICE, "no usable HIP device");
    auto leave = [&](int code, const char *what) {
        p->err = std::string(what) + ": " + porrt_last_error(c);
        porrt_destroy(c);
        return code;
    };
    int r = porrt_set_grid(c, p->occ.data(), p->W, p->H, p->low.data(), p->up.data(), p->domain);
    if (r) return leave(r, "set_grid");
    if (p->has_zones && (r = porrt_set_zones(c, p->zones.data(), p->visibility))) return leave(r, "set_zones");
    if ((size_t)porrt_n_worlds(c) != p->n_worlds) { porrt_destroy(c); return p->fail(PORRT_ERR_INVALID, "n_worlds differs from what the zone raster defines"); }
    uint64_t cs = p->cseed, ds = p->dseed;
    if (!p->seeded) { std::random_device rd; cs = ((uint64_t)rd() << 32) | rd(); ds = ((uint64_t)rd() << 32) | rd(); }     // new_true_random (pto_c.rs:213)
    if ((r = porrt_set_sampler(c, p->low.data(), p->up.data(), cs)) || (r = porrt_set_discrete_seed(c, ds))) return leave(r, "set_sampler");
    if ((r = porrt_set_square_goal(c, p->goal_centers.data(), p->goal_masks.data(), (uint32_t)p->goal_masks.size(), p->goal_l1))) return leave(r, "set_square_goal");
    r = porrt_grow(c, start, p->max_step, p->search_radius, p->n_iterations_min, p->n_iterations_max, p->batch_K, PORRT_MODE_PTO);
    if (r == PORRT_INCOMPLETE) return leave(PORRT_INCOMPLETE, "graph not grown up to solution");          // the reference: expect(..) panics
    if (r) return leave(r, "grow_graph");
    porrt_metrics m;
    porrt_get_metrics(c, &m);
    if ((r = porrt_build_belief_graph(c, p->start_belief.data(), (uint32_t)p->n_worlds))) return leave(r, "build_belief_graph");
    double bsec[8] = {0};
    porrt_bg_get_seconds(c, bsec, 8);
    if ((r = porrt_bg_compute_expected_costs(c))) return leave(r, "compute_expected_costs");
    double dp_total = 0, dp_dev = 0;
    uint32_t sweeps = 0;
    porrt_bg_get_dp_info(c, &dp_total, &dp_dev, &sweeps);
    double cost = 0;
    const int64_t np = porrt_bg_extract_policy(c, nullptr, nullptr, nullptr, 0, &cost);
    if (np < 0) return leave((int)np, "extract_policy");
    std::vector<uint64_t> oid((size_t)np);
    std::vector<int64_t> par((size_t)np);
    std::vector<uint8_t> leaf((size_t)np);
    porrt_bg_extract_policy(c, oid.data(), par.data(), leaf.data(), (uint64_t)np, &cost);
    const uint64_t N = porrt_num_nodes(c), B = porrt_bg_num_beliefs(c);
    std::vector<double> xy(2 * N);
    if ((r = porrt_get_tree(c, xy.data(), nullptr, nullptr))) return leave(r, "get_tree");
    // save_paths (pto_c.rs:274-304): leaf -> root, reversed
    for (int64_t k = 0; k < np; ++k) {
        if (!leaf[(size_t)k]) continue;
        std::vector<std::array<double, 2>> path;
        for (int64_t cur = k; cur >= 0; cur = par[(size_t)cur]) {
            const uint64_t node = oid[(size_t)cur] / B;
            path.push_back({xy[2 * node], xy[2 * node + 1]});
        }
        std::vector<std::array<double, 2>> rev(path.rbegin(), path.rend());
        p->paths_lengths.push_back(rev.size());
        p->paths.push_back(std::move(rev));
    }
    p->expected_costs = cost;
    p->n_iterations = (size_t)m.n_iter;
    p->graph_growth_s = m.total_s;
    p->belief_space_expansion_s = bsec[0];
    p->dynamic_programming_s = dp_total;
    p->refinement_s = 0.0;
    p->total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    p->err.clear();
    porrt_destroy(c);
    return PORRT_OK;
}

int get_planning_metrics(CPlanningProblem *p, size_t *n_it, double *g, double *b, double *d, double *rf, double *t) {
    if (!p) return PORRT_ERR_INVALID;
    if (n_it) *n_it = p->n_iterations;
    if (g) *g = p->graph_growth_s;
    if (b) *b = p->belief_space_expansion_s;
    if (d) *d = p->dynamic_programming_s;
    if (rf) *rf = p->refinement_s;
    if (t) *t = p->total_s;
    return PORRT_OK;
}
int get_paths_info(CPlanningProblem *p, size_t *n_paths, size_t **lengths, double *expected_cost) {
    if (!p) return PORRT_ERR_INVALID;
    if (n_paths) *n_paths = p->paths.size();
    if (lengths) *lengths = p->paths_lengths.data();
    if (expected_cost) *expected_cost = p->expected_costs;
    return PORRT_OK;
}
int get_paths_variable(CPlanningProblem *p, size_t path_id, size_t state_id, double **state, size_t *state_size) {
    if (!p || !state || !state_size) return PORRT_ERR_INVALID;
    if (path_id >= p->paths.size() || state_id >= p->paths[path_id].size()) return p->fail(PORRT_ERR_INVALID, "path / state index out of range (the reference panics)");
    *state = p->paths[path_id][state_id].data();
    *state_size = 2;
    return PORRT_OK;
}

} // extern "C"
