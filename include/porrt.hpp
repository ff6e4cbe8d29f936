// porrt.hpp -- C++ host mirror of the reference's operator interface for the accelerated path, header-only,
// on top of the C ABI (porrt_hip.h).  Names, argument meaning and error behaviour follow cambyse/po-rrt:
//   ContinuousSampler / DiscreteSampler   src/sample_space.rs:6-60
//   SquareGoal / ObservationGoal          src/common.rs:304-350, src/rrt.rs:325-341
//   MapShelfDomain / Map                  src/map_shelves_io.rs:65-148, src/map_io.rs:67-161 (open, add_zones)
//   RRTNode / RRTTree / RRT::plan(_several)  src/rrt.rs:14-62, 84-100, 183-246
//   PTO::grow_graph + PTOGraph + Reachability  src/pto.rs:55-139, src/pto_graph.rs:171-207, src/pto_reachability.rs
// The reference is Rust; this image has no Rust toolchain, so the compiled host is C++ (INTEGRATION.md has the
// Rust binding).  Panics of the reference become std::runtime_error.
#pragma once
#include "porrt_hip.h"

#include <array>
#include <cmath>
#include <cctype>
#include <cstdint>
#include <fstream>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

namespace po_rrt {

using State = std::array<double, 2>;
using WorldMask = uint64_t;          // bit w <-> world w (BitVec in the reference)

struct ContinuousSampler {           // sample_space.rs:6-28 (the reference hard-codes seed 0)
    State low, up;
    uint64_t seed = 0;
    ContinuousSampler(State l, State u, uint64_t s = 0) : low(l), up(u), seed(s) {}
};
struct DiscreteSampler {             // sample_space.rs:39-55
    uint64_t seed = 0;
    explicit DiscreteSampler(uint64_t s = 0) : seed(s) {}
};

struct SquareGoal {                  // common.rs:304-333
    std::vector<std::pair<State, WorldMask>> goal_to_validity;
    double max_dist;
    SquareGoal(std::vector<std::pair<State, WorldMask>> g, double d) : goal_to_validity(std::move(g)), max_dist(d) {
        if (goal_to_validity.empty()) throw std::runtime_error("should have at least one element");
    }
};
struct ObservationGoal {             // rrt.rs:325-341
    uint32_t zone_id;
};

// 8-bit gray raster, row-major; P5 and P2 PGM (what the reference loads through the `image` crate)
struct Raster {
    uint32_t W = 0, H = 0;
    std::vector<uint8_t> px;
    static Raster open(const std::string &path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("Impossible to open image: " + path);
        std::string magic;
        f >> magic;
        auto next_int = [&]() {
            for (;;) {
                int c = f.peek();
                if (c == '#') { std::string line; std::getline(f, line); }
                else if (isspace(c)) f.get();
                else break;
            }
            int v; f >> v; return v;
        };
        Raster r;
        r.W = next_int(); r.H = next_int();
        int maxv = next_int();
        (void)maxv;
        r.px.resize((size_t)r.W * r.H);
        if (magic == "P5") { f.get(); f.read((char *)r.px.data(), (std::streamsize)r.px.size()); }
        else if (magic == "P2") { for (auto &p : r.px) p = (uint8_t)next_int(); }
        else throw std::runtime_error("Wrong image format!");
        return r;
    }
};

struct GridDomain {                  // common part of MapShelfDomain / Map
    Raster img, zones;
    State low, up;
    double visibility_distance = 0.0;
    int domain;
    bool has_zones = false;
    GridDomain(Raster r, State l, State u, int d) : img(std::move(r)), low(l), up(u), domain(d) {}
    void add_zones(const std::string &path, double visibility) { zones = Raster::open(path); visibility_distance = visibility; has_zones = true; }
};
struct MapShelfDomain : GridDomain { // map_shelves_io.rs:80-114
    MapShelfDomain(Raster r, State l, State u) : GridDomain(std::move(r), l, u, PORRT_DOMAIN_SHELF) {}
    static MapShelfDomain open(const std::string &path, State l, State u) { return MapShelfDomain(Raster::open(path), l, u); }
};
struct Map : GridDomain {            // map_io.rs:82-128
    Map(Raster r, State l, State u) : GridDomain(std::move(r), l, u, PORRT_DOMAIN_DOOR) {}
    static Map open(const std::string &path, State l, State u) { return Map(Raster::open(path), l, u); }
};

struct RRTNode {                     // rrt.rs:14-18
    State state;
    std::optional<size_t> parent_id;
    double dist_from_root;
};
struct RRTTree {                     // rrt.rs:20-62
    std::vector<RRTNode> nodes;
    std::vector<State> get_path_to(size_t id) const {
        std::vector<State> path;
        const RRTNode *n = &nodes[id];
        path.push_back(n->state);
        while (n->parent_id) { n = &nodes[*n->parent_id]; path.push_back(n->state); }
        return std::vector<State>(path.rbegin(), path.rend());
    }
};

class Context {
public:
    explicit Context(int device = 0) : c_(porrt_create(device)) {
        if (!c_) throw std::runtime_error("porrt_create failed: no usable HIP device (there is no CPU fallback)");
    }
    ~Context() { porrt_destroy(c_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    porrt_ctx *get() const { return c_; }
    int check(int rc) const { if (rc < 0) throw std::runtime_error(porrt_last_error(c_)); return rc; }
    void set_domain(const GridDomain &m) {
        check(porrt_set_grid(c_, m.img.px.data(), m.img.W, m.img.H, m.low.data(), m.up.data(), m.domain));
        if (m.has_zones) check(porrt_set_zones(c_, m.zones.px.data(), m.visibility_distance));
    }
    void set_goal(const SquareGoal &g) {
        std::vector<double> c;
        std::vector<uint64_t> m;
        for (auto &p : g.goal_to_validity) { c.push_back(p.first[0]); c.push_back(p.first[1]); m.push_back(p.second); }
        check(porrt_set_square_goal(c_, c.data(), m.data(), (uint32_t)m.size(), g.max_dist));
    }
    void set_goal(const ObservationGoal &g) { check(porrt_set_observation_goal(c_, g.zone_id)); }
private:
    porrt_ctx *c_;
};

// rrt.rs:78-246.  `fns` is the grid-backed validator (the Funcs adapter of map_shelves_tamp_rrt.rs:35-47);
// nullptr = the default RTTFuncs (everything valid, rrt.rs:64-76).
class RRT {
public:
    uint32_t batch_K = 1024;         // samples per GPU step; 1 = the reference's sequential loop
    RRT(const ContinuousSampler &s, const MapShelfDomain *fns, int device = 0) : ctx_(device) {
        if (fns) ctx_.set_domain(*fns);
        ctx_.check(porrt_set_sampler(ctx_.get(), s.low.data(), s.up.data(), s.seed));
    }
    using Solution = std::pair<std::vector<State>, double>;
    template <class Goal>
    std::pair<std::optional<Solution>, RRTTree> plan(State start, const Goal &goal, double max_step, double search_radius,
                                                      size_t n_iter_min, size_t n_iter_max) {                     // rrt.rs:88-93
        auto [tree, finals] = grow_tree(start, goal, max_step, search_radius, n_iter_min, n_iter_max);
        return {get_best_solution(tree, finals), std::move(tree)};
    }
    template <class Goal>
    std::pair<std::vector<Solution>, RRTTree> plan_several(State start, const Goal &goal, double max_step, double search_radius,
                                                            size_t n_iter_min, size_t n_iter_max) {               // rrt.rs:95-100
        auto [tree, finals] = grow_tree(start, goal, max_step, search_radius, n_iter_min, n_iter_max);
        std::vector<Solution> out;
        for (size_t id : get_firstly_final_node_ids(tree, finals)) { auto p = tree.get_path_to(id); double c = path_cost(p); out.push_back({std::move(p), c}); }
        return {std::move(out), std::move(tree)};
    }
    // Many independent queries at once (the TAMP layer's pattern, map_shelves_tamp_rrt.rs:163-291): every planner keeps
    // its own map, sampler state and result; all of them advance in one launch sequence (porrt_grow_batch), each running the
    // loop of rrt.rs:109 with the caller's n_iter_min / n_iter_max (the TAMP search passes 2500 / 10000, main.rs:532).  Same
    // results as calling plan() on each, several times the throughput.
    template <class Goal>
    static std::vector<std::pair<std::optional<Solution>, RRTTree>> plan_batch(const std::vector<RRT *> &planners, const std::vector<State> &starts,
                                                                                const Goal &goal, double max_step, double search_radius, size_t n_iter_min,
                                                                                size_t n_iter_max = 0) {
        if (n_iter_max < n_iter_min) n_iter_max = n_iter_min;
        std::vector<porrt_ctx *> cs;
        std::vector<double> st;
        for (size_t q = 0; q < planners.size(); ++q) {
            planners[q]->ctx_.set_goal(goal);
            cs.push_back(planners[q]->ctx_.get());
            st.push_back(starts[q][0]); st.push_back(starts[q][1]);
        }
        planners.at(0)->ctx_.check(porrt_grow_batch(cs.data(), (uint32_t)cs.size(), st.data(), max_step, search_radius, n_iter_min, n_iter_max, planners[0]->batch_K,
                                                    PORRT_MODE_RRT));
        std::vector<std::pair<std::optional<Solution>, RRTTree>> out;
        for (RRT *p : planners) {
            auto [tree, finals] = p->read_tree();
            out.push_back({get_best_solution(tree, finals), std::move(tree)});
        }
        return out;
    }
    // Cost of each planner's best path (get_best_solution, rrt.rs:183-193) evaluated on the device without fetching the
    // trees; +inf = "No solution found".  After plan_batch / a grow of all planners it is one kernel launch.
    static std::vector<double> best_costs(const std::vector<RRT *> &planners) {
        std::vector<porrt_ctx *> cs;
        for (RRT *p : planners) cs.push_back(p->ctx_.get());
        std::vector<double> costs(cs.size());
        planners.at(0)->ctx_.check(porrt_best_cost_batch(cs.data(), (uint32_t)cs.size(), costs.data()));
        return costs;
    }
private:
    Context ctx_;
    template <class Goal>
    std::pair<RRTTree, std::vector<size_t>> grow_tree(State start, const Goal &goal, double max_step, double search_radius,
                                                      size_t n_iter_min, size_t n_iter_max) {                     // rrt.rs:102-174
        ctx_.set_goal(goal);
        ctx_.check(porrt_grow(ctx_.get(), start.data(), max_step, search_radius, n_iter_min, n_iter_max, batch_K, PORRT_MODE_RRT));
        return read_tree();
    }
    std::pair<RRTTree, std::vector<size_t>> read_tree() {
        const size_t n = porrt_num_nodes(ctx_.get());
        std::vector<double> xy(2 * n), dist(n);
        std::vector<int64_t> parent(n);
        ctx_.check(porrt_get_tree(ctx_.get(), xy.data(), parent.data(), dist.data()));
        RRTTree t;
        t.nodes.resize(n);
        for (size_t j = 0; j < n; ++j) {
            t.nodes[j].state = {xy[2 * j], xy[2 * j + 1]};
            if (parent[j] >= 0) t.nodes[j].parent_id = (size_t)parent[j];
            t.nodes[j].dist_from_root = dist[j];
        }
        std::vector<uint64_t> f(porrt_num_final(ctx_.get()));
        if (!f.empty()) ctx_.check(porrt_get_final_ids(ctx_.get(), f.data()));
        return {std::move(t), std::vector<size_t>(f.begin(), f.end())};
    }
    static double norm2(const State &a, const State &b) { double dx = b[0] - a[0], dy = b[1] - a[1]; return std::sqrt(dx * dx + dy * dy); }
    static double path_cost(const std::vector<State> &p) { double s = 0; for (size_t i = 0; i + 1 < p.size(); ++i) s += norm2(p[i], p[i + 1]); return s; }
    static std::optional<Solution> get_best_solution(const RRTTree &t, const std::vector<size_t> &finals) {      // rrt.rs:183-193
        std::optional<Solution> best;
        for (size_t id : finals) { auto p = t.get_path_to(id); double c = path_cost(p); if (!best || c < best->second) best = Solution{std::move(p), c}; }
        return best;                   // nullopt <=> Err("No solution found")
    }
    static std::vector<size_t> get_firstly_final_node_ids(const RRTTree &t, const std::vector<size_t> &finals) { // rrt.rs:229-246
        std::unordered_set<size_t> fin(finals.begin(), finals.end()), first;
        for (size_t id : finals) {
            size_t cur = id;
            while (t.nodes[cur].parent_id && fin.count(*t.nodes[cur].parent_id)) cur = *t.nodes[cur].parent_id;
            first.insert(cur);
        }
        return std::vector<size_t>(first.begin(), first.end());
    }
};

// pto_graph.rs:171-207
struct PTOEdge { size_t id; size_t validity_id; };
struct PTONode { State state; size_t validity_id; std::vector<PTOEdge> parents, children; };
struct PTOGraph {
    std::vector<PTONode> nodes;
    std::vector<WorldMask> validities;
    void add_edge(size_t from, size_t to, size_t v) { nodes[from].children.push_back({to, v}); nodes[to].parents.push_back({from, v}); }
};

// pto.rs:15-149 (growth part) with the Reachability read-outs of pto_reachability.rs:54-90
using BeliefState = std::vector<double>;                       // common.rs: one probability per world

// common.rs:256-264
inline bool is_compatible(const BeliefState &belief_state, WorldMask validity) {
    for (size_t w = 0; w < belief_state.size(); ++w)
        if (belief_state[w] > 0.0 && !((validity >> w) & 1)) return false;
    return true;
}

enum class BeliefNodeType : uint8_t { Unknown = 0, Action = 1, Observation = 2 };     // belief_graph.rs:13-17

// BeliefGraph (belief_graph.rs:19-71) in CSR form: node i has node_type(i), children(i), parents(i); its graph node is
// i / n_beliefs, its belief reachable_belief_states[i % n_beliefs].
struct BeliefGraph {
    std::vector<BeliefState> reachable_belief_states;
    std::vector<uint8_t> node_types;
    std::vector<uint64_t> children_offsets, parents_offsets;
    std::vector<uint32_t> children_ids, parents_ids;
    size_t n_nodes() const { return node_types.size(); }
    size_t n_beliefs() const { return reachable_belief_states.size(); }
    BeliefNodeType node_type(size_t i) const { return (BeliefNodeType)node_types[i]; }
    size_t belief_id(size_t i) const { return i % n_beliefs(); }
    std::pair<const uint32_t *, const uint32_t *> children(size_t i) const { return {children_ids.data() + children_offsets[i], children_ids.data() + children_offsets[i + 1]}; }
    std::pair<const uint32_t *, const uint32_t *> parents(size_t i) const { return {parents_ids.data() + parents_offsets[i], parents_ids.data() + parents_offsets[i + 1]}; }
};

// common.rs:24-40
struct PolicyNode {
    State state;
    BeliefState belief_state;
    std::optional<size_t> parent;
    std::vector<size_t> children;
    size_t original_node_id = 0;
};
struct Policy {
    std::vector<PolicyNode> nodes;
    std::vector<size_t> leafs;
    double expected_costs = 0.0;
    const PolicyNode &leaf(size_t id) const { return nodes[leafs[id]]; }                             // common.rs:66-68
    std::vector<State> path_to_leaf(size_t id) const {                                               // common.rs:70-83
        std::vector<State> path;
        const PolicyNode *n = &leaf(id);
        path.push_back(n->state);
        while (n->parent) { n = &nodes[*n->parent]; path.push_back(n->state); }
        return std::vector<State>(path.rbegin(), path.rend());
    }
};

class PTO {
public:
    uint32_t batch_K = 256;
    PTOGraph graph;
    size_t n_it = 0;
    PTO(const ContinuousSampler &cs, const DiscreteSampler &ds, const GridDomain &fns, int device = 0) : ctx_(device) {
        ctx_.set_domain(fns);
        ctx_.check(porrt_set_sampler(ctx_.get(), cs.low.data(), cs.up.data(), cs.seed));
        ctx_.check(porrt_set_discrete_seed(ctx_.get(), ds.seed));
    }
    // Ok <=> returns true; false <=> Err("final nodes are not reached for each world") (pto.rs:134-138)
    bool grow_graph(State start, const SquareGoal &goal, double max_step, double search_radius, size_t n_iter_min, size_t n_iter_max) {
        ctx_.set_goal(goal);
        int rc = ctx_.check(porrt_grow(ctx_.get(), start.data(), max_step, search_radius, n_iter_min, n_iter_max, batch_K, PORRT_MODE_PTO));
        const size_t n = porrt_num_nodes(ctx_.get());
        n_it = porrt_num_iterations(ctx_.get());
        std::vector<double> xy(2 * n);
        std::vector<uint32_t> vid(n);
        ctx_.check(porrt_get_tree(ctx_.get(), xy.data(), nullptr, nullptr));
        ctx_.check(porrt_get_node_validity(ctx_.get(), vid.data()));
        reach_.assign(n, 0);
        ctx_.check(porrt_get_reach(ctx_.get(), reach_.data()));
        uint64_t val[65];
        int nv = porrt_get_validities(ctx_.get(), val);
        graph.validities.assign(val, val + nv);
        graph.nodes.assign(n, PTONode{});
        for (size_t j = 0; j < n; ++j) { graph.nodes[j].state = {xy[2 * j], xy[2 * j + 1]}; graph.nodes[j].validity_id = vid[j]; }
        const size_t E = porrt_num_edges(ctx_.get());
        std::vector<uint32_t> f(E), t(E), v(E);
        if (E) ctx_.check(porrt_get_edges(ctx_.get(), f.data(), t.data(), v.data()));
        for (size_t e0 = 0; e0 < E;) {             // per new node: neighbour->new edges, then new->neighbour (pto.rs:111-120)
            size_t e1 = e0;
            while (e1 < E && t[e1] == t[e0]) ++e1;
            for (size_t e = e0; e < e1; ++e) graph.add_edge(f[e], t[e], v[e]);
            for (size_t e = e0; e < e1; ++e) graph.add_edge(t[e], f[e], v[e]);
            e0 = e1;
        }
        final_ids_.assign(porrt_num_final(ctx_.get()), 0);
        final_masks_.assign(final_ids_.size(), 0);
        if (!final_ids_.empty()) { ctx_.check(porrt_get_final_ids(ctx_.get(), final_ids_.data())); ctx_.check(porrt_get_final_masks(ctx_.get(), final_masks_.data())); }
        return rc == PORRT_OK;
    }
    WorldMask reachability(size_t id) const { return reach_[id]; }                                   // pto_reachability.rs:54-56
    std::vector<size_t> get_final_nodes_for_world(size_t world) const {                              // pto_reachability.rs:58-63
        std::vector<size_t> out;
        for (size_t k = 0; k < final_ids_.size(); ++k)
            if (((reach_[final_ids_[k]] >> world) & 1) && ((final_masks_[k] >> world) & 1)) out.push_back((size_t)final_ids_[k]);
        return out;
    }
    int n_worlds() const { return porrt_n_worlds(ctx_.get()); }

    // PTO::build_belief_graph (pto.rs:185-259) on the graph of the last grow_graph.  belief_graph.nodes[i] is the belief
    // node of graph node i / n_beliefs and belief i % n_beliefs (the add_node order of pto.rs:198-201); children and
    // parents are views into the two id arrays, in the reference's Vec::push order.
    void build_belief_graph(const BeliefState &start_belief_state) {
        ctx_.check(porrt_build_belief_graph(ctx_.get(), start_belief_state.data(), (uint32_t)start_belief_state.size()));
        BeliefGraph &g = belief_graph;
        const size_t nb = porrt_bg_num_beliefs(ctx_.get()), nn = porrt_bg_num_nodes(ctx_.get()), ne = porrt_bg_num_edges(ctx_.get());
        const size_t nw = (size_t)n_worlds();
        std::vector<double> flat(nb * nw);
        ctx_.check(porrt_bg_get_beliefs(ctx_.get(), flat.data()));
        g.reachable_belief_states.assign(nb, BeliefState(nw));
        for (size_t b = 0; b < nb; ++b) g.reachable_belief_states[b].assign(flat.begin() + b * nw, flat.begin() + (b + 1) * nw);
        g.node_types.assign(nn, 0);
        g.children_offsets.assign(nn + 1, 0); g.parents_offsets.assign(nn + 1, 0);
        g.children_ids.assign(ne, 0); g.parents_ids.assign(ne, 0);
        ctx_.check(porrt_bg_get_node_types(ctx_.get(), g.node_types.data()));
        ctx_.check(porrt_bg_get_children(ctx_.get(), g.children_offsets.data(), g.children_ids.data()));
        ctx_.check(porrt_bg_get_parents(ctx_.get(), g.parents_offsets.data(), g.parents_ids.data()));
        // node_to_belief_nodes (pto.rs:196-208): Some(id) iff the belief is compatible with the node's validity
        node_to_belief_nodes.assign(graph.nodes.size(), std::vector<std::optional<size_t>>(nb));
        for (size_t n = 0; n < graph.nodes.size(); ++n)
            for (size_t b = 0; b < nb; ++b)
                if (is_compatible(g.reachable_belief_states[b], graph.validities[graph.nodes[n].validity_id])) node_to_belief_nodes[n][b] = n * nb + b;
    }
    BeliefGraph belief_graph;
    std::vector<std::vector<std::optional<size_t>>> node_to_belief_nodes;

    // pto.rs:261-275 (conditional_dijkstra, belief_graph.rs:89-175); the costs stay on the device, fetch = copy them out
    void compute_expected_costs_to_goals(bool fetch = true) {
        ctx_.check(porrt_bg_compute_expected_costs(ctx_.get()));
        if (fetch) {
            expected_costs_to_goals.assign(porrt_bg_num_nodes(ctx_.get()), 0.0);
            ctx_.check(porrt_bg_get_expected_costs(ctx_.get(), expected_costs_to_goals.data()));
        }
    }
    std::vector<double> expected_costs_to_goals;
    // pto.rs:277-283 (extract_policy, belief_graph.rs:177-263)
    Policy extract_policy() {
        Policy policy;
        const int64_t n = porrt_bg_extract_policy(ctx_.get(), nullptr, nullptr, nullptr, 0, &policy.expected_costs);
        if (n < 0) ctx_.check((int)n);
        std::vector<uint64_t> oid((size_t)n);
        std::vector<int64_t> par((size_t)n);
        std::vector<uint8_t> leaf((size_t)n);
        porrt_bg_extract_policy(ctx_.get(), oid.data(), par.data(), leaf.data(), (uint64_t)n, nullptr);
        const size_t nb = porrt_bg_num_beliefs(ctx_.get());
        std::vector<double> flat(nb * (size_t)n_worlds());
        ctx_.check(porrt_bg_get_beliefs(ctx_.get(), flat.data()));
        for (size_t k = 0; k < (size_t)n; ++k) {
            PolicyNode pn;
            pn.state = graph.nodes[oid[k] / nb].state;
            pn.belief_state.assign(flat.begin() + (oid[k] % nb) * n_worlds(), flat.begin() + (oid[k] % nb + 1) * n_worlds());
            pn.original_node_id = (size_t)oid[k];
            if (par[k] >= 0) { pn.parent = (size_t)par[k]; policy.nodes[(size_t)par[k]].children.push_back(k); }
            policy.nodes.push_back(std::move(pn));
            if (leaf[k]) policy.leafs.push_back(k);
        }
        return policy;
    }
    // pto.rs:151-183
    Policy plan_belief_space(const BeliefState &start_belief_state) {
        build_belief_graph_on_device(start_belief_state);
        compute_expected_costs_to_goals(false);
        return extract_policy();
    }
    // the expansion without copying the lists to the host (plan_belief_space only needs the policy)
    void build_belief_graph_on_device(const BeliefState &start_belief_state) {
        ctx_.check(porrt_build_belief_graph(ctx_.get(), start_belief_state.data(), (uint32_t)start_belief_state.size()));
    }
private:
    Context ctx_;
    std::vector<uint64_t> reach_, final_ids_, final_masks_;
};

// prm.rs:13-109.  init(start) + grow_graph(...) build one roadmap per call pair (porrt_grow_prm evaluates all samples at
// once; growing an existing roadmap further is a new call with the full iteration count on a sampler reset to its seed).
class PRM {
public:
    PTOGraph graph;
    size_t n_it = 0;
    PRM(const ContinuousSampler &cs, const GridDomain &fns, int device = 0) : ctx_(device) {
        ctx_.set_domain(fns);
        ctx_.check(porrt_set_sampler(ctx_.get(), cs.low.data(), cs.up.data(), cs.seed));
    }
    void init(State start) { start_ = start; }                                                        // prm.rs:33-36
    void grow_graph(double max_step, double search_radius, size_t n_iter) {                          // prm.rs:38-51
        ctx_.check(porrt_grow_prm(ctx_.get(), start_.data(), max_step, search_radius, n_iter));
        n_it += n_iter;
        const size_t n = porrt_num_nodes(ctx_.get());
        std::vector<double> xy(2 * n);
        ctx_.check(porrt_get_tree(ctx_.get(), xy.data(), nullptr, nullptr));
        uint64_t val[65];
        int nv = porrt_get_validities(ctx_.get(), val);
        graph.validities.assign(val, val + nv);
        graph.nodes.assign(n, PTONode{});
        for (size_t j = 0; j < n; ++j) { graph.nodes[j].state = {xy[2 * j], xy[2 * j + 1]}; graph.nodes[j].validity_id = 0; }
        const size_t E = porrt_num_edges(ctx_.get());
        std::vector<uint32_t> f(E), t(E), v(E);
        if (E) ctx_.check(porrt_get_edges(ctx_.get(), f.data(), t.data(), v.data()));
        for (size_t e0 = 0; e0 < E;) {             // prm.rs:96-103: neighbour -> new for all, then new -> neighbour; validity 0
            size_t e1 = e0;
            while (e1 < E && t[e1] == t[e0]) ++e1;
            for (size_t e = e0; e < e1; ++e) graph.add_edge(f[e], t[e], 0);
            for (size_t e = e0; e < e1; ++e) graph.add_edge(t[e], f[e], 0);
            e0 = e1;
        }
    }
    // prm.rs:111-123: empty when start and goal are not connected
    std::vector<State> plan_path(State start, State goal) {
        const int64_t n = porrt_prm_plan_path(ctx_.get(), start.data(), goal.data(), nullptr, 0);
        if (n < 0) ctx_.check((int)n);
        std::vector<double> xy(2 * (size_t)n);
        if (n) porrt_prm_plan_path(ctx_.get(), start.data(), goal.data(), xy.data(), (uint64_t)n);
        std::vector<State> path((size_t)n);
        for (size_t k = 0; k < (size_t)n; ++k) path[k] = {xy[2 * k], xy[2 * k + 1]};
        return path;
    }
private:
    Context ctx_;
    State start_{0.0, 0.0};
};

} // namespace po_rrt
