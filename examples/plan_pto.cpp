// Grows a belief-space graph and expands it over the reachable beliefs with the C++ mirror of the reference interface
// (cf. src/pto.rs:466-490, test_plan_on_map1_2_goals: grow_graph, then build_belief_graph) and prints digests the tests
// compare with the CPU oracle.
// usage: plan_pto <map.pgm> <zone_ids.pgm> <n_iter> <batch_K> <seed>
#include "../include/porrt.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>

static uint64_t fnv(uint64_t h, uint64_t x) {
    for (int b = 0; b < 8; ++b) { h ^= (x >> (8 * b)) & 0xff; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s map.pgm zone_ids.pgm n_iter batch_K seed\n", argv[0]); return 2; }
    using namespace po_rrt;
    try {
        auto m = MapShelfDomain::open(argv[1], {-1.0, -1.0}, {1.0, 1.0});
        m.add_zones(argv[2], 0.5);
        SquareGoal goal({{{0.68, -0.45}, 0b01}, {{0.68, 0.38}, 0b10}}, 0.05);
        const uint64_t seed = std::strtoull(argv[5], nullptr, 10);
        PTO pto(ContinuousSampler({-1.0, -1.0}, {1.0, 1.0}, seed), DiscreteSampler(0), m);
        pto.batch_K = (uint32_t)std::atoi(argv[4]);
        const size_t n_iter = std::strtoull(argv[3], nullptr, 10);
        const bool ok = pto.grow_graph({-0.8, -0.8}, goal, 0.05, 5.0, n_iter, n_iter);
        pto.build_belief_graph({0.5, 0.5});
        const BeliefGraph &g = pto.belief_graph;
        uint64_t hc = 1469598103934665603ull, hp = hc, ht = hc;
        size_t n_some = 0;
        for (size_t i = 0; i < g.n_nodes(); ++i) {
            ht = fnv(ht, (uint64_t)g.node_type(i));
            auto c = g.children(i);
            hc = fnv(hc, (uint64_t)(c.second - c.first));
            for (const uint32_t *p = c.first; p != c.second; ++p) hc = fnv(hc, *p);
            auto q = g.parents(i);
            hp = fnv(hp, (uint64_t)(q.second - q.first));
            for (const uint32_t *p = q.first; p != q.second; ++p) hp = fnv(hp, *p);
        }
        for (auto &row : pto.node_to_belief_nodes) for (auto &x : row) n_some += x.has_value();
        // pto.rs:480-490: compute_expected_costs_to_goals + extract_policy (only when a policy exists)
        pto.compute_expected_costs_to_goals();
        uint64_t hd = 1469598103934665603ull;
        for (double d : pto.expected_costs_to_goals) { uint64_t u; std::memcpy(&u, &d, 8); hd = fnv(hd, u); }
        size_t n_policy = 0, n_leafs = 0;
        uint64_t hpol = 1469598103934665603ull;
        if (pto.expected_costs_to_goals[0] < 1e300) {
            Policy policy = pto.extract_policy();
            n_policy = policy.nodes.size(); n_leafs = policy.leafs.size();
            for (auto &pn : policy.nodes) { hpol = fnv(hpol, pn.original_node_id); hpol = fnv(hpol, pn.parent ? *pn.parent : ~0ull); }
        }
        std::printf("costs %016llx policy_nodes %zu leafs %zu policy %016llx ", (unsigned long long)hd, n_policy, n_leafs, (unsigned long long)hpol);
        std::printf("complete %d nodes %zu beliefs %zu belief_nodes %zu edges %zu types %016llx children %016llx parents %016llx compatible %zu\n",
                    ok ? 1 : 0, pto.graph.nodes.size(), g.n_beliefs(), g.n_nodes(), g.children_ids.size(), (unsigned long long)ht,
                    (unsigned long long)hc, (unsigned long long)hp, n_some);
        // prm.rs:146-170: a PRM* roadmap on the same domain
        PRM prm(ContinuousSampler({-1.0, -1.0}, {1.0, 1.0}, seed + 100), m);
        prm.init({-0.8, -0.8});
        prm.grow_graph(0.1, 5.0, 2000);
        uint64_t hr = 1469598103934665603ull;
        size_t n_arcs = 0;
        for (auto &nd : prm.graph.nodes) {
            hr = fnv(hr, nd.children.size());
            for (auto &c : nd.children) hr = fnv(hr, c.id);
            n_arcs += nd.children.size();
        }
        auto path = prm.plan_path({-0.8, -0.8}, {-0.5, 0.5});
        uint64_t hpath = 1469598103934665603ull;
        for (auto &st : path) for (double v : st) { uint64_t u; std::memcpy(&u, &v, 8); hpath = fnv(hpath, u); }
        std::printf("prm_nodes %zu prm_arcs %zu prm %016llx path_len %zu path %016llx\n", prm.graph.nodes.size(), n_arcs, (unsigned long long)hr,
                    path.size(), (unsigned long long)hpath);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
