// Plans one RRT* query with the C++ mirror of the reference interface (cf. src/rrt.rs:269-303,
// test_plan_on_map7_prefefined_goal) and prints a digest the tests compare with the CPU oracle.
// usage: plan_rrt <map.pgm> <n_iter_min> <n_iter_max> <batch_K> <seed> [n_queries]
// With n_queries > 1 the query is planned together with n_queries - 1 others (seeds seed+1, ...) by RRT::plan_batch
// (fixed budget n_iter_max); the digest printed is still the one of `seed`.
#include "../include/porrt.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>

int main(int argc, char **argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s map.pgm n_iter_min n_iter_max batch_K seed\n", argv[0]); return 2; }
    using namespace po_rrt;
    try {
        auto m = MapShelfDomain::open(argv[1], {-1.0, -1.0}, {1.0, 1.0});
        SquareGoal goal({{{0.9, 0.0}, 1}}, 0.05);
        RRT rrt(ContinuousSampler({-1.0, -1.0}, {1.0, 1.0}, std::strtoull(argv[5], nullptr, 10)), &m);
        rrt.batch_K = (uint32_t)std::atoi(argv[4]);
        const int n_queries = argc > 6 ? std::atoi(argv[6]) : 1;
        std::optional<RRT::Solution> result;
        RRTTree tree;
        if (n_queries > 1) {
            std::vector<std::unique_ptr<RRT>> others;
            std::vector<RRT *> all{&rrt};
            for (int q = 1; q < n_queries; ++q) {
                others.emplace_back(new RRT(ContinuousSampler({-1.0, -1.0}, {1.0, 1.0}, std::strtoull(argv[5], nullptr, 10) + q), &m));
                others.back()->batch_K = rrt.batch_K;
                all.push_back(others.back().get());
            }
            auto res = RRT::plan_batch(all, std::vector<State>(all.size(), State{0.0, -1.0}), goal, 0.1, 2.0, std::strtoull(argv[3], nullptr, 10));
            const std::vector<double> costs = RRT::best_costs(all);        // device-side costs == the host walk's
            for (size_t q = 0; q < all.size(); ++q) {
                const bool found = res[q].first.has_value();
                if (found ? costs[q] != res[q].first->second : costs[q] != std::numeric_limits<double>::infinity()) {
                    std::fprintf(stderr, "best_costs[%zu] disagrees with get_best_solution\n", q);
                    return 3;
                }
            }
            result = std::move(res[0].first);
            tree = std::move(res[0].second);
        } else {
            auto r1 = rrt.plan({0.0, -1.0}, goal, 0.1, 2.0, std::strtoull(argv[2], nullptr, 10), std::strtoull(argv[3], nullptr, 10));
            result = std::move(r1.first);
            tree = std::move(r1.second);
        }
        uint64_t h = 1469598103934665603ull;                 // FNV-1a over parents and coordinate bits
        for (auto &n : tree.nodes) {
            uint64_t v[3];
            v[0] = n.parent_id ? *n.parent_id : ~0ull;
            std::memcpy(&v[1], &n.state[0], 8);
            std::memcpy(&v[2], &n.state[1], 8);
            for (uint64_t x : v) for (int b = 0; b < 8; ++b) { h ^= (x >> (8 * b)) & 0xff; h *= 1099511628211ull; }
        }
        std::printf("nodes %zu digest %016llx ", tree.nodes.size(), (unsigned long long)h);
        if (result) std::printf("path %zu cost %.17g\n", result->first.size(), result->second);
        else std::printf("No solution found\n");
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
