import sys, time
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/tools"]
import cases, po_rrt_amd
e = po_rrt_amd.Engine(0)
for K in (1, 256):
    for seed in range(6):
        case = cases.cfg_map4(5000, seed)
        cases.configure(e, case)
        cases.grow(e, case, K=K)
        e.build_belief_graph([1.0 / 16] * 16)
        e.compute_expected_costs()
        t = time.perf_counter()
        try:
            (oid, par, leaf), c = e.extract_policy()
            print(K, seed, "policy nodes", len(oid), "leafs", int(leaf.sum()), "cost", 7.65 * c, "%.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
        except Exception as ex:
            print(K, seed, "ERR", str(ex)[:120], "%.1f ms" % (1e3 * (time.perf_counter() - t)), flush=True)
