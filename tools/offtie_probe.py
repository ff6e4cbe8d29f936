import sys
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/tools"]
import cases, po_rrt_amd
c2 = cases.cfg2(111500)
es = [cases.configure(po_rrt_amd.Engine(0), cases.Case(c2, seed=j)) for j in range(32)]
po_rrt_amd.Engine.grow_batch(es, [c2.start] * 32, c2.max_step, c2.search_radius, c2.n_iter_min, 1024)
print("bench workload: steps the kd structure was built for (leader's counter):", sorted(e.get_option("kd_lca_steps") for e in es), "built after:", es[0].get_option("kd_built_after"))
cs = cases.tamp_queries(64)
es = [cases.configure(po_rrt_amd.Engine(0), c) for c in cs]
po_rrt_amd.Engine.grow_batch(es, [c.start for c in cs], 0.1, 2.0, 2500, 128, n_iter_max=10000)
print("tamp: ", sorted(e.get_option("kd_lca_steps") for e in es), "steps", sorted(e.metrics()["n_steps"] for e in es)[-1], "built after:", es[0].get_option("kd_built_after"))
