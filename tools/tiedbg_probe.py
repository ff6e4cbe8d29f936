import sys
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/tools"]
import cases, po_rrt_amd
c2 = cases.cfg2(111500)
es = [cases.configure(po_rrt_amd.Engine(0), cases.Case(c2, seed=j)) for j in range(8)]
po_rrt_amd.Engine.grow_batch(es, [c2.start] * 8, c2.max_step, c2.search_radius, c2.n_iter_min, 1024)
