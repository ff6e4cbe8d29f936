import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/tools"]
import cases, po_rrt_amd
Q = 64
case = cases.cfg2(111500)
engs = [cases.configure(po_rrt_amd.Engine(0), case) for _ in range(Q)]
for rep in range(2):
    for j, e in enumerate(engs):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), j)
        e.set_option("batch_streams", 1)
        e.set_option("profile", rep)
    po_rrt_amd.Engine.grow_batch(engs, [case.start] * Q, case.max_step, case.search_radius, case.n_iter_min, 1024)
