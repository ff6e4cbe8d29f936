#!/usr/bin/env python3
"""Throughput of porrt_grow_batch vs the number of concurrent queries per GPU (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases, po_rrt_amd
case = cases.cfg2(111500)
for Q in [int(x) for x in os.environ.get("QS", "1,2,4,8,16").split(",")]:
    engs = [cases.configure(po_rrt_amd.Engine(0), cases.cfg2(111500, seed=100 + q)) for q in range(Q)]
    starts = [case.start] * Q
    if os.environ.get("KDG"):
        engs[0].set_option("kd_group", int(os.environ["KDG"]))
    for w in range(2):
        po_rrt_amd.Engine.grow_batch(engs, starts, case.max_step, case.search_radius, case.n_iter_min, 1024)
    R = 5
    t0 = time.perf_counter()
    n = 0
    for r in range(R):
        po_rrt_amd.Engine.grow_batch(engs, starts, case.max_step, case.search_radius, case.n_iter_min, 1024)
        n += sum(e.num_nodes() - 1 for e in engs)
    dt = time.perf_counter() - t0
    print("Q=%2d  %.1f M nodes/s  (%.2f ms per batch, %.2f ms per query)" % (Q, n / dt / 1e6, dt / R * 1e3, dt / R / Q * 1e3), flush=True)
    del engs
