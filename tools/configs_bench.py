#!/usr/bin/env python3
"""Timings of BASELINE.json's other configs (parity-test cases, not bench lines): cfg1, cfg3, cfg4, door map, on the
GPU and with the C oracle on the host, for the table in BASELINE.md.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases
import po_rrt_amd
from oracle import orc

orc.build()
rows = [("cfg1 K=1", cases.cfg1(), 1), ("cfg1 K=64", cases.cfg1(), 64),
        ("cfg3 K=256 (n_iter_min 20000)", cases.cfg3(20000, 100000), 256),
        ("cfg4 K=256 (n_iter_min 20000)", cases.cfg4(20000, 100000), 256),
        ("door paper map K=256 (n_iter_min 20000)", cases.cfg_door(20000, 100000, paper=True), 256)]
for name, case, K in rows:
    e = cases.configure(po_rrt_amd.Engine(), case)
    cases.grow(e, case, K=K)                      # warm-up (graph instantiation, tables)
    e.set_sampler((-1.0, -1.0), (1.0, 1.0), case.seed)
    if case.mode == cases.PTO:
        e.set_discrete_seed(case.seed)
    t0 = time.perf_counter(); cases.grow(e, case, K=K); tg = time.perf_counter() - t0
    m = e.metrics()
    o = cases.configure(orc.Oracle(), case)
    t0 = time.perf_counter(); cases.grow(o, case, K=1, algo=orc.ALGO_SEQ); tc = time.perf_counter() - t0
    print("%-42s GPU: %6d nodes %7d iters %8.2f ms (device %7.2f ms) | CPU seq: %6d nodes %8.1f ms" %
          (name, e.num_nodes(), e.num_iterations(), tg * 1e3, m["device_s"] * 1e3, o.num_nodes(), tc * 1e3), flush=True)
