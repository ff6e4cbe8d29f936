#!/usr/bin/env python3
"""developer probe: the single query with pipeline = 1 / 2 / 3 (python tools/coop_probe.py [n_iter]): parity against pipeline 1 and timing"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import cases, po_rrt_amd
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 111500
case = cases.cfg2(n_iter)
ref = None
for pl in [int(a) for a in sys.argv[2:]] or (1, 4, 0):
    e = cases.configure(po_rrt_amd.Engine(0), case)
    e.set_option("pipeline", pl)
    ts = []
    try:
        for r in range(4):
            e.set_sampler((-1.0, -1.0), (1.0, 1.0), 5)
            t0 = time.perf_counter()
            cases.grow(e, case, K=1024)
            ts.append(time.perf_counter() - t0)
    except Exception as ex:
        print("pipeline", pl, "FAILED:", ex, flush=True)
        continue
    tr = e.tree() + (e.final_ids(),)
    same = ref is None or all(np.array_equal(a, b) for a, b in zip(tr, ref))
    if ref is None:
        ref = tr
    print("pipeline", pl, "ms", ["%.2f" % (1e3 * t) for t in ts], "nodes", e.num_nodes(), "same as pipeline 1:", same, "tie fallbacks", e.metrics()["n_tie_fallbacks"], flush=True)
