"""Times porrt_grow_prm (PRM::grow_graph, prm.rs:38-109) on the benchmark map.  usage: python tools/prm_probe.py [n ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases  # noqa: E402
import po_rrt_amd  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [200000, 1000000]:
    e = po_rrt_amd.Engine()
    e.set_grid(cases.load_map("map_benchmark_like"), (-1.0, -1.0), (1.0, 1.0), cases.SHELF)
    for rep in range(3):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), rep)
        t = time.perf_counter()
        e.grow_prm((0.0, -0.8), 0.1, 2.0, n)
        wall = time.perf_counter() - t
    m = e.metrics()
    print("samples %8d  wall %.3f ms  device %.3f ms  %.1f M nodes/s" % (n, 1e3 * wall, 1e3 * m["device_s"], n / wall / 1e6), flush=True)
