"""Times porrt_build_belief_graph on the 12-shelf problem (main.rs:386-408) at growing graph sizes; prints edges,
seconds (total / device / host tables) and the CSR write rate.  usage: python tools/belief_probe.py [n_iter ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases  # noqa: E402
import po_rrt_amd  # noqa: E402

iters = [int(a) for a in sys.argv[1:]] or [2000, 5000]
for n_worlds_possible in [int(w) for w in os.environ.get("WORLDS", "8,12").split(",")]:
    for n in iters:
        case = cases.cfg4(n, n)
        case.update(start=(0.0, -0.3))
        e = cases.configure(po_rrt_amd.Engine(), case)
        cases.grow(e, case, K=256)
        prior = [1.0 / n_worlds_possible] * n_worlds_possible + [0.0] * (12 - n_worlds_possible)
        for rep in range(2):
            t = time.perf_counter()
            e.build_belief_graph(prior)
            wall = time.perf_counter() - t
        s = e.bg_seconds()
        E = e.bg_num_edges()
        nb = e.num_nodes() * (2 ** n_worlds_possible - 1)
        bytes_out = 8.0 * E + 2 * 8.0 * nb + nb
        print("worlds %2d  graph nodes %6d  belief nodes %9d  edges %11d  wall %.4f s  total %.4f  device %.4f  tables %.4f  "
              "lists %.1f GB/s (device time)" % (n_worlds_possible, e.num_nodes(), nb, E, wall, s["total_s"], s["device_s"],
                                                 s["host_tables_s"], bytes_out / max(s["device_s"], 1e-9) / 1e9), flush=True)
        print("     " + "  ".join("%s %.2f ms" % (k[:-2], 1e3 * v) for k, v in s.items()), flush=True)
        for rep in range(2):
            t = time.perf_counter()
            e.compute_expected_costs()
            wall = time.perf_counter() - t
        info = e.dp_info()
        print("     expected costs: wall %.2f ms  device %.2f ms  %d sweeps  root cost %r" % (1e3 * wall, 1e3 * info["device_s"], info["sweeps"],
                                                                                          e.expected_cost_of(0)), flush=True)
        del e
