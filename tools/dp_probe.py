"""Developer probe: expected costs on the bench's 4095-belief graph alone (python tools/dp_probe.py [runs]); prints ms per
computation, the sweep count and a digest of the costs (to compare builds)."""
import sys, os, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import cases, po_rrt_amd
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_iter = 20000
case = cases.cfg4(n_iter, n_iter)
case.update(start=(0.0, -0.3))
e = cases.configure(po_rrt_amd.Engine(0), case)
for o in sys.argv[2:]:
    k, v = o.split("=")
    e.set_option(k, int(v))
cases.grow(e, case, K=256)
e.build_belief_graph([1.0 / 12] * 12)
for _ in range(runs):
    t0 = time.perf_counter()
    e.compute_expected_costs()
    dt = time.perf_counter() - t0
    print("expected costs: %.2f ms wall, %s" % (1e3 * dt, e.dp_info()), flush=True)
d = e.expected_costs()
print("edges %d, nodes %d, digest %s, root %r" % (e.bg_num_edges(), e.num_nodes(), hashlib.sha256(np.ascontiguousarray(d).tobytes()).hexdigest()[:16], float(d[0])), flush=True)
