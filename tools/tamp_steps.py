"""Developer probe: per-step kernel times of the TAMP-shaped batch (python tools/tamp_steps.py [queries] [K]); PORRT_DEBUG_STEPS=1"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
os.environ["PORRT_DEBUG_STEPS"] = "1"
import cases, po_rrt_amd
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cs = cases.tamp_queries(nq)
engs = [cases.configure(po_rrt_amd.Engine(0), c) for c in cs]
for rep in range(2):
    for j, e in enumerate(engs):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), j)
        e.set_option("batch_streams", 1)
        e.set_option("profile", rep)
    po_rrt_amd.Engine.grow_batch(engs, [c.start for c in cs], 0.1, 2.0, 2500, K, n_iter_max=10000)
