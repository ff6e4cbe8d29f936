"""Developer probe: per-step kernel times of one query and of a batch (PORRT_DEBUG_STEPS), python tools/step_probe.py [Q] [opt=val ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
os.environ["PORRT_DEBUG_STEPS"] = "1"
os.environ["PORRT_DEBUG"] = "1"
import cases, po_rrt_amd
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opts = [a.split("=") for a in sys.argv[2:] if not a.startswith("n_iter=")]
n_iter = [int(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("n_iter=")]
case = cases.cfg2(n_iter[0] if n_iter else 111500)
engs = [cases.configure(po_rrt_amd.Engine(0), case) for _ in range(Q)]
for e in engs:
    for k, v in opts:
        e.set_option(k, int(v))
def go():
    for j, e in enumerate(engs):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), j)
    if Q == 1:
        cases.grow(engs[0], case, K=1024)
    else:
        po_rrt_amd.Engine.grow_batch(engs, [case.start] * Q, case.max_step, case.search_radius, case.n_iter_min, 1024)
go()
for e in engs:
    e.set_option("profile", 1)
print("==== profiled run", file=sys.stderr)
go()
