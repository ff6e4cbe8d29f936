"""Developer probe: one grow_mm_prm of the bench's size (python tools/mm_probe.py [reps])."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases, po_rrt_amd
case = cases.cfg4(1000, 1000)
e = cases.configure(po_rrt_amd.Engine(0), case)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    e.set_discrete_seed(rep)
    t0 = time.perf_counter()
    g = e.grow_mm_prm(case.start, [1.0 / 12] * 12, 0.1, 2.0, 100)
    dt = time.perf_counter() - t0
    print("modes %d nodes %d wall %.3f s" % (len(g["modes"]), sum(len(m["xy"]) for m in g["modes"]), dt), e.mm_seconds(), flush=True)
