"""Developer probe: a few single queries of the bench's size (python tools/single_probe.py [opt=val ...])."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases, po_rrt_amd
case = cases.cfg2(111500)
e = cases.configure(po_rrt_amd.Engine(0), case)
for a in sys.argv[1:]:
    e.set_option(a.split("=")[0], int(a.split("=")[1]))
for r in range(4):
    e.set_sampler((-1.0, -1.0), (1.0, 1.0), 700 + r)
    t0 = time.perf_counter()
    cases.grow(e, case, K=1024)
    print("single query %.3f ms, %d nodes" % (1e3 * (time.perf_counter() - t0), e.num_nodes()), flush=True)
