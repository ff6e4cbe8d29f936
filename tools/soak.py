"""Developer probe: many batches on the same contexts -- does the time per batch, the host memory or the device memory drift?
(python tools/soak.py [batches] [Q])"""
import sys, os, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np, torch
import cases, po_rrt_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 150
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 128
case = cases.cfg2(111500)
engs = [cases.configure(po_rrt_amd.Engine(0), case) for _ in range(Q)]
bufs = [(np.zeros((111502, 2)), np.zeros(111502, dtype=np.int64), np.zeros(111502)) for _ in range(Q)]
digest0 = None
for s in range(B):
    for j, e in enumerate(engs):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), (s % 5) * Q + j)         # seeds repeat every five batches: so must the trees
    t0 = time.perf_counter()
    po_rrt_amd.Engine.grow_batch(engs, [case.start] * Q, case.max_step, case.search_radius, case.n_iter_min, 1024)
    ms = 1e3 * (time.perf_counter() - t0)
    if s % 5 == 0:
        out = po_rrt_amd.Engine.trees(engs[:4], bufs[:4])
        d = hash(tuple(o[1].tobytes() for o in out))
        if digest0 is None:
            digest0 = d
        assert d == digest0, "batch %d: the same seeds gave other trees" % s
    if s % 10 == 0 or s == B - 1:
        free, total = torch.cuda.mem_get_info(0)
        print("batch %4d  %.2f ms  host maxrss %.0f MiB  device used %.0f MiB" % (s, ms, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, (total - free) / 2**20), flush=True)
print("soak ok")
