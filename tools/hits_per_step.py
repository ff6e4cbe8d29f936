"""Developer probe (GPU box): how many nodes lie within the search radius of a step's new nodes -- the work of the radius search and of
the connect pass per sample -- for the first steps of a configs[1] query (python tools/hits_per_step.py [steps] [K])."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import cases, po_rrt_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sizes = []
tree = None
for b in range(1, steps + 2):
    case = cases.cfg2(b * K)
    e = cases.configure(po_rrt_amd.Engine(0), case)
    cases.grow(e, case, K=K)
    sizes.append(e.num_nodes())
    if b == steps + 1:
        tree = e.tree()
xy = np.asarray(tree[0])
sizes = [1] + sizes                      # N before step b
print("step  N_before  radius   mean hits  p50   p99   max   share > 80")
for b in range(steps + 1):
    n0, n1 = sizes[b], sizes[b + 1]
    r = min(case.search_radius * math.sqrt(math.log(n0) / n0), case.max_step) if n0 > 1 else case.max_step
    new = xy[n0:n1]
    old = xy[:n0]
    cnt = np.zeros(len(new), dtype=np.int64)
    for i in range(0, len(new), 256):
        d = np.hypot(new[i:i + 256, None, 0] - old[None, :, 0], new[i:i + 256, None, 1] - old[None, :, 1])
        cnt[i:i + 256] = (d <= r).sum(axis=1)
    print("%4d  %8d  %.4f  %9.1f  %4d  %4d  %4d   %.3f" % (b, n0, r, cnt.mean(), np.percentile(cnt, 50), np.percentile(cnt, 99), cnt.max(), (cnt > 80).mean()), flush=True)
