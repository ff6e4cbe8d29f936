"""Developer probe: the bench step as G concurrent porrt_grow_batch calls from G host threads (python tools/overlap_probe.py Q G)."""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases, po_rrt_amd
Q, G = int(sys.argv[1]), int(sys.argv[2])
case = cases.cfg2(111500)
engs = [cases.configure(po_rrt_amd.Engine(0), case) for _ in range(Q)]
groups = [engs[g::G] for g in range(G)]
def run(step):
    for j, e in enumerate(engs):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), step * Q + j)
    def work(gr):
        po_rrt_amd.Engine.grow_batch(gr, [case.start] * len(gr), case.max_step, case.search_radius, case.n_iter_min, 1024)
    ts = [threading.Thread(target=work, args=(gr,)) for gr in groups]
    for t in ts: t.start()
    for t in ts: t.join()
    return sum(e.num_nodes() - 1 for e in engs)
run(100); run(101)
t0 = time.perf_counter(); n = 0
for s in range(4): n += run(s)
dt = time.perf_counter() - t0
print("Q %d in %d groups: %.1f M expansions/s, %.1f ms per step" % (Q, G, n / dt / 1e6, 1e3 * dt / 4))
