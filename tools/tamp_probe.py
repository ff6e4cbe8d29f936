"""Developer probe: the bench's TAMP row alone -- n TAMP-shaped queries in one porrt_grow_batch (python tools/tamp_probe.py [n] [K] [reps] [opt=val ...], or n:K)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases, po_rrt_amd
args = sys.argv[1:]
if args and ":" in args[0]:                      # (the older form: Q:K)
    q, k = args[0].split(":")
    args = [q, k] + args[1:]
n = int(args[0]) if len(args) > 0 else 1024
K = int(args[1]) if len(args) > 1 else 128
reps = int(args[2]) if len(args) > 2 else 4
cs = cases.tamp_queries(n)
engs = [cases.configure(po_rrt_amd.Engine(0), c) for c in cs]
for e in engs:
    for o in args[3:]:
        k, v = o.split("=")
        e.set_option(k, int(v))
starts = [c.start for c in cs]
for rep in range(reps):
    for j, e in enumerate(engs):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), 1000 * rep + j)
    t0 = time.perf_counter()
    po_rrt_amd.Engine.grow_batch(engs, starts, 0.1, 2.0, 2500, K, n_iter_max=10000)
    dt = time.perf_counter() - t0
    its = sum(e.num_iterations() for e in engs)
    print("rep %d: %.2f ms, %.0f queries/s, %d iterations (max %d per query), %d nodes, kd built after the steps: %d, compactions %d" % (
        rep, 1e3 * dt, n / dt, its, max(e.num_iterations() for e in engs), sum(e.num_nodes() - 1 for e in engs), engs[0].get_option("kd_built_after"),
        engs[0].get_option("compactions")), flush=True)
