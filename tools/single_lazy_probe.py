"""Developer probe: the single query with and without kd_lazy: time, whether the kd structure was built after the steps and for how many"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases, po_rrt_amd
case = cases.cfg2(111500)
for lazy in (2, 1):
    e = cases.configure(po_rrt_amd.Engine(0), case)
    e.set_option("kd_lazy", lazy)
    for seed in range(776, 784):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), seed)
        t0 = time.perf_counter()
        cases.grow(e, case, K=1024)
        ms = 1e3 * (time.perf_counter() - t0)
        print("kd_lazy %d seed %d: %.2f ms, device %.2f ms, built after %d (steps %d)" % (lazy, seed, ms, 1e3 * e.metrics()["device_s"], e.get_option("kd_built_after"), e.get_option("kd_lca_steps")), flush=True)
