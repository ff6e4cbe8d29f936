"""Developer probe: the single query with and without kd_lazy: time, whether the kd structure was built after the steps and for how many"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases, po_rrt_amd
case = cases.cfg2(111500)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for lazy in (2, 1):
    e = cases.configure(po_rrt_amd.Engine(0), case)
    e.set_option("kd_lazy", lazy)
    ms_all, built = [], 0
    for seed in range(776, 776 + n_seeds):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), seed)
        t0 = time.perf_counter()
        cases.grow(e, case, K=1024)
        ms = 1e3 * (time.perf_counter() - t0)
        if seed > 776:
            ms_all.append(ms)
            built += e.get_option("kd_built_after")
        if n_seeds <= 8 or e.get_option("kd_built_after"):
            print("kd_lazy %d seed %d: %.2f ms, device %.2f ms, built after %d (steps %d)" % (lazy, seed, ms, 1e3 * e.metrics()["device_s"], e.get_option("kd_built_after"), e.get_option("kd_lca_steps")), flush=True)
    ms_all.sort()
    print("kd_lazy %d: %d queries, mean %.3f ms, median %.3f, max %.3f; the kd structure built after the steps in %d of them" % (
        lazy, len(ms_all), sum(ms_all) / len(ms_all), ms_all[len(ms_all) // 2], ms_all[-1], built), flush=True)
