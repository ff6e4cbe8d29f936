"""Developer probe: single queries under forms of the tie order's upkeep -- the goal path's workgroup as a kernel of its own beside the step
kernel (the default), as a workgroup of the step kernel (gtrack_side=0), the whole kd structure beside the steps (kd_lazy=0): time, and how
often the kd structure was built after the steps (python tools/single_lazy_probe.py [queries + 1])"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases, po_rrt_amd
case = cases.cfg2(111500)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for lazy, side in ((1, 1), (1, 0), (0, 0)):
    e = cases.configure(po_rrt_amd.Engine(0), case)
    e.set_option("kd_lazy", lazy)
    e.set_option("gtrack_side", side)
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
            print("kd_lazy %d gtrack_side %d seed %d: %.2f ms, device %.2f ms, built after %d (steps %d)" % (lazy, side, seed, ms, 1e3 * e.metrics()["device_s"], e.get_option("kd_built_after"), e.get_option("kd_lca_steps")), flush=True)
    ms_all.sort()
    print("kd_lazy %d gtrack_side %d: %d queries, mean %.3f ms, median %.3f, max %.3f; the kd structure built after the steps in %d of them" % (
        lazy, side, len(ms_all), sum(ms_all) / len(ms_all), ms_all[len(ms_all) // 2], ms_all[-1], built), flush=True)
