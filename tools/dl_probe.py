"""Developer probe: where the time of fetching all trees of a batch goes (python tools/dl_probe.py [Q])."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import cases, po_rrt_amd
Q = int(sys.argv[1]) if len(sys.argv) > 1 else 128
case = cases.cfg2(111500)
engs = [cases.configure(po_rrt_amd.Engine(0), case) for _ in range(Q)]
for j, e in enumerate(engs):
    e.set_sampler((-1.0, -1.0), (1.0, 1.0), j)
po_rrt_amd.Engine.grow_batch(engs, [case.start] * Q, case.max_step, case.search_radius, case.n_iter_min, 1024)
for rep in range(3):
    t0 = time.perf_counter()
    sizes = [e.num_nodes() for e in engs]
    t1 = time.perf_counter()
    out = [(np.empty((m, 2)), np.empty(m, dtype=np.int64), np.empty(m)) for m in sizes]
    t2 = time.perf_counter()
    arr = (C.c_void_p * Q)(*[e._c for e in engs])
    ptrs = [(C.c_void_p * Q)(*[o[k].ctypes.data for o in out]) for k in range(3)]
    t3 = time.perf_counter()
    engs[0]._l.porrt_get_trees(arr, Q, ptrs[0], ptrs[1], ptrs[2])
    t4 = time.perf_counter()
    engs[0]._l.porrt_get_trees(arr, Q, ptrs[0], ptrs[1], ptrs[2])      # into memory that is already mapped
    t5 = time.perf_counter()
    print("sizes %.2f ms, np.empty %.2f ms, pointers %.2f ms, get_trees %.2f ms, again into the same arrays %.2f ms" %
          tuple(1e3 * d for d in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)), flush=True)
# the same into arrays handed to the device once (porrt_host_pin): one kernel writes the caller's layout over the link
cap = max(e.num_nodes() for e in engs)
bufs = [(np.zeros((cap, 2)), np.zeros(cap, dtype=np.int64), np.zeros(cap)) for _ in range(Q)]
t0 = time.perf_counter()
po_rrt_amd.Engine.pin_buffers(bufs)
print("pinning %d arrays (%.0f MB): %.1f ms" % (3 * Q, sum(a.nbytes for b in bufs for a in b) / 1e6, 1e3 * (time.perf_counter() - t0)), flush=True)
for blocks in (4, 16, 64):
    engs[0].set_option("tree_out_blocks", blocks)
    for rep in range(3):
        t0 = time.perf_counter()
        po_rrt_amd.Engine.trees(engs, bufs)
        dt = time.perf_counter() - t0
    print("pinned arrays, %d workgroups per tree: %.2f ms (%.1f GB/s over the link)" % (blocks, 1e3 * dt, sum(32.0 * e.num_nodes() for e in engs) / dt / 1e9), flush=True)
po_rrt_amd.Engine.unpin_buffers(bufs)
