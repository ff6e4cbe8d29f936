"""Developer probe: GPU tree against the oracle for one case, printing where they part (python tools/debug_parity.py)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import numpy as np
import cases
import po_rrt_amd
from oracle import orc

def run(case, K, **opts):
    e = po_rrt_amd.Engine()
    for k, v in opts.items():
        e.set_option(k, v)
    cases.configure(e, case)
    cases.grow(e, case, K=K)
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=K, algo=orc.ALGO_BATCHED_KD)
    xe, pe, de = e.tree()
    xo, po, do = o.tree()
    n = min(len(pe), len(po))
    bad = np.nonzero(pe[:n] != po[:n])[0]
    badd = np.nonzero(de[:n].view(np.uint64) != do[:n].view(np.uint64))[0]
    badx = np.nonzero((xe[:n].view(np.uint64) != xo[:n].view(np.uint64)).any(axis=1))[0]
    print(case.name, "K", K, opts, "nodes", len(pe), len(po), "bad parents", bad.size, "bad dist", badd.size, "bad xy", badx.size,
          "tie_fallbacks", e.metrics()["n_tie_fallbacks"])
    for i in bad[:6]:
        print("   node", i, "gpu parent", pe[i], "orc parent", po[i], "dist gpu/orc", de[i], do[i], "d(par) gpu", de[pe[i]] if pe[i] >= 0 else None,
              "orc", do[po[i]] if po[i] >= 0 else None)
    return bad.size + badd.size + badx.size

if __name__ == "__main__":
    tot = 0
    for gl in (64, 32, 16):
        for case in (cases.empty_space(1000, 10000), cases.cfg2(4000)):
            for K in (64, 1024):
                tot += run(case, K, group_lanes=gl)
    print("total mismatches", tot)
