import sys
sys.path[:0]=["tests","tools","."]
import numpy as np, cases, po_rrt_amd
from oracle import orc
cs=[cases.cfg2(30000,seed=s) for s in range(9)]
engs=[cases.configure(po_rrt_amd.Engine(),c) for c in cs]
for e in engs: e.set_option("kd_after",1)
po_rrt_amd.Engine.grow_batch(engs,[c.start for c in cs],0.1,2.0,30000,1024)
ok=True
for c,e in zip(cs,engs):
    o=cases.configure(orc.Oracle(),c); cases.grow(o,c,K=1024,algo=orc.ALGO_BATCHED_KD)
    ok &= np.array_equal(e.tree()[1],o.tree()[1]) and np.array_equal(e.tree()[2],o.tree()[2]) and e.metrics()["n_tie_fallbacks"]==0
print("kd_after parity", ok)
