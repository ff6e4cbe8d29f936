import os, sys, time, threading
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tools')
import cases, po_rrt_amd
case = cases.cfg2(111500)
for Q in [int(x) for x in os.environ.get("QS","1,2,3,4").split(",")]:
    engs = [cases.configure(po_rrt_amd.Engine(0), case) for _ in range(Q)]
    def work(e, seeds, out):
        n = 0
        for s in seeds:
            e.set_sampler((-1.0,-1.0),(1.0,1.0), s)
            cases.grow(e, case, K=1024)
            n += e.num_nodes() - 1
        out.append(n)
    # warmup
    outs=[]
    ths=[threading.Thread(target=work, args=(e,[1000+i],outs)) for i,e in enumerate(engs)]
    [t.start() for t in ths]; [t.join() for t in ths]
    R = 6
    outs=[]
    t0=time.perf_counter()
    ths=[threading.Thread(target=work, args=(e,[i*R+j for j in range(R)],outs)) for i,e in enumerate(engs)]
    [t.start() for t in ths]; [t.join() for t in ths]
    dt=time.perf_counter()-t0
    print("Q=%2d  %.1f M nodes/s  (%.2f ms per query per context)" % (Q, sum(outs)/dt/1e6, dt/R*1e3), flush=True)
    del engs
