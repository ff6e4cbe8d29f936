"""The exchange of a query-sharded job through the C ABI (porrt_comm_*, porrt_exchange_best) on the GPU box: one rank (RCCL
refuses two ranks on one device), real ncclAllGather / ncclBroadcast calls, trees compared with the contexts' own."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


def test_exchange_best_per_map_one_rank():
    import po_rrt_amd
    from po_rrt_amd import sharding
    maps = "abc"
    cs = [cases.cfg2(25000, seed=q, grid="map_benchmark_like_%s" % maps[q % 3]) for q in range(8)]
    cs.append(cases.cfg2(300, seed=99, grid="map_benchmark_like_a"))       # map 3: too short to reach the goal -> unsolved
    engs = [cases.configure(po_rrt_amd.Engine(), c) for c in cs]
    po_rrt_amd.Engine.grow_batch(engs[:8], [c.start for c in cs[:8]], cs[0].max_step, cs[0].search_radius, cs[0].n_iter_min, 1024)
    cases.grow(engs[8], cs[8], K=64)
    map_ids = [q % 3 for q in range(8)] + [3]
    comm = sharding.make_comm(0)
    win = sharding.exchange_best_per_map(comm, engs, map_ids, 4)
    costs = [e.best_cost() for e in engs]
    for m in range(3):
        qs = [q for q in range(8) if q % 3 == m and costs[q] is not None]
        assert qs, "25000 iterations reach the goal on these maps"
        best = min(qs, key=lambda q: (costs[q], q))
        assert win[m]["rank"] == 0 and win[m]["cost"] == costs[best] and win[m]["n_nodes"] == engs[best].num_nodes()
        xy, parent, dist_root = comm.tree(m)
        exy, eparent, edist = engs[best].tree()
        assert np.array_equal(xy.view(np.uint64), exy.view(np.uint64)) and np.array_equal(parent, eparent)
        assert np.array_equal(dist_root.view(np.uint64), edist.view(np.uint64))
    assert costs[8] is None and win[3]["rank"] == -1 and win[3]["n_nodes"] == 0 and np.isinf(win[3]["cost"])
    # a second exchange on the same communicator (buffers reused), fewer contexts
    win2 = sharding.exchange_best_per_map(comm, engs[:3], map_ids[:3], 4)
    assert [w["rank"] for w in win2] == [0, 0, 0, -1]
    comm.close()


def test_bench_two_ranks_rehearsal():
    """`bench.py --gpus 2` end to end on this box: the launcher starts two ranks, both grow their queries, the timings are
    reduced over ranks and rank 0 prints the one line.  PORRT_BENCH_REHEARSE lets the ranks share a device when the box has one
    (process group gloo); RCCL then refuses the communicator on every rank alike (make_comm is all-or-none), the line says so
    and carries each map's local winner.  With two devices the exchange itself runs."""
    import json
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PORRT_BENCH_REHEARSE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--queries", "9", "--steps", "2", "--warmup", "1",
                          "--n-iter", "20000", "--no-single-query", "--no-profile"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rehearsal"] is True and d["steps"] == 2 and d["value"] > 0
    assert d["config"]["queries_per_step_per_gpu"] == 9
    win = d["config"]["exchange_winners"]
    assert len(win) == 9 and all(w["rank"] in (0, 1) and w["cost"] > 0 for w in win)
    if torch.cuda.device_count() < 2:
        assert "make_comm" in d["config"]["exchange_error"]
    else:
        assert not d["config"].get("exchange_error")
