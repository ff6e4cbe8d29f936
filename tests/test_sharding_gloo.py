"""N > 1 path on CPU: two gloo ranks shard queries round-robin, gather their per-map best entries and decide the winners
with the library's own host code (porrt_exchange_decide: step 2 of porrt_exchange_best).

There is no GPU in this container, and RCCL refuses two ranks on one device, so steps 1 and 3 of the exchange
(ncclAllGather / ncclBroadcast, csrc/porrt_exchange.hpp) run in tests/test_gpu_exchange.py on one rank and at N > 1 only in
the driver's scaling run.  The trees here come from the CPU oracle.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests"), os.path.join(%(root)r, "tools")]
    import numpy as np, torch, torch.distributed as dist
    import cases
    from oracle import orc
    from po_rrt_amd import sharding
    from po_rrt_amd.engine import BEST_ENTRY
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_maps, n_queries = 3, 9                        # query q plans on map q %% 3 with seed q
    mine = sharding.queries_of_rank(n_queries, rank, world)
    assert mine == list(range(rank, n_queries, world))
    table = np.zeros(n_maps, dtype=BEST_ENTRY)
    table["cost"], table["rank"] = np.inf, rank
    trees = {}
    for q in mine:                                   # independent queries: different maps and seeds
        m = q %% n_maps
        case = cases.cfg2(1500, seed=q, grid="map_benchmark_like_%%s" %% "abc"[m])
        o = cases.configure(orc.Oracle(), case)
        cases.grow(o, case, K=64, algo=orc.ALGO_BATCHED_KD)
        sol = o.best_solution()
        cost = sol[1] if sol is not None else float("inf")
        if cost < table["cost"][m]:
            table["cost"][m], table["n_nodes"][m] = cost, o.num_nodes()
            trees[m] = o.tree()
    gathered = [None] * world
    dist.all_gather_object(gathered, table.tobytes())
    entries = np.stack([np.frombuffer(g, dtype=BEST_ENTRY) for g in gathered])
    win = sharding.decide_from_gathered(entries)
    for m in range(n_maps):                          # the rule: first minimum of (cost, rank)
        exp = min(range(world), key=lambda r: (entries[r, m]["cost"], r))
        if not np.isfinite(entries[exp, m]["cost"]):
            exp = -1
        assert win[m] == exp, (m, win, entries)
    box = [trees.get(0) if win[0] == rank else None]
    dist.broadcast_object_list(box, src=int(win[0]) if win[0] >= 0 else 0)
    if win[0] >= 0:
        xy, parent, dr = box[0]
        assert len(parent) == entries[win[0], 0]["n_nodes"] and parent[0] == -1 and dr[0] == 0.0
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok winners", win.tolist())
""")


def test_two_ranks_shard_and_decide(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok winners") == 2


def test_decide_rules():
    """porrt_exchange_decide on hand-made tables: ties go to the lowest rank, unsolved maps have no winner, an entry without
    a tree never wins"""
    from po_rrt_amd import sharding
    from po_rrt_amd.engine import BEST_ENTRY
    e = np.zeros((3, 4), dtype=BEST_ENTRY)
    e["cost"] = [[2.0, 1.0, np.inf, 5.0], [1.5, 1.0, np.inf, 4.0], [1.5, 3.0, np.inf, 4.0]]
    e["n_nodes"] = [[10, 10, 0, 10], [10, 10, 0, 0], [10, 10, 0, 10]]
    e["rank"] = [[0] * 4, [1] * 4, [2] * 4]
    assert sharding.decide_from_gathered(e).tolist() == [1, 0, -1, 2]
