"""N > 1 path on CPU: two gloo ranks shard queries round-robin and exchange the best tree.

The trees here come from the CPU oracle (there is no GPU in this container); the code under test is the
partition + the one collective step of po_rrt_amd/sharding.py, identical for "nccl" on the GPU node.
"""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [%(root)r, os.path.join(%(root)r, "tests"), os.path.join(%(root)r, "tools")]
    import numpy as np, torch, torch.distributed as dist
    import cases
    from oracle import orc
    from po_rrt_amd import sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.queries_of_rank(6, rank, world)
    assert mine == list(range(rank, 6, world))
    best = (float("inf"), None)
    for q in mine:                                   # independent queries: different seeds
        case = cases.cfg2(1500, seed=q)
        o = cases.configure(orc.Oracle(), case)
        cases.grow(o, case, K=64, algo=orc.ALGO_BATCHED_KD)
        sol = o.best_solution()
        cost = sol[1] if sol is not None else float("inf")
        if cost < best[0] or best[1] is None:
            best = (cost, o.tree())
    cost, (xy, parent, dr) = best
    w, wcost, wxy, wpar, wdr = sharding.exchange_best_tree(cost, xy, parent, dr, dist=dist)
    # every rank ends with the same winner and the same bytes
    allc = [None] * world
    dist.all_gather_object(allc, (cost, int(len(parent)), float(np.asarray(xy).sum())))
    exp = min(range(world), key=lambda r: (allc[r][0], r))
    assert w == exp and wcost == allc[exp][0] and len(wpar) == allc[exp][1]
    assert abs(float(wxy.sum()) - allc[exp][2]) == 0.0
    assert wpar[0] == -1 and wdr[0] == 0.0
    if rank == w:
        assert np.array_equal(wxy, xy) and np.array_equal(wpar, parent) and np.array_equal(wdr, dr)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok winner", w)
""")


def test_two_ranks_shard_and_exchange(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok winner") == 2


def test_partition_covers_every_query_once():
    from po_rrt_amd import sharding
    for world in (1, 2, 3, 8):
        seen = sorted(q for r in range(world) for q in sharding.queries_of_rank(576, r, world))
        assert seen == list(range(576))
