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


def test_partition_covers_every_query_once():
    """the partition the exchange and bench.py's query -> map assignment rest on: q -> rank q mod world"""
    from po_rrt_amd import sharding
    for world in (1, 2, 3, 8):
        parts = [sharding.queries_of_rank(576, r, world) for r in range(world)]
        assert sorted(q for p in parts for q in p) == list(range(576))
        assert all(q % world == r for r, p in enumerate(parts) for q in p)
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert sharding.queries_of_rank(3, 5, 8) == [] and sharding.queries_of_rank(0, 0, 2) == []


def test_agree_rules():
    """porrt_exchange_agree (step 0 of porrt_exchange_best): every rank takes the same way out.  A failing rank returns its own
    code, the others PORRT_ERR_PEER (-8); different map counts are PORRT_ERR_INVALID (-1) for all; only a clean table goes on."""
    from po_rrt_amd import sharding
    ok = [(0, 9)] * 4
    assert [sharding.agree_from_gathered(ok, r) for r in range(4)] == [(0, -1)] * 4
    one_bad = [(0, 9), (0, 9), (-4, 9), (0, 9)]
    assert [sharding.agree_from_gathered(one_bad, r) for r in range(4)] == [(-8, 2), (-8, 2), (-4, 2), (-8, 2)]
    two_bad = [(-1, 9), (0, 9), (-4, 9), (0, 9)]
    assert [sharding.agree_from_gathered(two_bad, r)[0] for r in range(4)] == [-1, -8, -4, -8]
    maps_differ = [(0, 9), (0, 9), (0, 8)]
    assert [sharding.agree_from_gathered(maps_differ, r) for r in range(3)] == [(-1, 2)] * 3
    # a failing rank outranks a map-count mismatch (its n_maps may be garbage)
    assert sharding.agree_from_gathered([(0, 9), (-1, 0)], 0) == (-8, 1)
    assert sharding.agree_from_gathered([(0, 1)], 0) == (0, -1)


def test_a_rank_that_fails_inside_the_sequence_aborts_its_communicator():
    """porrt_exchange.hpp, comm_fail: before the first agreement a local failure is a status word -- the code comes back and the
    communicator stays usable; at or after it (inside comm_agree, the all-gather, between the agreements, the broadcast group, a
    wait that timed out) the rank aborts its communicator before returning, exactly once, and every later call is refused -- no
    rank is left inside a collective the failed rank never enters.  Stand-in communicators (no RCCL, no device): the function
    every failure of the real path goes through."""
    from po_rrt_amd import load_library
    L = load_library()
    for stage in (1, 2, 3, 4):
        c = L.porrt_comm_test_new(1, 4)
        assert c and L.porrt_comm_usable(c) == 1 and L.porrt_comm_test_aborts(c) == 0
        assert L.porrt_comm_test_fail(c, 0, -5) == -5                      # before the first agreement: nothing is torn down
        assert L.porrt_comm_usable(c) == 1 and L.porrt_comm_test_aborts(c) == 0
        assert L.porrt_comm_test_fail(c, stage, -4) == -4                  # inside the sequence: aborted
        assert L.porrt_comm_usable(c) == 0 and L.porrt_comm_test_aborts(c) == 1
        assert b"aborted" in L.porrt_comm_last_error(c)
        assert L.porrt_comm_test_fail(c, stage, -9) == -4                  # (refused: PORRT_ERR_DEVICE)
        assert L.porrt_comm_test_aborts(c) == 1                            # (once)
        # the real entry point refuses an aborted communicator before it touches a device or a peer
        import ctypes as C
        import numpy as np
        from po_rrt_amd.engine import BEST_ENTRY
        win = np.zeros(1, dtype=BEST_ENTRY)
        ids = np.zeros(1, dtype=np.uint32)
        assert L.porrt_exchange_best(c, (C.c_void_p * 1)(), 0, ids, 1, win.ctypes.data_as(C.c_void_p)) == -4
        assert b"make a new one" in L.porrt_comm_last_error(c)
        assert L.porrt_comm_set_timeout_ms(c, 0) == -1 and L.porrt_comm_set_timeout_ms(c, 5000) == 0
        L.porrt_comm_destroy(c)
    assert not L.porrt_comm_test_new(4, 4) and L.porrt_comm_usable(None) == 0


AGREE_WORKER = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [%(root)r]
    import torch.distributed as dist
    from po_rrt_amd import sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for case, word in enumerate([(0, 9), (-5 if rank == 1 else 0, 9), (0, 9 - rank)]):      # clean; rank 1 fails; map counts differ
        words = [None] * world
        dist.all_gather_object(words, word)                 # the status all-gather of comm_agree
        rc, bad = sharding.agree_from_gathered(words, rank)
        went_on = [None] * world
        dist.all_gather_object(went_on, rc == 0)
        assert all(went_on) or not any(went_on), "some ranks would enter the next collective and some would not"
        exp = [0, -5 if rank == 1 else -8, -1][case]
        assert rc == exp, (case, rank, rc)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "agree ok")
""")


def test_two_ranks_agree_before_every_collective(tmp_path):
    script = tmp_path / "agree_worker.py"
    script.write_text(AGREE_WORKER % {"root": ROOT})
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29534", str(script)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("agree ok") == 2


COMM_WORKER = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [%(root)r]
    import torch.distributed as dist
    from po_rrt_amd import sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Stub:                                              # stands in for engine.Comm (RCCL needs one GPU per rank)
        closed = 0
        def __init__(self, device, rank, world, uid):
            if FAIL == ("create", rank):
                raise RuntimeError("no communicator here")
            self.uid = uid
        def close(self):
            Stub.closed += 1

    def make_id():
        if FAIL == ("id", 0):
            raise RuntimeError("no unique id")
        return b"u" * 128

    for FAIL in (None, ("id", 0), ("create", 1), ("create", 0)):
        Stub.closed = 0
        try:
            c = sharding.make_comm(0, dist, _factory=(make_id, Stub))
            got = "comm"
            assert c.uid == b"u" * 128
        except RuntimeError as ex:
            got = str(ex)
        outcomes = [None] * world
        dist.all_gather_object(outcomes, got == "comm")
        assert all(outcomes) or not any(outcomes), "some ranks hold a communicator and some do not"
        assert (got == "comm") == (FAIL is None), (FAIL, got)
        if FAIL is not None:                                 # every rank names the failing rank; a made communicator is closed again
            assert "rank %%d" %% FAIL[1] in got, got
            assert Stub.closed == (1 if FAIL[0] == "create" and FAIL[1] != rank else 0), (FAIL, rank, Stub.closed)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "make_comm ok")
""")


def test_two_ranks_make_comm_all_or_none(tmp_path):
    """sharding.make_comm: a rank that cannot make the unique id or its communicator takes every rank out the same way"""
    script = tmp_path / "comm_worker.py"
    script.write_text(COMM_WORKER % {"root": ROOT})
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29536", str(script)],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("make_comm ok") == 2


def _bench(*args, env_drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True, text=True, timeout=600)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two rank processes itself (RANK / WORLD_SIZE / MASTER_* set,
    rendezvous on 127.0.0.1) and relays rank 0's one line; --launch-check stops after the rendezvous, so no GPU is needed"""
    import json
    out = _bench("--gpus", "2", "--launch-check")
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["launch_check"] is True and rec["n_gpus"] == 2 and rec["agree"] == 0
    for r in (0, 1):
        assert "launch-check rank %d of 2 local_rank %d" % (r, r) in out.stderr


def test_bench_refuses_more_gpus_than_visible():
    """without N devices the launcher says so and exits non-zero -- it never silently benchmarks one GPU"""
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    out = _bench("--gpus", str(n))
    assert out.returncode == 3 and out.stdout.strip() == "" and "HIP device(s) visible" in out.stderr


def test_bench_rank_checks_world_size():
    """a rank started by a launcher with another world size than --gpus stops at once"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=4" in out.stderr
