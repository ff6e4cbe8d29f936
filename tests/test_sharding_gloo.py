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


XCHG_WORKER = textwrap.dedent("""
    import ctypes as C, datetime, os, sys
    sys.path[:0] = [%(root)r]
    import numpy as np
    import torch
    import torch.distributed as dist
    from po_rrt_amd import load_library
    from po_rrt_amd.engine import BEST_ENTRY, TreeDeviceView
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=8))
    L = load_library()

    # ---- the transport: collectives over gloo on host memory; "device" buffers are numpy arrays kept alive here
    AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    BC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
    AL = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
    RL = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
    FE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
    AB = C.CFUNCTYPE(C.c_int, C.c_void_p)
    class Ops(C.Structure):
        _fields_ = [("self", C.c_void_p), ("all_gather", AG), ("broadcast", BC), ("alloc", AL), ("release", RL), ("fetch", FE), ("abort", AB)]
    live, calls = {}, {"all_gather": 0, "broadcast": 0, "abort": 0, "fail_all_gather_at": -1}
    def bytes_at(p, n):
        return torch.from_numpy(np.frombuffer(C.string_at(p, n), dtype=np.uint8).copy())
    def all_gather(_, send, recv, n):
        calls["all_gather"] += 1
        try:
            outs = [torch.empty(n, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(outs, bytes_at(send, n))
            for r in range(world):
                C.memmove(recv + r * n, outs[r].numpy().ctypes.data, n)
        except Exception as ex:                       # a peer that left: the collective times out
            sys.stderr.write("rank " + str(rank) + ": all_gather failed: " + str(ex)[:80] + "\\n")
            return 7
        return 9 if calls["all_gather"] == calls["fail_all_gather_at"] else 0
    def broadcast(_, send, recv, n, root):
        calls["broadcast"] += 1
        try:
            buf = bytes_at(send, n) if rank == root else torch.empty(n, dtype=torch.uint8)
            dist.broadcast(buf, root)
            C.memmove(recv, buf.numpy().ctypes.data, n)
        except Exception as ex:
            return 7
        return 0
    def alloc(_, n):
        a = np.zeros(max(n, 1), dtype=np.uint8)
        live[a.ctypes.data] = a
        return a.ctypes.data
    def release(_, p):
        live.pop(p, None)
    def fetch(_, dst, src, n):
        C.memmove(dst, src, n)
        return 0
    def abort(_):
        calls["abort"] += 1
        return 0
    ops = Ops(None, AG(all_gather), BC(broadcast), AL(alloc), RL(release), FE(fetch), AB(abort))
    comm = L.porrt_comm_test_new_ops(rank, world, C.byref(ops))
    assert comm and L.porrt_comm_usable(comm) == 1

    def tree(r, m, n):
        g = np.random.default_rng(1000 * r + m)
        return g.random(n), g.random(n), g.random(n), g.integers(-1, n, size=n).astype(np.int32)
    def exchange(table, n_maps, bad_view=False):
        ent = np.zeros(n_maps, dtype=BEST_ENTRY)
        views = (TreeDeviceView * n_maps)()
        keep = []
        for m in range(n_maps):
            cost, n = table.get(m, (np.inf, 0))
            ent[m] = (cost, -5, n)
            if n:
                nx, ny, d, p = tree(rank, m, n)
                keep.append((nx, ny, d, p))
                views[m] = TreeDeviceView(nx.ctypes.data, ny.ctypes.data, d.ctypes.data, p.ctypes.data, n + (1 if bad_view else 0))
        win = np.zeros(n_maps, dtype=BEST_ENTRY)
        rc = L.porrt_exchange_tables(comm, ent.ctypes.data, C.addressof(views), n_maps, win.ctypes.data)
        return rc, win

    # ---- a clean exchange: map 0 is won by rank 1 (rank 0 receives a tree it does not own), map 1 ties on cost (the lower rank keeps it),
    # map 2 has no solution anywhere
    mine = {0: {0: (5.0, 100), 1: (2.0, 50)}, 1: {0: (3.0, 70), 1: (2.0, 60)}}[rank]
    rc, win = exchange(mine, 3)
    assert rc == 0, (rc, L.porrt_comm_last_error(comm))
    assert [int(w["rank"]) for w in win] == [1, 0, -1] and [int(w["n_nodes"]) for w in win] == [70, 50, 0] and win["cost"][0] == 3.0
    for m, (r, n) in enumerate([(1, 70), (0, 50)]):
        assert L.porrt_exchange_num_nodes(comm, m) == n
        xy, parent, d = np.zeros((n, 2)), np.zeros(n, dtype=np.int64), np.zeros(n)
        assert L.porrt_exchange_get_tree(comm, m, xy, parent, d) == 0
        nx, ny, dd, pp = tree(r, m, n)
        assert np.array_equal(xy[:, 0], nx) and np.array_equal(xy[:, 1], ny) and np.array_equal(d, dd) and np.array_equal(parent, pp)
    assert L.porrt_exchange_num_nodes(comm, 2) == 0
    # ---- the ranks called with different numbers of maps: everybody out the same way, the communicator stays usable
    rc, _ = exchange(mine, 3 if rank == 0 else 2)
    assert rc == -1 and L.porrt_comm_usable(comm) == 1, rc
    # ---- rank 1 fails locally before the first agreement (a view that does not match its entry): its own code there, PEER on rank 0
    rc, _ = exchange(mine, 3, bad_view=(rank == 1))
    assert rc == (-1 if rank == 1 else -8) and L.porrt_comm_usable(comm) == 1, rc
    # ---- a second clean exchange with other winners: buffers are reused, larger trees re-allocated
    mine2 = {0: {0: (1.0, 300)}, 1: {1: (4.0, 20), 2: (9.0, 10)}}[rank]
    rc, win = exchange(mine2, 3)
    assert rc == 0 and [int(w["rank"]) for w in win] == [0, 1, 1] and L.porrt_exchange_num_nodes(comm, 0) == 300
    xy, parent, d = np.zeros((10, 2)), np.zeros(10, dtype=np.int64), np.zeros(10)
    assert L.porrt_exchange_get_tree(comm, 2, xy, parent, d) == 0 and np.array_equal(d, tree(1, 2, 10)[2])
    # ---- the transport fails on rank 1 INSIDE the sequence (the all-gather of the tables): rank 1 aborts its communicator and leaves;
    # rank 0's next collective then fails (here: gloo times out, as a torn-down RCCL connection would make it fail) and it aborts its own
    calls["fail_all_gather_at"] = calls["all_gather"] + 2 if rank == 1 else -1
    rc, _ = exchange(mine, 3)
    assert rc == -4 and L.porrt_comm_usable(comm) == 0 and calls["abort"] == 1, (rank, rc, calls)
    assert L.porrt_exchange_tables(comm, None, None, 3, None) == -4           # refused from now on
    print("rank", rank, "exchange ok", flush=True)
    os._exit(0)
""")


def test_two_ranks_run_the_whole_exchange_over_a_stand_in_transport(tmp_path):
    """porrt_exchange_tables -- the collective part of porrt_exchange_best: the agreements, the all-gather of the tables, the decision, the
    broadcasts from whichever rank wins (incl. to a rank that does not own the winner), the getters -- between TWO processes, on a transport
    the test brings (gloo over host memory; porrt_comm_test_new_ops).  RCCL itself is not exercised; everything above it is: clean
    exchanges, differing map counts, a rank failing before the first agreement, and a rank whose transport fails inside the sequence
    (it aborts its communicator; the other rank's next collective fails and it aborts its own)."""
    script = tmp_path / "xchg_worker.py"
    script.write_text(XCHG_WORKER % {"root": ROOT})
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)],
                         capture_output=True, text=True, timeout=300)
    assert out.stdout.count("exchange ok") == 2, out.stdout[-2000:] + out.stderr[-4000:]


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
