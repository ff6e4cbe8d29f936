#!/usr/bin/env python3
"""Headline benchmark: RRT node-expansions/sec on the map_benchmark-like map (BASELINE.json configs[1]).

A "step" is one pass of the hot path over one batch of synthetic input: Q independent planning queries of
configs[1] per GPU (default Q = 256, --queries), each an RRT* tree grown with batch K=1024 samples per grow step
until n_iter iterations are spent (~100k-node tree) on the synthetic 200x200 map_benchmark stand-in (the
reference's raster is a Git-LFS pointer), all Q advanced together by porrt_grow_batch (one launch sequence, one
grid row per query: their dependent-load chains overlap inside every kernel).  A single query is latency bound
(8 ms); the TAMP caller issues thousands of independent ones, so throughput is what a GPU is for here.  The
single-query figure is measured too and reported in `config.single_query`.  Inputs (grid, tables) are resident in
HBM before the timed region.  `value` counts the device-to-host copy of every tree: a step's trees are fetched while
the next step grows on a second set of contexts; `value_trees_on_device` is the same loop with the trees left on the GPU.

  python bench.py --gpus N --steps K --warmup W
For N > 1 there is one rank per GPU.  Under a launcher (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) this process IS a rank; started bare (`python bench.py --gpus 8`) it starts the N rank processes itself, before
anything has touched a GPU, relays rank 0's JSON line and exits with the worst rank's code (launch_ranks below; it refuses
with exit code 3 when fewer than N devices are visible).  Every rank plans its own independent
queries (different RNG seeds, at N > 1 spread over the nine maps map_benchmark_like_{a..i}; no data-path collective:
weak scaling) and the job ends with ONE exchange behind the C ABI (porrt_exchange_best: ncclAllGather of the best path
cost per map and rank, ncclBroadcast of the winning trees).
Rank 0 prints one JSON line.  `roofline` is measured live with HIP events around the search and connect kernels;
`cpu_baseline` times the single-thread C restatement of the reference loop (oracle/) on the host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

T_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s


def launch_ranks(n, argv, check_only):
    """`bench.py --gpus N` without a launcher's environment: start the N rank processes (children of this one; this process
    never initialises a GPU -- torch.cuda.device_count() does not, on this image), rank 0's stdout is relayed, the others' goes to
    stderr.  Returns the exit code: 0, the first failing rank's code, or 3 when fewer than N devices are visible."""
    import socket
    import subprocess
    if not check_only and not os.environ.get("PORRT_BENCH_REHEARSE"):
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but %d HIP device(s) visible: not started\n" % (n, have))
            return 3
    with socket.socket() as sk:                       # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC (RCCL across processes)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    import threading
    box = []
    reader = threading.Thread(target=lambda: box.append(procs[0].stdout.read()), daemon=True)      # rank 0 writes its one line at the very end
    reader.start()
    code = 0
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0) and code == 0:
                code = rc if rc > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, rc))
                for q in procs:                        # a rank that died leaves the others in a collective: end exactly these children
                    if q.poll() is None:
                        q.terminate()
        time.sleep(0.1)
    for r, p in enumerate(procs):
        if p.returncode != 0 and code == 0:
            code = p.returncode if p.returncode > 0 else 1
            sys.stderr.write("bench.py: rank %d exited with code %d\n" % (r, p.returncode))
    reader.join(10)
    out0 = box[0] if box else b""
    if code == 0:
        sys.stdout.buffer.write(out0)
        sys.stdout.flush()
    return code


def launch_check(rank, local_rank, world, real_stdout):
    """--launch-check: what a rank does with the launcher's environment, without a GPU -- the gloo rendezvous, the partition of the
    queries, and the exchange's two host-side decisions on gathered words.  Used by the CPU tests of the N > 1 start-up."""
    import numpy as np
    import torch.distributed as dist
    from po_rrt_amd import sharding
    from po_rrt_amd.engine import BEST_ENTRY
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.queries_of_rank(4 * world + 1, rank, world)
    words = [None] * world
    dist.all_gather_object(words, (0, 9))
    rc, bad = sharding.agree_from_gathered(words, rank)
    table = np.zeros(9, dtype=BEST_ENTRY)
    table["cost"], table["rank"], table["n_nodes"] = 1.0 + ((rank + np.arange(9)) % world), rank, 10
    tables = [None] * world
    dist.all_gather_object(tables, table.tobytes())
    win = sharding.decide_from_gathered(np.stack([np.frombuffer(t, dtype=BEST_ENTRY) for t in tables]))
    dist.barrier()
    dist.destroy_process_group()
    sys.stderr.write("launch-check rank %d of %d local_rank %d queries %s agree %d\n" % (rank, world, local_rank, mine, rc))
    if rank == 0:
        os.write(real_stdout, (json.dumps({"launch_check": True, "n_gpus": world, "agree": rc, "winners": win.tolist()}) + "\n").encode())
    return 0 if rc == 0 and bad == -1 else 1


def note(msg):
    """progress on stderr (stdout carries the one JSON line): a long run stays visibly alive, and a stuck one says where"""
    sys.stderr.write("[bench %7.1f s] %s\n" % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-iter", type=int, default=111500, help="iterations per query (~100k-node tree)")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--queries", type=int, default=256, help="independent queries advanced together per step and GPU (on the device 244 M/s at 128, 253 at 256, 261 at 384, 264 at 512 at the end of round 4: 256 keeps the default run short and 95 GB of HBM in use)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event roofline pass")
    ap.add_argument("--no-single-query", action="store_true", help="skip the single-query (latency mode) reference run")
    ap.add_argument("--no-belief", action="store_true", help="skip the belief-space expansion measurement (SURVEY 8f.1)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="engine option for every context (porrt_set_option), e.g. group_lanes=32")
    ap.add_argument("--profile-steps", type=int, default=2, help="extra steps run with per-kernel HIP events for `roofline`")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, rendezvous over gloo and stop (no GPU needed)")
    ap.add_argument("--pin", action="store_true", help="fetch the trees by one kernel writing into the caller's arrays, pinned once (porrt_host_pin), instead of staging copies: "
                                                        "16 ms instead of 25 for 256 trees on an idle GPU, but 208 instead of 230 M/s beside a growing batch")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure the HBM traffic of the step kernels")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) one grow step of --queries queries and nothing else: what the counter passes profile")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.launch_check))
    # stdout carries the one JSON line and nothing else: RCCL prints a version banner there when the first communicator is made,
    # and any other library may follow -- everything this process writes to descriptor 1 goes to stderr, the line to the real one
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.launch_check:
        raise SystemExit(launch_check(rank, local_rank, world, real_stdout))

    import numpy as np
    import torch
    import torch.distributed as dist

    import cases
    import po_rrt_amd

    # PORRT_BENCH_REHEARSE=1: the N > 1 flow on a box with fewer GPUs than ranks (ranks share devices, the process group is gloo;
    # RCCL refuses two ranks on one device, so the exchange reports that and the line carries it).  Never a measurement: the
    # line says "rehearsal": true.
    rehearse = bool(os.environ.get("PORRT_BENCH_REHEARSE")) and world > 1
    if rehearse:
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    case = cases.cfg2(args.n_iter)
    Q = max(1, args.queries)
    if args.pmc_child:
        # profiled by rocprofv3 --pmc (measure_traffic below): the timed call once, nothing around it
        ce = [cases.configure(po_rrt_amd.Engine(local_rank), cases.Case(case, seed=j)) for j in range(Q)]
        for e in ce:
            for ov in args.opt:
                e.set_option(ov.split("=")[0], int(ov.split("=")[1]))
        po_rrt_amd.Engine.grow_batch(ce, [case.start] * Q, case.max_step, case.search_radius, case.n_iter_min, args.batch)
        os.write(real_stdout, b"{}\n")
        return
    # N = 1: every query plans on the map_benchmark stand-in (configs[1], the configuration the metric is quoted on).
    # N > 1 (configs[4]): the queries of a rank are spread over the nine maps map_benchmark_like_{a..i}; same tree size,
    # same parameters, so the work per GPU stays what it is at N = 1 (weak scaling); the exchange picks a winner per map.
    # Two sets of Q contexts: while one set grows, the trees of the other (the step before) are fetched to the host -- SURVEY 8(d)
    # counts the D2H copy of every tree, and a planner that wants every tree would overlap it the same way.
    if world > 1:
        n_maps = 9
        map_ids = [j % n_maps for j in range(Q)]
        sets = [[cases.configure(po_rrt_amd.Engine(local_rank), cases.cfg2(args.n_iter, grid="map_benchmark_like_%s" % "abcdefghi"[m])) for m in map_ids] for _ in range(2)]
    else:
        n_maps, map_ids = 1, [0] * Q
        sets = [[cases.configure(po_rrt_amd.Engine(local_rank), case) for _ in range(Q)] for _ in range(2)]
    engs = sets[0]
    eng = engs[0]
    from po_rrt_amd import sharding
    # RCCL communicator of the job's one exchange (outside the timed region).  The exchange is not the metric: if the communicator
    # cannot be made, or the exchange fails (collectively: every rank gets a code), the line still carries the growth's numbers and
    # says so in config.exchange_error -- the local best costs then stand in for the winners.
    exchange_error = None
    try:
        comm = sharding.make_comm(local_rank, dist if world > 1 else None)
    except Exception as ex:                  # noqa: BLE001
        comm, exchange_error = None, "communicator: %s" % ex
    for e in sets[0] + sets[1]:
        e.set_option("profile", 0)
        for ov in args.opt:
            e.set_option(ov.split("=")[0], int(ov.split("=")[1]))
    starts = [case.start] * Q
    # porrt_grow_batch advances the Q queries as G launch sequences side by side (option batch_streams: 2 from 32 queries on);
    # one kernel launch serves the Q_launch queries of one of them
    G = 0
    for ov in args.opt:
        if ov.split("=")[0] == "batch_streams":
            G = int(ov.split("=")[1])
    G = min(G if G else (2 if Q >= 32 else 1), Q)
    Q_launch = Q // G

    def run_step(s, which=0):
        """one step = Q queries; query ids (= RNG seeds) are unique over steps, ranks and slots"""
        for j, e in enumerate(sets[which]):
            e.set_sampler((-1.0, -1.0), (1.0, 1.0), (s * world + rank) * Q + j)
        po_rrt_amd.Engine.grow_batch(sets[which], starts, case.max_step, case.search_radius, case.n_iter_min, args.batch)
        return sum(e.num_nodes() - 1 for e in sets[which])

    # the caller's arrays for the trees are made once and reused, as a planner fetching trees query after query would (mapping
    # 410 MB of fresh pages costs three times the copies)
    cap = args.n_iter + 2
    bufs = [(np.zeros((cap, 2)), np.zeros(cap, dtype=np.int64), np.zeros(cap)) for _ in range(Q)]
    # --pin: ... and handed to the device once (porrt_host_pin): the fetch is then one kernel writing every tree into them in its final
    # layout.  Faster by itself (the link's 51 GB/s against 33), but the batch growing beside it loses more than the fetch gains, so the
    # staged path (copies into pinned staging, host threads laying the trees out) stays the bench's
    trees_pinned = args.pin
    if trees_pinned:
        try:
            po_rrt_amd.Engine.pin_buffers(bufs)
        except Exception as ex:              # noqa: BLE001
            sys.stderr.write("bench: the tree arrays could not be pinned (%s): staged fetch\n" % ex)
            trees_pinned = False
    import threading

    class Fetch(threading.Thread):
        """porrt_get_trees of one set (eight worker threads inside, pinned staging, copy streams of their own) beside the next step"""
        def __init__(self, which):
            super().__init__()
            self.which, self.err, self.seconds = which, None, 0.0

        def run(self):
            try:
                t1 = time.perf_counter()
                po_rrt_amd.Engine.trees(sets[self.which], bufs)
                self.seconds = time.perf_counter() - t1
            except Exception as ex:          # noqa: BLE001
                self.err = ex

    def timed_steps(first_seed, with_download):
        nodes, pending, dl_s = 0, None, 0.0
        for s in range(args.steps):
            which = s % 2
            nodes += run_step(first_seed + s, which)
            if with_download:
                if pending is not None:
                    pending.join()
                    dl_s += pending.seconds
                    if pending.err:
                        raise pending.err
                pending = Fetch(which)
                pending.start()
        if pending is not None:
            pending.join()
            dl_s += pending.seconds
            if pending.err:
                raise pending.err
        return nodes, dl_s

    note("contexts made; warm-up")
    for w in range(args.warmup):
        run_step(10_000 + w, w % 2)
    if args.warmup < 2:                       # both sets must have run once (buffers, streams, graphs), and the staging of the fetch exist
        run_step(20_000, 1 if args.warmup == 1 else 0)
        if args.warmup == 0:
            run_step(20_001, 1)
    po_rrt_amd.Engine.trees(sets[0], bufs)
    po_rrt_amd.Engine.trees(sets[1], bufs)

    agg = dict(nodes=0, device_s=0.0, setup_s=0.0)
    note("timed steps")
    barrier()
    t0 = time.perf_counter()
    agg["nodes"], t_download_total = timed_steps(0, True)
    t_loop = time.perf_counter() - t0
    engs = sets[(args.steps - 1) % 2]        # the set of the last step: its trees are what the exchange looks at
    eng = engs[0]
    # the one exchange of the job (porrt_exchange_best, behind the C ABI): per map, who holds the best tree?  Path costs are
    # evaluated on the device, 16 bytes per map and rank are all-gathered, the winning trees are broadcast device to device
    # (RCCL) and stay on the device; one of them -- the best of all maps -- is fetched to the host here.
    win = None
    note("timed steps done (%.3f s); exchange" % t_loop)
    if comm is not None:
        try:
            win = sharding.exchange_best_per_map(comm, engs, map_ids, n_maps)
        except Exception as ex:              # noqa: BLE001
            exchange_error = "exchange: %s" % ex
    if win is None:                          # this rank's own best trees per map (costs evaluated on the device all the same)
        from po_rrt_amd.engine import BEST_ENTRY
        costs = po_rrt_amd.Engine.best_cost_batch(engs)
        win = np.zeros(n_maps, dtype=BEST_ENTRY)
        win["cost"], win["rank"] = np.inf, -1
        for j, cj in enumerate(costs):
            if cj < win["cost"][map_ids[j]]:
                win["cost"][map_ids[j]], win["rank"][map_ids[j]], win["n_nodes"][map_ids[j]] = cj, rank, engs[j].num_nodes()
    solved = [m for m in range(n_maps) if win[m]["rank"] >= 0]
    mbest = min(solved, key=lambda m: (win[m]["cost"], m)) if solved else 0
    winner, win_cost = int(win[mbest]["rank"]), float(win[mbest]["cost"])
    win_nodes = 0
    if comm is not None and exchange_error is None:
        _, wparent, _ = comm.tree(mbest)
        win_nodes = len(wparent)
    barrier()
    elapsed = time.perf_counter() - t0
    note("timed region %.3f s; checks and the second loop (trees left on the device)" % elapsed)

    # outside the timed region: what the last overlapped fetch left in the caller's arrays against the contexts' own porrt_get_tree,
    # for the first and last member of each launch sequence (the fetch ran on a thread beside the growing batch of the other set;
    # tests/test_gpu_parity_r3.py::test_trees_fetched_beside_a_growing_batch checks every member and the oracle)
    fetch_checked = True
    for j in sorted({0, max(Q_launch - 1, 0), min(Q_launch, Q - 1), Q - 1}):
        exy, eparent, edist = engs[j].tree()
        m = len(eparent)
        fetch_checked = fetch_checked and m == engs[j].num_nodes() and np.array_equal(bufs[j][0][:m].view(np.uint64), exy.view(np.uint64)) \
            and np.array_equal(bufs[j][1][:m], eparent) and np.array_equal(bufs[j][2][:m].view(np.uint64), edist.view(np.uint64))
    if not fetch_checked:
        raise SystemExit("bench: a tree fetched beside the growing batch differs from its context's own porrt_get_tree")

    # the same steps with the trees left on the device (a caller that only downloads the winner): the second figure of the line
    barrier()
    t1 = time.perf_counter()
    nodes_dev, _ = timed_steps(50_000, False)
    barrier()
    elapsed_dev = time.perf_counter() - t1

    # max over ranks of the timed region, sum over ranks of the work
    if world > 1:
        t_el = torch.tensor([elapsed, elapsed_dev], dtype=torch.float64, device="cuda")
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        elapsed, elapsed_dev = float(t_el[0].item()), float(t_el[1].item())
        t_nodes = torch.tensor([float(agg["nodes"]), float(nodes_dev)], dtype=torch.float64, device="cuda")
        dist.all_reduce(t_nodes, op=dist.ReduceOp.SUM)
        total_nodes, total_nodes_dev = float(t_nodes[0].item()), float(t_nodes[1].item())
    else:
        total_nodes, total_nodes_dev = float(agg["nodes"]), float(nodes_dev)
    engs = sets[0]
    eng = engs[0]

    # latency mode for reference: the same query alone (porrt_grow, one context), outside the timed region
    single = None
    note("single query, roofline pass")
    if rank == 0 and not args.no_single_query:
        eng.set_sampler((-1.0, -1.0), (1.0, 1.0), 777)
        cases.grow(eng, case, K=args.batch)
        ts = []
        for r in range(3):
            eng.set_sampler((-1.0, -1.0), (1.0, 1.0), 778 + r)
            t1 = time.perf_counter()
            cases.grow(eng, case, K=args.batch)
            ts.append((time.perf_counter() - t1, eng.num_nodes() - 1))
        ts.sort()
        single = {"ms_per_query": 1e3 * ts[1][0], "node_expansions_per_s": ts[1][1] / ts[1][0]}

    # roofline pass: the same step again with HIP events around the search and connect kernels (eager launches on the
    # engine's own stream; the timed region above replays the steps as a hipGraph, which cannot carry events)
    prof = dict(scan_s=0.0, scan_pairs=0.0, scan_bytes=0.0, scan_launches=0, device_s=0.0, connect_s=0.0, nodes=0.0)
    if rank == 0 and not args.no_profile:
        eng.set_option("profile", 1)
        for s in range(args.profile_steps):
            run_step(s)
            prof["nodes"] += sum(e.num_nodes() - 1 for e in engs[:Q_launch])      # eng leads the first launch sequence: its metrics cover these
            m = eng.metrics()
            for k in ("scan_s", "scan_pairs", "scan_bytes", "device_s", "connect_s"):
                prof[k] += m[k]
            prof["scan_launches"] += m["scan_launches"]
        eng.set_option("profile", 0)

    if rank == 0:
        out = {
            "metric": "RRT node-expansions/sec on map_benchmark.pgm",
            "value": total_nodes / elapsed,
            "unit": "node-expansions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            **({"rehearsal": True} if rehearse else {}),
            "value_trees_on_device": total_nodes_dev / elapsed_dev,
            "value_note": "value counts the device-to-host copy of EVERY tree (SURVEY 8d): the %d trees of a step are fetched (porrt_get_trees, %.1f ms per "
                          "step on average) while the next step grows on a second set of contexts, and the last fetch is inside the timed region; "
                          "value_trees_on_device is the same loop with the trees left on the GPU (only the exchange's winner is fetched)"
                          % (Q, 1e3 * t_download_total / max(args.steps, 1)),
            "single_query": single,
            "config": {
                "workload": "map_benchmark-like 200x200 synthetic map (reference raster is a Git-LFS pointer), 2D RRT* "
                            "(rrt.rs grow_tree), batch K=%d samples/step, %d iterations -> ~%d-node tree per query, "
                            "max_step 0.1, search_radius 2.0, start (0,-1), SquareGoal (0.9,0) r=0.05; "
                            "%d independent queries per step per GPU (porrt_grow_batch), seeds differ"
                            % (args.batch, args.n_iter, agg["nodes"] // max(args.steps * Q, 1) + 1, Q),
                "batch_K": args.batch,
                "n_iter": args.n_iter,
                "queries_per_step_per_gpu": Q,
                "nodes_per_query": agg["nodes"] / max(args.steps * Q, 1),
                "ms_per_query": 1e3 * elapsed / max(args.steps * Q, 1),
                "single_query": single,
                "parallelism": "independent queries: %d per GPU advanced together x %d GPU(s); one exchange at the end behind the C ABI "
                               "(porrt_exchange_best: ncclAllGather of 16 B per map and rank + ncclBroadcast of the winning trees, device to device)" % (Q, world),
                "maps": "map_benchmark_like" if world == 1 else "map_benchmark_like_{a..i}, queries of a rank spread over the nine (configs[4])",
                "exchange_winners": [{"map": m, "rank": int(win[m]["rank"]), "cost": float(win[m]["cost"]), "nodes": int(win[m]["n_nodes"])} for m in range(n_maps)],
                "exchange_error": exchange_error,
                "best_path_cost": win_cost,
                "winner_rank": winner,
                "winner_nodes": win_nodes,
                "loop_s_rank0": t_loop,
                "launch_sequences": G,
                "launch_mode": sets[0][0].get_option("launch_mode"),
                "launch": ("%d launch sequences side by side (porrt_grow_batch, option batch_streams), on streams chosen by measurement to sit on "
                           "different hardware queues, launched step by step from a host thread each; the tie order comes from the goal path of "
                           "the kd-tree, tracked inside the connect kernel (kd_lazy), the whole structure is built after the steps for the rows a tie asks for" % G) if G > 1 else
                          "hipGraph replay of all steps (search, connect; the tie order from the goal path of the kd-tree, tracked inside the connect kernel)",
                "kd_lazy": sets[0][0].get_option("kd_lazy"),
                "fetch_checked": fetch_checked,
                "trees_fetched_into_pinned_arrays": trees_pinned,
                "kd_built_after_the_steps": [e.get_option("kd_built_after") for e in (sets[0][0], sets[0][Q_launch] if G > 1 else sets[0][0])],
            },
        }
        if prof["scan_s"] > 0:
            L = max(prof["scan_launches"], 1)
            # SURVEY 8(d), per step and query: B_alg = 16 N_b (node x, y) + 8 N_b (dist_root, read by the connect phase)
            # + 36 K (samples in; nn id / distance / new state out) + 28 K_valid (committed nodes) + W H (the raster).
            # A step is two kernels: k_nn2 (nearest neighbour, steer, validity) takes the 16 N_b + 36 K, k_conn2 (radius search,
            # raycasts, best parent, commit) the 8 N_b + W H + 28 K_valid -- it reads the coordinates a second time (the radius
            # search runs around the steered state), which is real traffic, not algorithmic bytes.  All figures per launch =
            # summed over the Q_launch queries of a launch, N_b averaged over the run's steps.
            grid_bytes = 200.0 * 200.0
            n_sum = prof["scan_pairs"] / (2.0 * args.batch * L)          # sum over the launch's queries of N_b
            nn_bytes = 16.0 * n_sum + 36.0 * args.batch * Q_launch
            conn_bytes = 8.0 * n_sum + Q_launch * grid_bytes + 28.0 * prof["nodes"] / L
            nn_us, conn_us = 1e6 * prof["scan_s"] / L, 1e6 * prof["connect_s"] / L
            pm, pm_src = {}, None
            if not args.no_pmc and world == 1:               # (N > 1: the other ranks must not wait for rank 0's counter passes)
                note("counter passes (two rocprofv3 --pmc child runs)")
                pm, pm_src = measure_traffic(args, Q)
            if not pm:
                for cand in ("r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"):      # committed rocprofv3 --pmc passes
                    try:
                        pm = json.load(open(os.path.join(ROOT, "profiles", cand)))
                        pm_src = "profiles/" + cand + " (committed; not measured in this run)"
                        break
                    except Exception:
                        pass

            def traffic_of(prefix):
                ks = [pm[k] for k in pm if k.startswith(prefix + "<16") or k == prefix] or [pm[k] for k in pm if k.startswith(prefix)]
                if not ks:
                    return None
                return sum(2.0 * k["fetch_bytes_per_launch_raw"] + k["write_bytes_per_launch"] for k in ks) / len(ks)

            dom = "k_conn2" if conn_us >= nn_us else "k_nn2"
            dom_bytes, dom_us = (conn_bytes, conn_us) if dom == "k_conn2" else (nn_bytes, nn_us)
            achieved = dom_bytes / (dom_us * 1e-6) / 1e9
            step_bytes, step_us = nn_bytes + conn_bytes, nn_us + conn_us
            # FP64 work of SURVEY 8(d): 6 flop per (sample, node) pair and search kind, 12 K N_b per step; peak = the measured
            # v_fma_f64 issue rate of tools/valu_peak.hip on this GPU (profiles/r2_valu_peak.txt), 2 flop per lane-instruction
            flops = 6.0 * prof["scan_pairs"] / L
            valu_peak = None
            try:
                for ln in open(os.path.join(ROOT, "profiles", "r2_valu_peak.txt")):
                    if "fma_f64" in ln and "blocks 4096" in ln:
                        valu_peak = 2.0 * float(ln.split("fma_f64")[1].split()[0])
            except Exception:
                pass
            out["roofline"] = {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic_of(dom),
                "traffic_source": pm_src,
                "traffic_note": "HBM bytes per launch = 2 x FETCH_SIZE (gfx950 counts a 128-B request as 64 B for wide streaming reads; for gathers "
                                "the raw figure may be the truer one) + WRITE_SIZE, each from its own rocprofv3 --pmc pass (they do not fit one) over one grow "
                                "step of the same queries, run as child processes of this bench after the timed region.  Kernels are SERIALISED while "
                                "counters are collected, avg_launch_us is measured under the overlap of the launch sequences: traffic is bytes per launch, "
                                "not to be divided by avg_launch_us for a bandwidth (the kernel alone takes ~440 us where the overlapped figure is ~690)",
                "traffic_raw": {k: pm[k] for k in pm if k.startswith("k_conn2") or k.startswith("k_nn2")} if pm else None,
                "avg_launch_us": dom_us,
                "launches": L,
                "queries_per_launch": Q_launch,
                "launch_sequences_side_by_side": G,
                "algorithmic_bytes_per_launch": dom_bytes,
                "formula": "SURVEY 8(d): k_nn2 16 N_b + 36 K, k_conn2 8 N_b + W H + 28 K_valid, per query, summed over the queries of a launch",
                "note": "One launch serves Q / G queries of the step; the G launch sequences run side by side (avg_launch_us is measured under that overlap, on the first one's stream).  The step kernels are bound by dependent-load latency at the occupancy their "
                        "registers and LDS allow, not by bandwidth: the searches touch only the region pages a query disc meets.",
                "kernels": {
                    "k_nn2": {"avg_launch_us": nn_us, "algorithmic_bytes_per_launch": nn_bytes, "achieved_GBs": nn_bytes / (nn_us * 1e-6) / 1e9,
                              "traffic": traffic_of("k_nn2")},
                    "k_conn2": {"avg_launch_us": conn_us, "algorithmic_bytes_per_launch": conn_bytes, "achieved_GBs": conn_bytes / (conn_us * 1e-6) / 1e9,
                                "reread_of_coordinates_bytes": 16.0 * n_sum, "traffic": traffic_of("k_conn2")},
                },
                "step": {"algorithmic_bytes": step_bytes, "us": step_us, "achieved_GBs": step_bytes / (step_us * 1e-6) / 1e9,
                         "frac": step_bytes / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS},
                "fp64_valu": {"brute_force_flops_per_step": flops, "brute_force_equivalent_TFLOPs": flops / (step_us * 1e-6) / 1e12,
                              "peak_TFLOPs": valu_peak, "ratio_to_peak": (flops / (step_us * 1e-6) / 1e12 / valu_peak) if valu_peak else None,
                              "note": "SURVEY 8(d)'s FP64 figure: 6 flop per (sample, node) pair and search kind, 12 K N_b per step -- the work of the brute-force "
                                      "DEFINITION of the two searches -- over the time of both step kernels; peak = measured v_fma_f64 rate "
                                      "(tools/valu_peak.hip, profiles/r2_valu_peak.txt).  The searches are exact but only visit the region pages a query disc "
                                      "meets (about 0.3 % of the pairs), so a ratio above 1 says how much of that work is avoided, not how busy the VALUs are "
                                      "(they are not the bound: DESIGN.md section 6)"},
                "share_of_device_time": (prof["scan_s"] + prof["connect_s"]) / max(prof["device_s"], 1e-12),
            }
        if not args.no_cpu_baseline and world == 1:
            note("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(case, args)
        if not args.no_belief and world == 1:
            # the rows after the growth are extras of the line: a failure there must not cost the headline measurement
            for key, fn in (("tamp_queries", tamp_queries), ("map4_pomdp", map4_pomdp), ("belief_space", belief_space), ("prm_roadmap", prm_roadmap), ("mm_prm", mm_prm)):
                note("row " + key)
                try:
                    out["config"][key] = fn(local_rank, not args.no_cpu_baseline)
                except Exception as ex:                      # noqa: BLE001
                    out["config"][key] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        note("done")
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


def measure_traffic(args, Q, timeout_s=240):
    """HBM traffic of the step kernels, measured in this run: two child processes of this script under rocprofv3 --pmc -- FETCH_SIZE,
    then WRITE_SIZE (MI355X_MICROARCH.md: they do not fit one pass; KiB units) -- each profiling one grow step of the same Q queries
    (--pmc-child).  Returns ({kernel: {launches, fetch_bytes_per_launch_raw, write_bytes_per_launch}}, source) or ({}, None)."""
    import collections
    import csv
    import shutil
    import signal
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return {}, None
    acc = collections.defaultdict(dict)
    for counter, key in (("FETCH_SIZE", "fetch_bytes_per_launch_raw"), ("WRITE_SIZE", "write_bytes_per_launch")):
        d = tempfile.mkdtemp(prefix="porrt_pmc_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
               "--queries", str(Q), "--n-iter", str(args.n_iter), "--batch", str(args.batch)] + [x for ov in args.opt for x in ("--opt", ov)]
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)          # exactly the child's own process group
                p.wait()
                rc = -9
            if rc != 0:
                return {}, None
            files = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
            if not files:
                return {}, None
            agg = collections.defaultdict(lambda: [0, 0.0])
            for row in csv.DictReader(open(files[0])):
                k = row["Kernel_Name"].split("(")[0].split("::")[-1]
                agg[k][0] += 1
                agg[k][1] += float(row["Counter_Value"])
            for k, (n, v) in agg.items():
                acc[k]["launches"] = n
                acc[k][key] = v / n * 1024.0
        except Exception:                                 # noqa: BLE001
            return {}, None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {k: v for k, v in acc.items() if "fetch_bytes_per_launch_raw" in v and "write_bytes_per_launch" in v}
    return out, "measured in this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one pass each, %d queries, one grow step)" % Q


def belief_traffic():
    """HBM bytes of one belief-graph build from the committed counter passes (tools/profile_belief.sh: two builds of this size)"""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r1_belief_pmc_traffic.json")))
    except Exception:
        return None
    ks = [v for k, v in pm.items() if k.startswith("k_bg_") or k.startswith("k_scan_")]
    builds = max(1, min((v["launches"] for k, v in pm.items() if k.startswith("k_bg_fill")), default=2))
    return sum((2.0 * v["fetch_bytes_per_launch_raw"] + v["write_bytes_per_launch"]) * v["launches"] for v in ks) / builds if ks else None


def dp_traffic():
    """HBM bytes of one expected-costs computation (all k_dp_* launches: 2 x FETCH_SIZE + WRITE_SIZE) from the committed counter passes
    (tools/pmc_dp.sh on the same 4095-belief graph with this round's sweep kernels: profiles/r4_dp_pmc_traffic.json)"""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r4_dp_pmc_traffic.json")))
        return float(pm["_all"]["hbm_bytes_2x_fetch_plus_write"])
    except Exception:
        return None


def belief_space(device, with_cpu):
    """The rows after the growth, outside the timed region: PTO::plan_belief_space (pto.rs:151-183) on the
    12-shelf problem of main.rs:386-408 -- a PTO graph of 20000 iterations expanded over the 4095 reachable beliefs
    (build_belief_graph), expected costs to the goals (conditional_dijkstra), policy extraction."""
    import cases
    import po_rrt_amd
    n_iter = 20000
    case = cases.cfg4(n_iter, n_iter)
    case.update(start=(0.0, -0.3))
    e = cases.configure(po_rrt_amd.Engine(device), case)
    cases.grow(e, case, K=256)
    prior = [1.0 / 12] * 12
    runs = []
    for _ in range(4):
        t0 = time.perf_counter()
        e.build_belief_graph(prior)
        runs.append((time.perf_counter() - t0, e.bg_seconds()))
    first = runs[0][0]
    same_graph = sorted(r[0] for r in runs[1:])[1]         # rebuilt on the same graph: its adjacency lists are kept as well
    fresh = []
    for _ in range(3):                                     # the usual case: a new graph, the prior already known to the context
        cases.grow(e, case, K=256)                         # (the sampler moved on: another graph of the same size)
        t0 = time.perf_counter()
        e.build_belief_graph(prior)
        fresh.append((time.perf_counter() - t0, e.bg_seconds()))
    wall, sec = sorted(fresh, key=lambda r: r[0])[1]
    E, N = e.bg_num_edges(), e.num_nodes()
    nb = N * 4095
    list_bytes = 2 * 4.0 * E + 2 * 8.0 * (nb + 1) + nb     # both id arrays, both offset arrays, the node types
    dps = []
    for _ in range(2):
        t0 = time.perf_counter()
        e.compute_expected_costs()
        dps.append(time.perf_counter() - t0)
    info = e.dp_info()
    t0 = time.perf_counter()
    (oid, par, leaf), root_cost = e.extract_policy()
    t_policy = time.perf_counter() - t0
    out = {
        "what": "PTO::plan_belief_space: %d graph nodes x 4095 beliefs (12 shelves, uniform prior)" % N,
        "belief_nodes": nb, "edges": E,
        "build_belief_graph": {
            "ms_wall": 1e3 * wall, "ms_first_build_with_this_prior": 1e3 * first, "ms_rebuild_on_the_same_graph": 1e3 * same_graph,
            "ms_device": 1e3 * sec["device_s"],
            "ms_host_tables": 1e3 * sec["host_tables_s"], "edges_per_s": E / wall,
            "roofline": {"bound": "hbm", "kernel": "k_bg_fill + k_bg_*_count + k_scan_*", "achieved": list_bytes / sec["device_s"] / 1e9,
                         "peak": 8000.0, "unit": "GB/s", "frac": list_bytes / sec["device_s"] / 1e9 / 8000.0, "algorithmic_bytes": list_bytes,
                         "traffic": belief_traffic(),
                         "note": "bytes of the result (CSR ids, offsets, types) over the device time of all belief kernels (HIP events); "
                                 "traffic = HBM bytes of one build (2 x FETCH_SIZE + WRITE_SIZE summed over the k_bg_* and k_scan_* launches, "
                                 "separate --pmc passes on the same graph, profiles/r1_belief_pmc_traffic.json)"}},
        "expected_costs": {"ms_wall": 1e3 * min(dps), "ms_device": 1e3 * info["device_s"], "sweeps": info["sweeps"], "root_cost": root_cost,
                           "roofline": {"bound": "hbm", "kernel": "k_dp_level_sweep", "achieved": (8.0 * E + 16.0 * nb) / info["device_s"] / 1e9, "peak": 8000.0,
                                        "unit": "GB/s", "frac": (8.0 * E + 16.0 * nb) / info["device_s"] / 1e9 / 8000.0,
                                        "algorithmic_bytes": 8.0 * E + 16.0 * nb, "sweep_rows": info["sweep_rows"], "traffic": dp_traffic(),
                                        "traffic_source": "profiles/r4_dp_pmc_traffic.json (committed: separate --pmc passes of FETCH_SIZE and WRITE_SIZE over one "
                                                          "computation on the same 4095-belief graph with these kernels, all k_dp_* launches summed; not measured in this run)",
                                        "note": "algorithmic = every edge of the belief graph relaxed ONCE (the neighbour's cost, 8 B) and every row's cost read and "
                                                "written once (16 B): what any method must move; the sweeps re-evaluate a row three times on average (%d sweeps), "
                                                "which is why the measured traffic is ~5x that.  Until round 3 this block counted 10 B per row and sweep -- the flag "
                                                "pass of kernels that no longer make one; over the device time of the whole computation (HIP events)" % info["sweeps"]},
                           "edge_relaxations_per_s_lower_bound": E / min(dps),
                           "note": "conditional_dijkstra as sweeps to the same fixpoint; every edge is relaxed at least once"},
        "extract_policy": {"ms_wall": 1e3 * t_policy, "policy_nodes": int(len(oid)), "leafs": int(leaf.sum())},
    }
    if with_cpu:
        from oracle import orc
        o = cases.configure(orc.Oracle(), case)
        cases.grow(o, case, K=256, algo=orc.ALGO_BATCHED_KD)
        sample_prior = [0.125] * 8 + [0.0] * 4            # 255 beliefs: 1/16 of the workload, same graph
        t0 = time.perf_counter()
        o.build_belief_graph(sample_prior)
        dt = time.perf_counter() - t0
        Eo = int(o._l.orc_bg_num_edges(o._c))
        t0 = time.perf_counter()
        d = o.expected_costs()
        dt2 = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": Eo / dt, "unit": "edges/s", "cores": 1, "kind": "port",
                               "sample": "the same PTO graph with 8 of the 12 worlds possible (255 beliefs, %d edges): build_belief_graph %.2f s, "
                                         "conditional_dijkstra %.2f s (%.1f M edges/s; root cost %r); C restatement of pto.rs:185-275, "
                                         "belief_graph.rs:89-175 (oracle/belief.c, oracle/dp.c)" % (Eo, dt, dt2, Eo / dt2 / 1e6, float(d[0]))}
    return out


def map4_pomdp(device, with_cpu):
    """The reference's OWN recorded problem, outside the timed region: test_plan_on_navigation_map4_pomdp (main.rs:893-908) on the one
    raster of the reference that is recoverable here (paper_map_4.pgm, embedded in data/maps_paper/map_4/map.svg; zone k = the door of
    map_door_k.svg): PTO::grow_graph(start (0.8, -0.8), goal (-0.8, 0.8) in all 16 worlds, max_step 0.1, search_radius 5, n_iter_min 5000,
    n_iter_max 100000), plan_belief_space(uniform prior over the 16 worlds) = build_belief_graph + conditional_dijkstra + extract_policy.
    Per phase, the median of five seeds, at K = 1 (the reference's loop, one sample per step) and at K = 256; the C restatement beside
    it; the reference's own recorded timings as context (results/maps_paper/map_4/costs_and_timings_5000_20.txt:2-4, CPU not stated)."""
    import numpy as np
    import cases
    import po_rrt_amd
    prior = [1.0 / 16] * 16
    out = {"what": "main.rs:893-908 on paper_map_4.pgm: PTO growth (n_iter_min 5000) -> belief graph over the reachable beliefs of 16 worlds -> expected "
                   "costs -> policy; ms per phase, median of 5 seeds",
           "reference_recorded_ms": {"graph_creation": 61.574401, "belief_expansion": 538.1212920999999, "dynamic_programming": 270.77340795000003,
                                     "total_with_refinement": 873.5060905999999, "cost_x_7.65": 43.990279797576896,
                                     "source": "results/maps_paper/map_4/costs_and_timings_5000_20.txt (30 true-random runs, refined policy, CPU not stated): context, not a baseline"}}
    e = po_rrt_amd.Engine(device)
    for K in (1, 256):
        rows = []
        for seed in range(6):                 # (the first run carries the one-off allocations and the belief tables of the prior: dropped)
            case = cases.cfg_map4(5000, seed)
            cases.configure(e, case)
            t0 = time.perf_counter()
            cases.grow(e, case, K=K)
            t1 = time.perf_counter()
            e.build_belief_graph(prior)
            t2 = time.perf_counter()
            e.compute_expected_costs()
            t3 = time.perf_counter()
            root_cost = e.expected_cost_of(0)
            try:
                (oid, par, leaf), _ = e.extract_policy()
                t4, n_pol = time.perf_counter(), len(oid)
            except po_rrt_amd.engine.PorrtError:          # (the walk of belief_graph.rs:193-213 does not end on this graph: reported, not timed)
                t4, n_pol = float("nan"), -1
            rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t3 - t0, e.num_iterations(), e.num_nodes(), e.bg_num_edges(), 7.65 * root_cost, n_pol))
        a = np.array(rows[1:], dtype=np.float64)
        r = np.median(a, axis=0)
        ok = a[:, 9] >= 0
        out["K=%d" % K] = {"ms_growth": 1e3 * r[0], "ms_belief_expansion": 1e3 * r[1], "ms_expected_costs": 1e3 * r[2],
                            "ms_policy": 1e3 * float(np.median(a[ok, 3])) if ok.any() else None,
                            "ms_total_without_policy": 1e3 * r[4], "iterations": r[5], "graph_nodes": r[6], "belief_graph_edges": r[7], "cost_x_7.65": r[8],
                            "policy_nodes": float(np.median(a[ok, 9])) if ok.any() else None,
                            "seeds_whose_policy_walk_does_not_terminate": int((~ok).sum())}
    e.close()
    if with_cpu:
        from oracle import orc
        rows = []
        for seed in range(1, 4):
            case = cases.cfg_map4(5000, seed)
            o = cases.configure(orc.Oracle(), case)
            t0 = time.perf_counter()
            cases.grow(o, case, K=1, algo=orc.ALGO_SEQ)
            t1 = time.perf_counter()
            o.build_belief_graph(prior)
            t2 = time.perf_counter()
            d = o.expected_costs()
            t3 = time.perf_counter()
            rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0, 7.65 * float(d[0])))
        r = np.median(np.array(rows), axis=0)
        out["cpu_baseline"] = {"value": 1e3 * r[3], "unit": "ms per plan (growth + belief expansion + expected costs)", "cores": 1, "kind": "port",
                               "ms_growth": 1e3 * r[0], "ms_belief_expansion": 1e3 * r[1], "ms_expected_costs": 1e3 * r[2], "cost_x_7.65": r[4],
                               "sample": "the full problem, seeds 1..3, median; C restatement of pto.rs:55-139,185-275, belief_graph.rs:89-175 (the reference's "
                                         "own sequential loop with its kd-tree)"}
    return out


def tamp_queries(device, with_cpu, n_queries=1024, K=128, opts=()):
    """The reference's real many-query caller, outside the timed region: the TAMP search plans twice per search edge with
    rrt.plan(.., max_step 0.1, search_radius 2.0, n_iter_min 2500, n_iter_max 10000) -- an ObservationGoal and a pickup SquareGoal,
    starts of their own (map_shelves_tamp_rrt.rs:224,232; main.rs:532) -- thousands of independent queries.  Here: n_queries of them
    in ONE porrt_grow_batch call, every member running the loop of rrt.rs:109 on its own and dropping out when it ends."""
    import numpy as np
    import cases
    import po_rrt_amd
    cs = cases.tamp_queries(n_queries)
    t0 = time.perf_counter()
    engs = [cases.configure(po_rrt_amd.Engine(device), c) for c in cs]
    t_make = time.perf_counter() - t0
    for e in engs:
        for name, val in opts:
            e.set_option(name, val)
    starts = [c.start for c in cs]
    runs = []
    for rep in range(4):
        for j, e in enumerate(engs):
            e.set_sampler((-1.0, -1.0), (1.0, 1.0), 1000 * rep + j)
        t0 = time.perf_counter()
        po_rrt_amd.Engine.grow_batch(engs, starts, 0.1, 2.0, 2500, K, n_iter_max=10000)
        dt = time.perf_counter() - t0
        # SURVEY 8(d)'s bytes of both step kernels per row and step: 24 N_b + 36 K + 28 K_valid + W H; a row's N_b taken as growing linearly
        # to its final size over its own steps (the tree sizes per step stay on the device)
        alg = sum((-(-e.num_iterations() // K)) * (24.0 * 0.5 * e.num_nodes() + 36.0 * K + 200.0 * 200.0) + 28.0 * (e.num_nodes() - 1) for e in engs)
        runs.append((dt, sum(e.num_nodes() - 1 for e in engs), sum(e.num_iterations() for e in engs), sum(1 for e in engs if e.num_final() > 0), alg,
                     sum(e.get_option("compactions") for e in (engs[0], engs[len(engs) // 2]))))
    dt, nodes, its, solved, alg, ncomp = sorted(runs[1:])[1]
    costs = po_rrt_amd.Engine.best_cost_batch(engs)
    for e in engs:                      # (a thousand contexts: freed here, not by the collector in the middle of the next row)
        e.close()
    out = {"what": "%d TAMP-shaped RRT* queries in one porrt_grow_batch (alternating ObservationGoal / SquareGoal, per-query starts, n_iter_min 2500, "
                   "n_iter_max 10000, K = %d): each runs the loop of rrt.rs:109 and leaves the launches when it ends" % (n_queries, K),
           "ms_wall": 1e3 * dt, "queries_per_s": n_queries / dt, "node_expansions_per_s": nodes / dt, "iterations_per_s": its / dt,
           "mean_iterations_per_query": its / n_queries, "queries_solved": solved, "queries_with_a_path_cost": int(np.isfinite(costs).sum()),
           "ms_creating_the_contexts_once": 1e3 * t_make, "batch_K": K, "row_compactions": ncomp,
           "roofline": {"bound": "hbm", "kernel": "k_nn2 + k_conn2 (both step kernels of every step of the batch)", "achieved": alg / dt / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": alg / dt / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg, "traffic": None,
                        "note": "SURVEY 8(d)'s per-step bytes (24 N_b + 36 K + 28 K_valid + W H per row) summed over every row's own steps, N_b taken as linear in "
                                "the step, over the WALL time of the call (preparation, the rows' schedule and the read-back included) -- these trees stay in the "
                                "young-tree regime of the step kernels (DESIGN.md section 14), which is latency- and slot-bound like the headline's (section 6)"}}
    if with_cpu:
        from oracle import orc
        m = 32
        t0 = time.perf_counter()
        nodes_o = its_o = 0
        for c in cs[:m]:
            o = cases.configure(orc.Oracle(), c)
            cases.grow(o, c, K=1, algo=orc.ALGO_SEQ)
            nodes_o += o.num_nodes() - 1
            its_o += o.num_iterations()
        dto = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": m / dto, "unit": "queries/s", "cores": 1, "kind": "port",
                               "sample": "the first %d of the queries, the reference's own loop (K = 1, rrt.rs:102-174 with its kd-tree; C restatement): "
                                         "%.3f s, %d iterations, %.0f node-expansions/s" % (m, dto, its_o, nodes_o / dto)}
    return out


def prm_roadmap(device, with_cpu):
    """SURVEY 8f.3, outside the timed region: PRM::grow_graph (prm.rs:38-109) -- a 200 000-sample PRM* roadmap on the
    benchmark map (max_step 0.1, search_radius 2.0 as the RRT* workload)."""
    import cases
    import po_rrt_amd
    n = 200000
    e = po_rrt_amd.Engine(device)
    e.set_grid(cases.load_map("map_benchmark_like"), (-1.0, -1.0), (1.0, 1.0), cases.SHELF)
    ts = []
    for rep in range(4):
        e.set_sampler((-1.0, -1.0), (1.0, 1.0), rep)
        t0 = time.perf_counter()
        e.grow_prm((0.0, -0.8), 0.1, 2.0, n)
        ts.append((time.perf_counter() - t0, e.metrics()["device_s"], len(e.edges()[0]) if rep == 3 else 0))
    wall, dev, _ = sorted(ts[1:])[1]
    E = ts[3][2]
    t0 = time.perf_counter()
    path = e.prm_plan_path((0.0, -0.8), (0.9, 0.0))        # PRM::plan_path on that roadmap (edges already ordered and fetched)
    t_path = time.perf_counter() - t0
    out = {"what": "PRM::grow_graph, %d samples -> %d forward edges" % (n, E), "ms_wall": 1e3 * wall, "ms_device": 1e3 * dev,
           "nodes_per_s": n / wall, "edges_per_s": E / wall, "plan_path": {"ms_wall": 1e3 * t_path, "states": int(len(path))}}
    if with_cpu:
        from oracle import orc
        o = orc.Oracle()
        o.set_grid(cases.load_map("map_benchmark_like"), (-1.0, -1.0), (1.0, 1.0), cases.SHELF)
        o.set_sampler((-1.0, -1.0), (1.0, 1.0), 3)
        t0 = time.perf_counter()
        o.grow_prm((0.0, -0.8), 0.1, 2.0, n)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        po = o.prm_plan_path((0.0, -0.8), (0.9, 0.0))
        dt2 = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dt, "unit": "nodes/s", "cores": 1, "kind": "port",
                               "sample": "the full workload, %.2f s (plan_path: %.2f s, %d states); C restatement of prm.rs:33-123 with the "
                                         "reference's kd-tree" % (dt, dt2, len(po))}
    return out


def mm_prm(device, with_cpu):
    """SURVEY 8f.3, second half, outside the timed region: MapShelfDomainTampPRM::grow_mm_prm (map_shelves_tamp_prm.rs:328-393) on the
    12-shelf problem (main.rs:386-408 map and prior): one PRM* roadmap per mode (belief), grown on the GPU from the point lists the
    reference's loop assigns to the modes."""
    import cases
    import po_rrt_amd
    case = cases.cfg4(1000, 1000)
    prior, n_iter = [1.0 / 12] * 12, 100
    e = cases.configure(po_rrt_amd.Engine(device), case)
    ts = []
    for rep in range(3):
        e.set_discrete_seed(rep)
        t0 = time.perf_counter()
        g = e.grow_mm_prm(case.start, prior, 0.1, 2.0, n_iter)
        ts.append((time.perf_counter() - t0, e.mm_seconds()))
    wall, sec = sorted(ts, key=lambda r: r[0])[1]
    nodes = sum(len(m["xy"]) for m in g["modes"])
    edges = sum(len(m["edges"][0]) for m in g["modes"])
    out = {"what": "grow_mm_prm: 12 shelves, uniform prior (%d reachable beliefs), %d samples per belief -> %d modes, %d transitions, %d roadmap nodes, "
                   "%d forward edges" % (g["n_beliefs"], n_iter, len(g["modes"]), len(g["transitions"]), nodes, edges),
           "ms_wall": 1e3 * wall, "ms_host_mode_tree": 1e3 * sec["host_s"], "ms_roadmaps": 1e3 * sec["roadmap_s"], "ms_device": 1e3 * sec["device_s"],
           "nodes_per_s": nodes / wall, "nodes_per_s_inside_the_library": nodes / max(sec["host_s"] + sec["roadmap_s"], 1e-9),
           "note": "the mode tree is decided on the host (the reference's sequential loop without the graphs), the roadmaps of all modes are built "
                   "in one pass on the GPU (k_mm_connect / k_mm_order: four launches for all modes); ms_wall also includes fetching every mode's "
                   "nodes and edges into Python"}
    if with_cpu:
        from oracle import orc
        o = cases.configure(orc.Oracle(), case)
        o.set_discrete_seed(2)
        t0 = time.perf_counter()
        go = o.grow_mm_prm(case.start, prior, 0.1, 2.0, n_iter)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": sum(len(m["xy"]) for m in go["modes"]) / dt, "unit": "nodes/s", "cores": 1, "kind": "port",
                               "sample": "the full workload, %.2f s; C restatement of map_shelves_tamp_prm.rs:135-393 with one kd-tree PRM per mode "
                                         "(oracle/mmprm.c)" % dt}
    return out


def cpu_baseline(case, args):
    """Single-thread C restatement of RRT::grow_tree (kd-tree and all) on the same workload -- the Rust
    reference cannot be built here (no rustc/cargo; crates not vendored)."""
    import cases
    from oracle import orc
    orc.build()
    reps, times, nodes = 5, [], 0
    for r in range(reps):
        o = cases.configure(orc.Oracle(), cases.Case(case, seed=r))
        t0 = time.perf_counter()
        cases.grow(o, case, K=1, algo=orc.ALGO_SEQ)
        times.append(time.perf_counter() - t0)
        nodes = o.num_nodes() - 1
    times.sort()
    return {
        "value": nodes / times[len(times) // 2],
        "unit": "node-expansions/s",
        "cores": 1,
        "kind": "port",
        "sample": "the full workload (%d iterations, %d nodes), median of %d runs, seeds 0..%d; "
                  "C restatement of rrt.rs:102-174 with the reference's kd-tree, gcc -O2 -ffp-contract=off"
                  % (args.n_iter, nodes, reps, reps - 1),
        "host_cpus": os.cpu_count(),
        "seconds_median": times[len(times) // 2],
    }


if __name__ == "__main__":
    main()
