#!/usr/bin/env python3
"""Writes tests/golden/trees/*.npz: trees / graphs produced by the CPU oracle (oracle-generated, NOT reference-generated --
the Rust reference cannot be built here) for a few seeds and maps, so that a drift of the oracle itself, or of
the HIP path against a frozen answer, shows up without the two being compared with each other.
usage: python tools/make_golden_trees.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import cases  # noqa: E402
from oracle import orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "trees")


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def specs():
    for seed in (0, 1, 2):
        yield "rrt_seq_map0_seed%d" % seed, cases.cfg1(2000, seed=seed), 1, orc.ALGO_SEQ
        yield "rrt_seq_benchmark_seed%d" % seed, cases.cfg2(2000, seed=seed), 1, orc.ALGO_SEQ
    yield "rrt_batched1024_benchmark_seed0", cases.cfg2(6000, seed=0), 1024, orc.ALGO_BATCHED_KD
    yield "pto_seq_2_goals_seed0", cases.cfg3(1500, 1500, seed=0), 1, orc.ALGO_SEQ
    yield "pto_batched256_near_goals_seed0", cases.cfg3_near(1500), 64, orc.ALGO_BATCHED_KD
    # BASELINE.json configs[3] at the size bench.py measures it (main.rs:386-408: map5-like, 12 shelves, visibility 0.2, start
    # (0, -0.8), max_step 0.05, search_radius 5; 20 000 iterations): digests only, and a prior with 8 of the 12 worlds possible
    # (255 beliefs: the CPU oracle builds that belief graph in ~20 s; the uniform prior's 4095 beliefs are out of its reach)
    yield "pto_cfg4_20000_it_255_beliefs", cases.cfg4(20000, 20000), 256, orc.ALGO_BATCHED_KD


CFG4_PRIOR = [0.125] * 8 + [0.0] * 4


def build(name, case, K, algo):
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=K, algo=algo)
    xy, parent, dist = o.tree()
    if name.startswith("pto_cfg4"):
        o.build_belief_graph(CFG4_PRIOR)
        beliefs, types, (coff, cid), (poff, pid) = o.belief_graph()
        d = o.expected_costs()
        oid, par, leaf = o.extract_policy(d)
        return dict(n_nodes=np.int64(len(xy)), node_digest=np.array(digest(xy, o.reach(), o.node_validity(), o.final_ids().astype(np.uint64))),
                    n_belief_edges=np.int64(len(cid)), belief_digest=np.array(digest(beliefs, types, coff, cid, poff, pid)),
                    cost_digest=np.array(digest(d)), root_cost_bits=np.array([d[0]]).view(np.uint64), policy_ids=oid, policy_parents=par,
                    policy_leaf=leaf.astype(np.uint8), K=np.int64(K), algo=np.int64(algo))
    rec = dict(xy_bits=xy.view(np.uint64), parent=parent.astype(np.int32), dist_bits=dist.view(np.uint64),
               final_ids=o.final_ids().astype(np.uint64), K=np.int64(K), algo=np.int64(algo))
    if case.mode == cases.PTO:
        f, t, v = o.edges()
        rec.update(reach=o.reach(), edge_digest=np.array(digest(f, t, v)))
        if name.startswith("pto_batched"):
            o.build_belief_graph([0.5, 0.5])
            beliefs, types, (coff, cid), (poff, pid) = o.belief_graph()
            d = o.expected_costs()
            oid, par, leaf = o.extract_policy(d)
            rec.update(belief_digest=np.array(digest(beliefs, types, coff, cid, poff, pid)), cost_digest=np.array(digest(d)),
                       root_cost_bits=np.array([d[0]]).view(np.uint64), policy_ids=oid, policy_parents=par)
    return rec


# bench.py's timed call (BASELINE.json configs[1] as the headline runs it): 256 queries of cfg2(111500), seeds = slot, K = 1024,
# advanced together as two sub-batches of 128.  Frozen here: the first and last member of each sub-batch.
BENCH = dict(Q=256, n_iter=111500, K=1024, members=(0, 127, 128, 255))


def build_bench_member(j):
    case = cases.cfg2(BENCH["n_iter"], seed=j)
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=BENCH["K"], algo=orc.ALGO_BATCHED_KD)
    xy, parent, dist = o.tree()
    sol = o.best_solution()
    cost = sol[1] if sol is not None else float("inf")
    return {"n_nodes_%d" % j: np.int64(len(xy)),
            "digest_%d" % j: np.array(digest(xy.view(np.uint64), parent.astype(np.int32), dist.view(np.uint64), o.final_ids().astype(np.uint64))),
            "cost_bits_%d" % j: np.array([cost]).view(np.uint64)}


def build_bench():
    rec = dict(Q=np.int64(BENCH["Q"]), n_iter=np.int64(BENCH["n_iter"]), K=np.int64(BENCH["K"]), members=np.array(BENCH["members"], dtype=np.int64))
    for j in BENCH["members"]:
        rec.update(build_bench_member(j))
    return rec


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    if not only or "bench_members" in only:
        np.savez_compressed(os.path.join(OUT, "bench_members.npz"), **build_bench())
        print("bench_members")
    for name, case, K, algo in specs():
        if only and name not in only:
            continue
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **build(name, case, K, algo))
        print(name)
