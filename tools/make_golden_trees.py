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


def build(name, case, K, algo):
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=K, algo=algo)
    xy, parent, dist = o.tree()
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


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name, case, K, algo in specs():
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **build(name, case, K, algo))
        print(name)
