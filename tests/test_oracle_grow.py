"""Oracle self-consistency: the batched contract reduces to the reference loop.

ref_batched(K=1) must equal ref_seq (the literal restatement with the kd-tree) bit for
bit, and the kd-accelerated batched variant must equal the brute-force definition.
CPU only.
"""
import numpy as np
import pytest

import cases
from oracle import orc


def run(case, K, algo):
    o = cases.configure(orc.Oracle(), case)
    rc = cases.grow(o, case, K=K, algo=algo)
    return o, rc


def assert_same_tree(a, b):
    xa, pa, da = a.tree()
    xb, pb, db = b.tree()
    assert a.num_iterations() == b.num_iterations()
    assert xa.shape == xb.shape
    assert np.array_equal(pa, pb)                       # parents: bit exact
    assert np.array_equal(xa.view(np.uint64), xb.view(np.uint64))   # coordinates: bit exact
    assert np.array_equal(da.view(np.uint64), db.view(np.uint64))
    assert np.array_equal(a.final_ids(), b.final_ids())
    assert np.array_equal(a.final_masks(), b.final_masks())


def edge_sets(o):
    f, t, v = o.edges()
    order = np.lexsort((f, t))
    return f[order], t[order], v[order]


def assert_same_graph(a, b, ordered):
    assert_same_tree(a, b)
    assert np.array_equal(a.reach(), b.reach())
    assert np.array_equal(a.node_validity(), b.node_validity())
    if ordered:
        for x, y in zip(a.edges(), b.edges()):
            assert np.array_equal(x, y)
    else:   # ref_seq lists neighbours in kd pre-order, the batched contract in ascending id
        for x, y in zip(edge_sets(a), edge_sets(b)):
            assert np.array_equal(x, y)
    assert a.is_final_set_complete() == b.is_final_set_complete()


RRT_CASES = [cases.empty_space(1000, 10000), cases.cfg1(3000), cases.cfg2(4000), cases.cfg2(3000, seed=3, grid="map_benchmark_like_c"),
             cases.cfg2_obs(1500)]
PTO_CASES = [cases.cfg3(1500, 20000), cases.cfg4(1500, 4000), cases.cfg_door(1200, 20000), cases.cfg_door(1500, 6000, paper=True)]


@pytest.mark.parametrize("case", RRT_CASES, ids=lambda c: c.name)
def test_rrt_batched1_equals_seq(oracle_lib, case):
    a, _ = run(case, 1, orc.ALGO_SEQ)
    b, _ = run(case, 1, orc.ALGO_BATCHED)
    assert a.num_nodes() > 100
    assert_same_tree(a, b)


@pytest.mark.parametrize("case", PTO_CASES, ids=lambda c: c.name)
def test_pto_batched1_equals_seq(oracle_lib, case):
    a, rca = run(case, 1, orc.ALGO_SEQ)
    b, rcb = run(case, 1, orc.ALGO_BATCHED)
    assert rca == rcb
    assert a.num_nodes() > 100
    assert_same_graph(a, b, ordered=False)


@pytest.mark.parametrize("K", [1, 7, 64, 1024])
@pytest.mark.parametrize("case", [cases.cfg1(2500), cases.cfg2(3000), cases.cfg2_obs(1200)], ids=lambda c: c.name)
def test_rrt_batched_kd_equals_brute(oracle_lib, case, K):
    a, _ = run(case, K, orc.ALGO_BATCHED)
    b, _ = run(case, K, orc.ALGO_BATCHED_KD)
    assert_same_tree(a, b)


@pytest.mark.parametrize("K", [1, 16, 256])
@pytest.mark.parametrize("case", [cases.cfg3(1200, 20000), cases.cfg4(1200, 3000), cases.cfg_door(1000, 20000)], ids=lambda c: c.name)
def test_pto_batched_kd_equals_brute(oracle_lib, case, K):
    a, rca = run(case, K, orc.ALGO_BATCHED)
    b, rcb = run(case, K, orc.ALGO_BATCHED_KD)
    assert rca == rcb
    assert_same_graph(a, b, ordered=True)


def test_plan_empty_space(oracle_lib):
    # rrt.rs:254-267: a path with more than two states is found
    o, _ = run(cases.empty_space(1000, 10000), 1, orc.ALGO_SEQ)
    path, cost = o.best_solution()
    assert len(path) > 2 and cost > 0.0
    assert list(path[0]) == [0.0, 0.0]
    assert o.goal(path[-1]) is not None
    # the cost of the path is the sum of its edge lengths (rrt.rs:223-227)
    assert cost == sum(float(np.sqrt(((path[i + 1] - path[i]) ** 2).sum())) for i in range(len(path) - 1)) or \
        abs(cost - np.sqrt(((path[1:] - path[:-1]) ** 2).sum(axis=1)).sum()) < 1e-12


def test_goal_bias_and_rng_stream(oracle_lib):
    # rrt.rs:176-181: every 100th iteration uses goal_example(0) and draws nothing.
    case = cases.cfg1(1000)
    a, _ = run(case, 1, orc.ALGO_SEQ)
    # replay the same stream as injected samples: 2 draws per non-goal iteration
    s = cases.configure(orc.Oracle(), case)
    xy = np.array([s.sample() for _ in range(1000 - 10)])
    b = cases.configure(orc.Oracle(), case)
    b.set_samples(xy)
    cases.grow(b, case, K=1, algo=orc.ALGO_SEQ)
    assert_same_tree(a, b)


def test_sampler_state_persists_across_grows(oracle_lib):
    # map_shelves_tamp_rrt.rs:196-232: one RRT object (one RNG stream) serves many plans
    case = cases.cfg1(500)
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=1, algo=orc.ALGO_SEQ)
    t1 = o.tree()[0].copy()
    cases.grow(o, case, K=1, algo=orc.ALGO_SEQ)
    t2 = o.tree()[0].copy()
    assert t1.shape != t2.shape or not np.array_equal(t1, t2)


def test_rewire_keeps_tree_consistent(oracle_lib):
    # rrt.rs:30-46: every non-root node has a parent with a smaller id or one that
    # rewired it later; parents always exist; root has none.
    o, _ = run(cases.cfg2(4000), 1, orc.ALGO_SEQ)
    xy, parent, dist = o.tree()
    assert parent[0] == -1 and (parent[1:] >= 0).all() and (parent[1:] < len(parent)).all()
    assert (parent[1:] != np.arange(1, len(parent))).all()
    assert (parent[1:] > np.arange(1, len(parent))).any()     # some node was rewired to a later node
    # no cycles
    depth = np.zeros(len(parent), dtype=np.int64)
    for j in range(1, len(parent)):
        p, steps = j, 0
        while p > 0:
            p = parent[p]
            steps += 1
            assert steps <= len(parent)


def test_pto_start_must_be_valid(oracle_lib):
    # pto.rs:61 expect("Start from a valid state!")
    case = cases.cfg3(100, 100)
    case.update(start=(-0.22, 0.0))     # inside a wall of map1_2_goals_like
    o = cases.configure(orc.Oracle(), case)
    with pytest.raises(RuntimeError):
        cases.grow(o, case, K=1, algo=orc.ALGO_SEQ)


def test_pto_completes(oracle_lib):
    # pto.rs:466-479: with enough iterations every world has a reachable final node
    o, rc = run(cases.cfg3(2000, 100000), 1, orc.ALGO_SEQ)
    assert rc == 0 and o.is_final_set_complete()
    fin = o.final_masks()
    assert (np.bitwise_or.reduce(fin) & 3) == 3
