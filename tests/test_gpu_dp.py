"""GPU parity of the expected costs over the belief graph (porrt_bg_compute_expected_costs / porrt_conditional_dijkstra,
through the C ABI): the reference's two known-answer graphs (src/belief_graph.rs:502-567) run on the device, and the
costs of grown graphs equal the oracle's queue-driven loop bit for bit."""
import numpy as np
import pytest

import cases
import kat_graphs
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    from po_rrt_amd import build
    build.build()
    import po_rrt_amd
    return po_rrt_amd


def gpu_kat(eng_mod, g):
    return eng_mod.conditional_dijkstra(g["xy"], g["belief_vec"], g["beliefs"], g["types"], g["children"], g["parents"], g["finals"])


def test_reference_graph_1_on_the_device(eng_mod):
    """belief_graph.rs:502-543 (the distance assertions; the policy part is host code)"""
    g = kat_graphs.graph_1()
    d = gpu_kat(eng_mod, g)
    assert d[0] < d[1] and d[0] < d[2] and d[4] < d[0]
    assert d[6] < d[5] and d[6] < d[8] and d[7] < d[6] and d[9] < d[7] and d[10] < d[9]
    assert d[12] < d[11] and d[12] < d[13] and d[14] < d[12] and d[15] < d[14] and d[16] < d[15]
    assert d[4] == 0.4 * d[5] + 0.6 * d[11]
    do, _, _ = orc.conditional_dijkstra(g["xy"], g["belief_vec"], g["beliefs"], g["types"], g["children"], g["parents"], g["finals"])
    assert np.array_equal(d.view(np.uint64), do.view(np.uint64))


def test_reference_graph_2_on_the_device(eng_mod):
    """belief_graph.rs:545-567"""
    g = kat_graphs.graph_2()
    d = gpu_kat(eng_mod, g)
    assert int(np.argmax(d)) == 10 and d.max() == 8.0
    do, _, _ = orc.conditional_dijkstra(g["xy"], g["belief_vec"], g["beliefs"], g["types"], g["children"], g["parents"], g["finals"])
    assert np.array_equal(d.view(np.uint64), do.view(np.uint64))


def test_explicit_graph_errors(eng_mod):
    g = kat_graphs.graph_1()
    types = list(g["types"])
    types[6] = 0                                             # a parent whose type is Unknown: the reference panics
    with pytest.raises(RuntimeError):
        eng_mod.conditional_dijkstra(g["xy"], g["belief_vec"], g["beliefs"], types, g["children"], g["parents"], g["finals"])
    d = eng_mod.conditional_dijkstra(g["xy"], g["belief_vec"], g["beliefs"], g["types"], g["children"], g["parents"], [])
    assert np.all(np.isinf(d))                               # no final node: nothing is reachable


def both(eng_mod, case, K, prior):
    e = cases.configure(eng_mod.Engine(), case)
    cases.grow(e, case, K=K)
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=K, algo=orc.ALGO_BATCHED_KD)
    e.build_belief_graph(prior)
    o.build_belief_graph(prior)
    e.compute_expected_costs()
    return e, o, e.expected_costs(), o.expected_costs()


def door_goal_behind_door_1(n, seed=0):
    c = cases.cfg_door(n, n, seed=seed)
    c.update(goals=[(0.5, 0.3)])                 # just behind the right-hand door: reached within a few thousand iterations
    return c


GROWN = {
    "shelf_2_worlds_near_goals": (lambda: cases.cfg3_near(1500), 64, [0.5, 0.5]),
    "shelf_2_worlds_skewed_prior": (lambda: cases.cfg3_near(2500, seed=2), 64, [0.3, 0.7]),
    "shelf_2_worlds_until_complete": (lambda: cases.cfg3(), 256, [0.5, 0.5]),                 # pto.rs:466-490: ~14k nodes
    "door_4_worlds_root_unreachable": (lambda: door_goal_behind_door_1(5000), 256, [0.1, 0.2, 0.3, 0.4]),
    "door_4_worlds_door_known_open": (lambda: door_goal_behind_door_1(5000), 256, [0.0, 0.0, 0.4, 0.6]),
    "door_paper_map_16_worlds_until_complete": (lambda: cases.cfg_door(paper=True), 256, [1.0 / 16] * 16),
}


@pytest.mark.parametrize("name", sorted(GROWN))
def test_expected_costs_equal_oracle(eng_mod, name):
    mk, K, prior = GROWN[name]
    e, o, de, do = both(eng_mod, mk(), K, prior)
    assert len(de) == len(do)
    assert np.array_equal(de.view(np.uint64), do.view(np.uint64)), "expected costs differ"
    assert e.expected_cost_of(0) == do[0]
    info = e.dp_info()
    assert info["sweeps"] > 0
    if np.isfinite(do[0]):                                   # PTO::extract_policy: the same policy tree, node for node
        (oid, par, leaf), cost = e.extract_policy()
        oo, po, lo = o.extract_policy(do)
        assert cost == do[0]
        assert np.array_equal(oid, oo) and np.array_equal(par, po) and np.array_equal(leaf, lo)
        assert leaf.sum() >= 1 and par[0] == -1 and oid[0] == 0
    else:
        with pytest.raises(RuntimeError):
            e.extract_policy()
    if name == "shelf_2_worlds_near_goals":
        assert np.isfinite(de[0]) and de[0] > 0.0            # a policy exists from the root
        assert (de == 0.0).sum() >= 2


def test_expected_costs_twelve_worlds(eng_mod):
    """4095 beliefs: the goals of the 12-shelf problem, sensor range long enough for a small graph"""
    case = cases.cfg4(700, 700)
    case.update(visibility=0.6, start=(0.0, -0.3))
    e, o, de, do = both(eng_mod, case, 64, [1.0 / 12] * 12)
    assert np.array_equal(de.view(np.uint64), do.view(np.uint64))
    assert np.isfinite(de).any()


def test_recompute_after_rebuild(eng_mod):
    case = cases.cfg3_near(1500)
    e = cases.configure(eng_mod.Engine(), case)
    cases.grow(e, case, K=64)
    with pytest.raises(RuntimeError):
        e.compute_expected_costs()                           # no belief graph yet
    e.build_belief_graph([0.5, 0.5])
    e.compute_expected_costs()
    d1 = e.expected_costs()
    e.build_belief_graph([0.9, 0.1])
    e.compute_expected_costs()
    d2 = e.expected_costs()
    assert np.isfinite(d1[0]) and np.isfinite(d2[0]) and d1[0] != d2[0]
    e.compute_expected_costs()
    assert np.array_equal(e.expected_costs().view(np.uint64), d2.view(np.uint64))


def test_batch_regrow_invalidates_the_belief_graph(eng_mod):
    """a context regrown through porrt_grow_batch loses its belief graph and costs, as after porrt_grow: the old CSR was
    sized for the old node count (round-1 advisor finding: out-of-bounds writes in the cost kernels)"""
    case = cases.cfg3_near(1500)
    es = [cases.configure(eng_mod.Engine(), cases.Case(case, seed=q)) for q in range(2)]
    for e in es:
        cases.grow(e, case, K=64)
        e.build_belief_graph([0.5, 0.5])
        e.compute_expected_costs()
    eng_mod.Engine.grow_batch(es, [case.start] * 2, case.max_step, case.search_radius, 2500, 64, mode=cases.PTO)
    for e in es:
        with pytest.raises(RuntimeError):
            e.compute_expected_costs()
        with pytest.raises(RuntimeError):
            e.extract_policy()
        assert e.bg_num_edges() == 0
        e.build_belief_graph([0.5, 0.5])                     # and a fresh build on the new graph works
        e.compute_expected_costs()
        assert np.isfinite(e.expected_costs()[0])


def test_layered_and_swept_evaluations_agree(eng_mod):
    """the context path solves the layers one after the other; option dp_sweeps = the general whole-graph sweeps"""
    case = cases.cfg_door(paper=True)
    e = cases.configure(eng_mod.Engine(), case)
    cases.grow(e, case, K=256)
    e.build_belief_graph([1.0 / 16] * 16)
    e.compute_expected_costs()
    d1, levels = e.expected_costs(), e.dp_info()["sweeps"]
    e.set_option("dp_sweeps", 1)
    e.compute_expected_costs()
    d2, sweeps = e.expected_costs(), e.dp_info()["sweeps"]
    assert np.array_equal(d1.view(np.uint64), d2.view(np.uint64))
    assert levels > 0 and sweeps > 0


def test_large_graph_few_beliefs(eng_mod):
    """many graph nodes, three beliefs: levels one or two beliefs wide"""
    case = cases.cfg3(30000, 30000)
    e, o, de, do = both(eng_mod, case, 256, [0.5, 0.5])
    assert e.num_nodes() > 21000
    assert np.array_equal(de.view(np.uint64), do.view(np.uint64))
    assert np.isfinite(de[0])


def test_policy_walk_that_does_not_terminate_is_an_error(eng_mod):
    """extract_policy (belief_graph.rs:184-217) has no memory: on the reference's own recorded problem (main.rs:893-908, paper_map_4.pgm) grown by
    its own loop (K = 1) with seed 0, two graph nodes at one place are each the other's first best child and the walk never ends -- the C
    restatement runs into its cap, the engine says so at once; seed 1 gives a policy, node for node the oracle's"""
    import time
    prior = [1.0 / 16] * 16
    for seed, ends in ((0, False), (1, True)):
        case = cases.cfg_map4(5000, seed)
        e = cases.configure(eng_mod.Engine(), case)
        cases.grow(e, case, K=1)
        e.build_belief_graph(prior)
        e.compute_expected_costs()
        o = cases.configure(orc.Oracle(), case)
        cases.grow(o, case, K=1, algo=orc.ALGO_SEQ)
        o.build_belief_graph(prior)
        d = o.expected_costs()
        assert d[0] == e.expected_cost_of(0)
        if ends:
            (oid, par, leaf), _ = e.extract_policy()
            ooid, opar, oleaf = o.extract_policy(d)
            assert np.array_equal(oid, ooid) and np.array_equal(par, opar) and np.array_equal(leaf, oleaf)
        else:
            t0 = time.perf_counter()
            with pytest.raises(eng_mod.PorrtError, match="returns to a belief node on its own path"):
                e.extract_policy()
            assert time.perf_counter() - t0 < 1.0
            with pytest.raises(RuntimeError):
                o.extract_policy(d, cap=1 << 14)
