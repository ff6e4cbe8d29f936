"""Pins the oracle's dynamic programming (oracle/dp.c) with the reference's two known-answer tests
(src/belief_graph.rs:502-567, graphs in tests/kat_graphs.py), then checks that the result is the fixpoint of the
relaxation whatever the order (a plain sweep-until-stable evaluation in Python gives the same bits).  CPU only."""
import numpy as np

import cases
import kat_graphs
from oracle import orc


def solve(g):
    dist, ccsr, _ = orc.conditional_dijkstra(g["xy"], g["belief_vec"], g["beliefs"], g["types"], g["children"], g["parents"], g["finals"])
    oid, par, leaf = orc.extract_policy(g["xy"], g["belief_id"], g["belief_vec"], g["beliefs"], ccsr, dist)
    return dist, oid, par, leaf


def path_to_leaf(g, oid, par, k):
    """Policy::path_to_leaf (common.rs:70-83)"""
    path = []
    while k >= 0:
        path.append(g["xy"][int(oid[k])])
        k = int(par[k])
    return path[::-1]


def test_graph_1():
    """belief_graph.rs:502-543"""
    g = kat_graphs.graph_1()
    d, oid, par, leaf = solve(g)
    assert d[0] < d[1] and d[0] < d[2] and d[4] < d[0]
    assert d[6] < d[5] and d[6] < d[8] and d[7] < d[6] and d[9] < d[7] and d[10] < d[9]
    assert d[12] < d[11] and d[12] < d[13] and d[14] < d[12] and d[15] < d[14] and d[16] < d[15]
    assert d[4] == 0.4 * d[5] + 0.6 * d[11]                        # belief transition
    leafs = np.flatnonzero(leaf)
    assert len(leafs) == 2
    assert g["xy"][int(oid[leafs[0]])] == [0.0, 4.0] and g["xy"][int(oid[leafs[1]])] == [0.0, 4.0]
    assert g["beliefs"][g["belief_vec"][int(oid[leafs[0]])]] == [0.0, 1.0]        # second belief first
    assert g["beliefs"][g["belief_vec"][int(oid[leafs[1]])]] == [1.0, 0.0]
    assert path_to_leaf(g, oid, par, leafs[0]) == [[0.0, 1.0], [0.0, 0.0], [0.0, 0.0], [0.0, 1.0], [1.0, 2.0], [10.0, 3.0], [0.0, 4.0]]
    assert path_to_leaf(g, oid, par, leafs[1]) == [[0.0, 1.0], [0.0, 0.0], [0.0, 0.0], [0.0, 1.0], [-1.0, 2.0], [-1.0, 3.0], [0.0, 4.0]]


def test_graph_2():
    """belief_graph.rs:545-567"""
    g = kat_graphs.graph_2()
    d, oid, par, leaf = solve(g)
    assert int(np.argmax(d)) == 10 and d.max() == 8.0
    leafs = np.flatnonzero(leaf)
    assert len(leafs) == 2
    assert g["xy"][int(oid[leafs[0]])] == [0.0, 3.0] and g["xy"][int(oid[leafs[1]])] == [0.0, 3.0]


def sweep_fixpoint(xy, bvec, beliefs, types, coff, cid, finals):
    """the same relaxations evaluated in sweeps over all nodes until nothing changes (the schedule of the GPU kernels)"""
    n = len(types)
    d = np.full(n, np.inf)
    d[np.asarray(finals, dtype=np.int64)] = 0.0
    xy = np.asarray(xy, dtype=np.float64)
    beliefs = np.asarray(beliefs, dtype=np.float64)
    changed = True
    while changed:
        changed = False
        for u in range(n):
            ch = cid[int(coff[u]):int(coff[u + 1])]
            if types[u] == 1:
                alt = np.inf
                for v in ch:
                    dx, dy = xy[v][0] - xy[u][0], xy[v][1] - xy[u][1]
                    alt = min(alt, 0.0 + (np.sqrt(0.0 + dx * dx + dy * dy) + d[v]))
            elif types[u] == 2:
                alt = 0.0
                for v in ch:
                    p = 0.0
                    for w in range(beliefs.shape[1]):
                        p = p + (beliefs[bvec[u]][w] if beliefs[bvec[v]][w] > 0.0 else 0.0)
                    dx, dy = xy[v][0] - xy[u][0], xy[v][1] - xy[u][1]
                    alt = alt + p * (np.sqrt(0.0 + dx * dx + dy * dy) + d[v])
            else:
                continue
            if alt < d[u]:
                d[u] = alt
                changed = True
    return d


def test_any_order_reaches_the_same_fixpoint():
    for g in (kat_graphs.graph_1(), kat_graphs.graph_2()):
        d, (coff, cid), _ = orc.conditional_dijkstra(g["xy"], g["belief_vec"], g["beliefs"], g["types"], g["children"], g["parents"], g["finals"])
        d2 = sweep_fixpoint(g["xy"], g["belief_vec"], g["beliefs"], g["types"], coff, cid, g["finals"])
        assert np.array_equal(d.view(np.uint64), d2.view(np.uint64))
    case = cases.cfg3_near(1500)                                    # a grown graph: 3 beliefs, ~1000 graph nodes
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=64, algo=orc.ALGO_BATCHED_KD)
    o.build_belief_graph([0.5, 0.5])
    d = o.expected_costs()
    beliefs, types, (coff, cid), _ = o.belief_graph()
    xy = np.repeat(o.tree()[0], len(beliefs), axis=0)
    bvec = np.arange(len(types)) % len(beliefs)
    finals = np.flatnonzero(d == 0.0)
    d2 = sweep_fixpoint(xy, bvec, beliefs, types, coff, cid, finals)
    assert np.array_equal(d.view(np.uint64), d2.view(np.uint64))
    assert np.isfinite(d[0]) and o.is_final_set_complete()
    oid, par, leaf = o.extract_policy(d)
    assert oid[0] == 0 and par[0] == -1 and leaf.sum() == 2          # one leaf per world
