"""Pins the oracle's belief-space functions (oracle/belief.c) with the reference's own unit tests, restated on the
synthetic maps (the reference rasters are LFS pointers): observation model and reachable beliefs of the door domain
(src/map_io.rs:666-723) and of the shelf domain (src/map_shelves_io.rs:595-660); then checks the C restatement of
PTO::build_belief_graph (src/pto.rs:185-259) against a second, independent pure-Python restatement on small graphs.
CPU only."""
import numpy as np
import pytest

import cases
from oracle import orc


def door_map(visibility=0.1):
    o = orc.Oracle()
    o.set_grid(cases.load_map("door_map_like"), (-1.0, -1.0), (1.0, 1.0), cases.DOOR)
    o.set_zones(cases.load_map("door_map_like_zone_ids"), visibility)
    return o


def shelf_map(visibility=0.1):
    o = orc.Oracle()
    o.set_grid(cases.load_map("map1_2_goals_like"), (-1.0, -1.0), (1.0, 1.0), cases.SHELF)
    o.set_zones(cases.load_map("map1_2_goals_like_zone_ids"), visibility)
    return o


def rows(a):
    return [list(r) for r in np.asarray(a)]


# ---------------------------------------------------------------- door domain, map_io.rs:666-723

def test_door_observation_model_in_zones():
    """map_io.rs:666-686 test_map_2_observation_model_in_zones (2 doors, 4 worlds)."""
    m = door_map(0.1)
    z0 = m.zone_positions()[0]
    at = (z0[0], z0[1] - 0.05)                                  # just below door 0, inside its visibility disc
    post = m.observe(at, [0.25] * 4)
    assert rows(post) == [[0.5, 0.0, 0.5, 0.0], [0.0, 0.5, 0.0, 0.5]]        # zone 0 closed, zone 0 open
    assert rows(m.observe(at, [1.0, 0.0, 0.0, 0.0])) == [[1.0, 0.0, 0.0, 0.0]]
    assert rows(m.observe(at, [0.0, 1.0, 0.0, 0.0])) == [[0.0, 1.0, 0.0, 0.0]]
    assert rows(m.observe(at, [0.5, 0.5, 0.0, 0.0])) == [[1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]]


def test_door_observation_model_outside_zones():
    """map_io.rs:688-695"""
    m = door_map(0.1)
    assert rows(m.observe((-0.3, -0.5), [0.25] * 4)) == [[0.25] * 4]


def test_door_two_zones_seen_at_the_same_time():
    """map_io.rs:697-704: the fold over zones in ascending order"""
    m = door_map(2.0)
    zp = m.zone_positions()
    at = (0.5 * (zp[0][0] + zp[1][0]), -0.5)                     # below the wall, both doors in line of sight
    assert m.zone_observable(at, 0) == 1 and m.zone_observable(at, 1) == 1
    assert rows(m.observe(at, [0.25] * 4)) == [[1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]]


def test_door_reachable_beliefs():
    """map_io.rs:706-722: 9 beliefs from the uniform one"""
    m = door_map(0.1)
    r = rows(m.reachable_beliefs([0.25] * 4))
    assert len(r) == 9
    for b in ([0.25] * 4, [0.5, 0.5, 0.0, 0.0], [0.5, 0.0, 0.5, 0.0], [0.0, 0.5, 0.0, 0.5], [0.0, 0.0, 0.5, 0.5],
              [1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]):
        assert b in r
    assert r[0] == [0.25] * 4                                   # the start belief is id 0 (map_io.rs:520)


def test_door_world_validities():
    """map_io.rs:724-735"""
    m = door_map(0.1)
    v = [int(x) for x in m.validities()]
    assert len(v) == 3 and 0b1111 in v and 0b1010 in v and 0b1100 in v


# ---------------------------------------------------------------- shelf domain, map_shelves_io.rs:595-660

def test_shelf_observation_model():
    """map_shelves_io.rs:595-632"""
    m = shelf_map(0.1)
    z0 = m.zone_positions()[0]
    assert rows(m.observe(z0, [0.5, 0.5])) == [[1.0, 0.0], [0.0, 1.0]]          # in zone 0 / not in zone 0
    for b in ([0.5, 0.5], [1.0, 0.0], [0.0, 1.0]):
        assert rows(m.observe((0.0, 0.0), b)) == [b]                              # outside zones
    assert rows(m.observe(z0, [1.0, 0.0])) == [[1.0, 0.0]]                        # already known
    assert rows(m.observe(z0, [0.0, 1.0])) == [[0.0, 1.0]]


def test_shelf_zone_seen_over_a_distance():
    """map_shelves_io.rs:634-641: a zone behind a high obstacle is not observed, the other one splits the belief"""
    m = shelf_map(1.0)
    seen = [m.zone_observable((0.5, 0.5), z) for z in (0, 1)]
    post = rows(m.observe((0.5, 0.5), [0.5, 0.5]))
    if seen == [0, 1]:
        assert post == [[0.0, 1.0], [1.0, 0.0]]                                   # the reference's expectation
    elif seen == [1, 1]:
        assert post == [[1.0, 0.0], [0.0, 1.0]]
    else:
        assert seen[1] == 1


def test_shelf_reachable_beliefs():
    """map_shelves_io.rs:643-652"""
    m = shelf_map(0.1)
    r = rows(m.reachable_beliefs([0.5, 0.5]))
    assert len(r) == 3 and [0.5, 0.5] in r and [1.0, 0.0] in r and [0.0, 1.0] in r


def test_hash_definition():
    """common.rs:352-355: sum (10^i + 1) * round(1000 p_i)"""
    o = orc.Oracle()
    assert o.belief_hash([0.25, 0.25, 0.25, 0.25]) == 250 * (2 + 11 + 101 + 1001)
    assert o.belief_hash([1.0, 0.0]) == 2000 and o.belief_hash([0.0, 1.0]) == 11000
    assert o.belief_hash([1.0 / 3, 2.0 / 3]) == 2 * 333 + 11 * 667


# ---------------------------------------------------------------- build_belief_graph vs a second restatement

def py_build_belief_graph(o, start):
    """pto.rs:185-259 in plain Python lists (small graphs only); returns types, children, parents."""
    R = [tuple(b) for b in o.reachable_beliefs(start)]
    hashes = [o.belief_hash(b) for b in R]
    V = [int(v) for v in o.validities()]
    nw = o.n_worlds()
    compat = [[all(not (p > 0.0) or (v >> w) & 1 for w, p in enumerate(b)) for v in V] for b in R]
    xy, _, _ = o.tree()
    nv = o.node_validity()
    f, t, ev = o.edges()
    N, B = len(xy), len(R)
    adj = [[] for _ in range(N)]                        # PTOGraph.children in push order (pto.rs:111-120)
    e = 0
    while e < len(t):
        e1 = e
        while e1 < len(t) and t[e1] == t[e]:
            e1 += 1
        for k in range(e, e1):
            adj[f[k]].append((int(t[k]), int(ev[k])))
        for k in range(e, e1):
            adj[t[k]].append((int(f[k]), int(ev[k])))
        e = e1
    n2b = [[(i * B + b) if compat[b][nv[i]] else None for b in range(B)] for i in range(N)]
    types = [0] * (N * B)
    children = [[] for _ in range(N * B)]
    parents = [[] for _ in range(N * B)]

    def add_edge(a, c):
        children[a].append(c)
        parents[c].append(a)

    for i in range(N):
        for b in range(B):
            for child in o.observe(xy[i], R[b]):
                h = o.belief_hash(child)
                if h != hashes[b]:
                    cb = hashes.index(h)
                    if n2b[i][b] is not None and n2b[i][cb] is not None:
                        types[n2b[i][b]] = 2
                        add_edge(n2b[i][b], n2b[i][cb])
    for i in range(N):
        for b in range(B):
            p = n2b[i][b]
            if p is None or types[p] == 2:
                continue
            for cid, cv in adj[i]:
                c = n2b[cid][b]
                if c is not None and compat[b][cv]:
                    types[p] = 1
                    add_edge(p, c)
    return types, children, parents


def csr_lists(off, ids):
    return [list(ids[int(off[i]):int(off[i + 1])]) for i in range(len(off) - 1)]


SMALL = {
    "shelf_seq": (lambda: cases.cfg3(1500, 1500), 1, [0.5, 0.5]),
    "shelf_batched": (lambda: cases.cfg3(2500, 2500), 64, [0.5, 0.5]),
    "door_seq": (lambda: cases.cfg_door(300, 300), 1, [0.25] * 4),
    "door_batched_skewed_prior": (lambda: cases.cfg_door(400, 400), 64, [0.1, 0.2, 0.3, 0.4]),
}


@pytest.mark.parametrize("name", sorted(SMALL))
def test_build_belief_graph_c_equals_python(name):
    mk, K, start = SMALL[name]
    case = mk()
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=K, algo=orc.ALGO_BATCHED_KD if K > 1 else orc.ALGO_SEQ)
    o.build_belief_graph(start)
    beliefs, types, (coff, cid), (poff, pid) = o.belief_graph()
    t2, c2, p2 = py_build_belief_graph(o, start)
    assert list(types) == t2
    assert csr_lists(coff, cid) == c2
    assert csr_lists(poff, pid) == p2
    assert len(cid) == len(pid) == sum(len(c) for c in c2) > 0
    assert 2 in types and 1 in types                            # both observation and action nodes occur


def test_belief_graph_structure_properties():
    """What plan_belief_space relies on: observation nodes only have observation edges (same graph node, other
    belief), action nodes only geometric ones (same belief, other graph node), incompatible pairs stay isolated."""
    case = cases.cfg_door(600, 600)
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=64, algo=orc.ALGO_BATCHED_KD)
    o.build_belief_graph([0.25] * 4)
    beliefs, types, (coff, cid), (poff, pid) = o.belief_graph()
    B = len(beliefs)
    for i in np.flatnonzero(types == 2)[:2000]:
        ch = cid[int(coff[i]):int(coff[i + 1])]
        assert len(ch) and np.all(ch // B == i // B) and np.all(ch % B != i % B)
    for i in np.flatnonzero(types == 1)[:2000]:
        ch = cid[int(coff[i]):int(coff[i + 1])]
        assert len(ch) and np.all(ch % B == i % B) and np.all(ch // B != i // B)
    for i in np.flatnonzero(types == 0)[:2000]:
        assert coff[i] == coff[i + 1]
