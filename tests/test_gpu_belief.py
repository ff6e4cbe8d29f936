"""GPU parity of the belief-space expansion (porrt_build_belief_graph, through the C ABI) against the oracle's literal
restatement of PTO::build_belief_graph (src/pto.rs:185-259): reachable beliefs, node types, children and parents lists
bit for bit and in the reference's push order; plus size-independent properties at a larger size."""
import numpy as np
import pytest

import cases
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    from po_rrt_amd import build
    build.build()
    import po_rrt_amd
    return po_rrt_amd


def grown_pair(eng_mod, case, K):
    e = cases.configure(eng_mod.Engine(), case)
    cases.grow(e, case, K=K)
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=K, algo=orc.ALGO_BATCHED_KD)
    assert np.array_equal(e.tree()[1], o.tree()[1])
    return e, o


def assert_same_belief_graph(e, o):
    be, te, (ce_off, ce), (pe_off, pe) = e.belief_graph()
    bo, to, (co_off, co), (po_off, po) = o.belief_graph()
    assert np.array_equal(be.view(np.uint64), bo.view(np.uint64)), "reachable belief states differ"
    assert np.array_equal(te, to), "node types differ"
    assert np.array_equal(ce_off, co_off) and np.array_equal(ce, co), "children lists differ"
    assert np.array_equal(pe_off, po_off) and np.array_equal(pe, po), "parents lists differ"


def far_sighted(case, visibility):
    """the same problem started between the shelves and with a longer sensor range, so that a small graph already sees
    several zones at once"""
    case.update(visibility=visibility, start=(0.0, -0.3))
    return case


CASES = {
    "shelf_2_worlds": (lambda: cases.cfg3(4000, 4000), 256, [0.5, 0.5]),
    "shelf_2_worlds_skewed": (lambda: cases.cfg3(2500, 2500, seed=3), 64, [0.2, 0.8]),
    "shelf_12_worlds_far_sighted": (lambda: far_sighted(cases.cfg4(300, 300), 0.6), 64, [1.0 / 12] * 12),
    "shelf_8_of_12_worlds_possible": (lambda: far_sighted(cases.cfg4(1000, 1000), 0.4), 64, [0.125] * 8 + [0.0] * 4),
    "door_4_worlds": (lambda: cases.cfg_door(3000, 3000), 256, [0.25] * 4),
    "door_4_worlds_skewed": (lambda: cases.cfg_door(2000, 2000, seed=2), 64, [0.1, 0.2, 0.3, 0.4]),
    "door_known_world": (lambda: cases.cfg_door(1500, 1500), 64, [0.0, 0.0, 0.0, 1.0]),
    "door_paper_map_16_worlds": (lambda: cases.cfg_door(2500, 2500, paper=True), 256, [1.0 / 16] * 16),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_belief_graph_equals_oracle(eng_mod, name):
    mk, K, start = CASES[name]
    e, o = grown_pair(eng_mod, mk(), K)
    e.build_belief_graph(start)
    o.build_belief_graph(start)
    assert_same_belief_graph(e, o)
    vis = e.observable_zones()                       # is_zone_observable per node and zone (map_shelves_io.rs:259-265)
    xy = e.tree()[0]
    for i in list(range(0, len(xy), max(1, len(xy) // 200))):
        for z in range(len(e.zone_positions())):
            assert int((int(vis[i]) >> z) & 1) == o.zone_observable(xy[i], z)


def six_door_map():
    """two rooms split by a wall with six doors (64 worlds, 729 reachable beliefs): exercises the many-beliefs kernels
    with more than one world validity"""
    import make_maps
    a = np.full((200, 200), 255, np.uint8)
    make_maps.rect(a, -1.0, 0.0, 1.0, 0.04, 0)
    for x in (-0.85, -0.55, -0.25, 0.05, 0.35, 0.65):
        make_maps.rect(a, x, 0.0, x + 0.12, 0.04, 128)
    z, k = make_maps.door_zone_ids(a)
    assert k == 6
    return a, z


def test_belief_graph_six_doors(eng_mod):
    occ, zones = six_door_map()
    objs = []
    for mk in (eng_mod.Engine, orc.Oracle):
        x = mk()
        x.set_grid(occ, (-1.0, -1.0), (1.0, 1.0), cases.DOOR)
        x.set_zones(zones, 0.45)
        x.set_sampler((-1.0, -1.0), (1.0, 1.0), 11)
        x.set_square_goal(np.array([(-0.5, 0.6)]), np.array([(1 << 64) - 1], dtype=np.uint64), 0.05)
        objs.append(x)
    e, o = objs
    e.grow((0.5, -0.6), 0.05, 5.0, 1500, 1500, batch_K=64, mode=cases.PTO)
    o.grow((0.5, -0.6), 0.05, 5.0, 1500, 1500, batch_K=64, mode=cases.PTO, algo=orc.ALGO_BATCHED_KD)
    assert np.array_equal(e.tree()[1], o.tree()[1])
    prior = [1.0 / 64] * 64
    e.build_belief_graph(prior)
    o.build_belief_graph(prior)
    beliefs, types, _, _ = e.belief_graph(lists=False)
    assert len(beliefs) >= 256 and (types == 2).any() and (types == 1).any()
    assert_same_belief_graph(e, o)


def test_rebuild_with_another_prior_and_after_another_grow(eng_mod):
    case = cases.cfg_door(1500, 1500)
    e, o = grown_pair(eng_mod, case, 64)
    for start in ([0.25] * 4, [0.5, 0.5, 0.0, 0.0], [0.7, 0.1, 0.1, 0.1]):
        e.build_belief_graph(start)
        o.build_belief_graph(start)
        assert_same_belief_graph(e, o)
    cases.grow(e, case, K=64)                         # the sampler moved on: another graph; the old belief graph is gone
    cases.grow(o, case, K=64, algo=orc.ALGO_BATCHED_KD)
    assert e.bg_num_edges() == 0
    e.build_belief_graph([0.25] * 4)
    o.build_belief_graph([0.25] * 4)
    assert_same_belief_graph(e, o)


def test_errors(eng_mod):
    case = cases.cfg_door(500, 500)
    e = cases.configure(eng_mod.Engine(), case)
    with pytest.raises(RuntimeError):
        e.build_belief_graph([0.25] * 4)              # nothing grown yet
    cases.grow(e, case, K=64)
    with pytest.raises(RuntimeError):
        e.build_belief_graph([0.5, 0.5])              # one probability per world
    with pytest.raises(RuntimeError):
        e.build_belief_graph([0.5, 0.5, 0.5, 0.5])    # assert_belief_state_validity
    r = cases.cfg2(2000)
    e2 = cases.configure(eng_mod.Engine(), r)
    cases.grow(e2, r, K=256)
    with pytest.raises(RuntimeError):
        e2.build_belief_graph([1.0])                  # an RRT tree is no PTO graph


def test_properties_at_size(eng_mod):
    """12 shelves of which 8 may hold the object (255 beliefs), several thousand graph nodes, ~10^8 edges: too large for
    the literal oracle in a test, checked through what must hold for any correct result."""
    case = far_sighted(cases.cfg4(8000, 8000), 0.3)
    e = cases.configure(eng_mod.Engine(), case)
    cases.grow(e, case, K=256)
    e.build_belief_graph([0.125] * 8 + [0.0] * 4)
    beliefs, types, (coff, cid), (poff, pid) = e.belief_graph()
    B, N = len(beliefs), e.num_nodes()
    assert B == 2 ** 8 - 1 and len(types) == N * B
    E = len(cid)
    assert E == len(pid) == int(coff[-1]) == int(poff[-1]) == e.bg_num_edges() > 10 ** 7
    # the parents lists are the transpose of the children lists: the same multiset of (from, to) pairs
    src = np.repeat(np.arange(N * B, dtype=np.uint64), np.diff(coff).astype(np.int64))
    dst = np.repeat(np.arange(N * B, dtype=np.uint64), np.diff(poff).astype(np.int64))
    ka = (src << np.uint64(32)) | cid.astype(np.uint64)
    kb = (pid.astype(np.uint64) << np.uint64(32)) | dst
    mix = np.uint64(0x9E3779B97F4A7C15)
    for f in (lambda k: k, lambda k: k * mix, lambda k: (k * mix) ^ (k >> np.uint64(29))):
        assert np.add.reduce(f(ka)) == np.add.reduce(f(kb)) and np.bitwise_xor.reduce(f(ka)) == np.bitwise_xor.reduce(f(kb))
    # within one parents list: observation parents (same graph node) first, by ascending belief; then action parents by ascending node
    same_list = dst[1:] == dst[:-1]
    p_node, d_node = pid.astype(np.uint64) // np.uint64(B), dst // np.uint64(B)
    is_obs = p_node == d_node
    assert not np.any(same_list & ~is_obs[:-1] & is_obs[1:])
    both_obs = same_list & is_obs[:-1] & is_obs[1:]
    assert np.all(pid[1:][both_obs] > pid[:-1][both_obs])
    both_act = same_list & ~is_obs[:-1] & ~is_obs[1:]
    assert np.all(p_node[1:][both_act] > p_node[:-1][both_act])
    # observation nodes: edges stay on the graph node and change the belief; action nodes: the reverse
    tsrc = types[src.astype(np.int64)]
    assert np.all(tsrc != 0)
    obs = tsrc == 2
    c64 = cid.astype(np.uint64)
    assert np.all(src[obs] // B == c64[obs] // B) and np.all(src[obs] % B != c64[obs] % B)
    act = tsrc == 1
    assert np.all(src[act] % B == c64[act] % B) and np.all(src[act] // B != c64[act] // B)
    assert obs.any() and act.any()
    # action edges mirror the PTO graph (one validity in the shelf domain: every belief is compatible with every edge)
    f, t, _ = e.edges()
    pairs = set(zip(f.tolist(), t.tolist())) | set(zip(t.tolist(), f.tolist()))
    sample = np.flatnonzero(act)[:: max(1, int(act.sum()) // 5000)]
    for k in sample:
        assert (int(src[k] // B), int(c64[k] // B)) in pairs
    n_act_nodes = int((types == 1).sum())
    assert int(act.sum()) == sum(int(coff[i + 1] - coff[i]) for i in np.flatnonzero(types == 1)[:1000]) or n_act_nodes > 1000


def test_edge_cases_no_zones_single_world_tiny_graphs(eng_mod):
    """no zones (one world, one belief), a graph of the root alone, a roadmap of the start alone"""
    case = cases.Case(cases.cfg_door(300, 300), zones=None, visibility=0.0, masks=[1], goals=[(0.5, -0.4)])
    e = cases.configure(eng_mod.Engine(), case)
    o = cases.configure(orc.Oracle(), case)
    cases.grow(e, case, K=64)
    cases.grow(o, case, K=64, algo=orc.ALGO_BATCHED_KD)
    e.build_belief_graph([1.0])
    o.build_belief_graph([1.0])
    assert_same_belief_graph(e, o)
    beliefs, types, (coff, cid), _ = e.belief_graph()
    assert len(beliefs) == 1 and not (types == 2).any() and len(cid) > 0
    e.compute_expected_costs()
    assert np.array_equal(e.expected_costs().view(np.uint64), o.expected_costs().view(np.uint64))
    # zero iterations: the graph is the root
    c0 = cases.Case(cases.cfg3(0, 0))
    e0 = cases.configure(eng_mod.Engine(), c0)
    o0 = cases.configure(orc.Oracle(), c0)
    cases.grow(e0, c0, K=64)
    cases.grow(o0, c0, K=64, algo=orc.ALGO_BATCHED_KD)
    assert e0.num_nodes() == 1
    e0.build_belief_graph([0.5, 0.5])
    o0.build_belief_graph([0.5, 0.5])
    assert_same_belief_graph(e0, o0)
    assert e0.bg_num_edges() == 0
    e0.compute_expected_costs()
    assert np.all(np.isinf(e0.expected_costs()))
    with pytest.raises(RuntimeError):
        e0.extract_policy()                                    # no policy from the root
    # a roadmap without samples, and its path queries
    e0.grow_prm((-0.8, -0.8), 0.05, 5.0, 0)
    o0.grow_prm((-0.8, -0.8), 0.05, 5.0, 0)
    assert e0.num_nodes() == 1 and len(e0.edges()[0]) == 0
    pe, po = e0.prm_plan_path((0.0, 0.0), (0.5, 0.5)), o0.prm_plan_path((0.0, 0.0), (0.5, 0.5))
    assert pe.shape == po.shape == (1, 2) and np.array_equal(pe, po)      # start and goal snap to the one node: a path of one state
