"""GPU parity of the PRM* roadmap growth (porrt_grow_prm, through the C ABI) against the oracle's literal restatement of
PRM::init + PRM::grow_graph (src/prm.rs:33-109): the same nodes bit for bit, the same edges in the reference's adjacency
order -- for any size, since the device evaluates the sequential semantics exactly (no batch contract)."""
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


def pair(eng_mod, grid, zones, domain, visibility, seed):
    objs = []
    for mk in (eng_mod.Engine, orc.Oracle):
        x = mk()
        x.set_grid(cases.load_map(grid), (-1.0, -1.0), (1.0, 1.0), domain)
        if zones:
            x.set_zones(cases.load_map(zones), visibility)
        x.set_sampler((-1.0, -1.0), (1.0, 1.0), seed)
        objs.append(x)
    return objs


def assert_same_roadmap(e, o):
    assert e.num_nodes() == o.num_nodes()
    xe, pe, _ = e.tree()
    xo = o.tree()[0]
    assert np.array_equal(xe.view(np.uint64), xo.view(np.uint64)), "nodes differ"
    assert np.all(pe == -1)
    fe, te, ve = e.edges()
    fo, to, vo = o.edges()                    # sequential kd-tree growth: already in the reference's order
    assert len(fe) == len(fo)
    assert np.array_equal(fe, fo) and np.array_equal(te, to) and np.array_equal(ve, vo), "edges differ"


PRM_CASES = {
    # prm.rs:146-170 (map0, no zones), 172-200 (map with zones): max_step 0.1 / 0.05, search_radius 5.0
    "map0_like_no_zones": ("map0_like", None, cases.DOOR, 0.0, 0.1, 5.0, 3000, 0),
    "door_map_two_zones": ("door_map_like", "door_map_like_zone_ids", cases.DOOR, 0.3, 0.05, 5.0, 5000, 1),
    "paper_map_16_worlds": ("paper_map_4", "paper_map_4_zone_ids", cases.DOOR, 0.3, 0.05, 5.0, 4000, 2),
    "shelf_domain_12_zones": ("map5_like", "map5_like_12_goals_zone_ids", cases.SHELF, 0.2, 0.05, 5.0, 6000, 3),
    "benchmark_like_large_radius": ("map_benchmark_like", None, cases.SHELF, 0.0, 0.1, 2.0, 20000, 4),
}


@pytest.mark.parametrize("name", sorted(PRM_CASES))
def test_prm_growth_equals_oracle(eng_mod, name):
    grid, zones, domain, vis, max_step, search_radius, n_iter, seed = PRM_CASES[name]
    e, o = pair(eng_mod, grid, zones, domain, vis, seed)
    start = (0.0, -0.8) if grid != "door_map_like" else (0.5, -0.6)
    e.grow_prm(start, max_step, search_radius, n_iter)
    o.grow_prm(start, max_step, search_radius, n_iter)
    assert e.num_nodes() == n_iter + 1
    assert_same_roadmap(e, o)
    assert len(e.edges()[0]) > n_iter                          # a connected-looking roadmap, not a handful of edges


def test_prm_sampler_moves_on_and_injected_samples(eng_mod):
    e, o = pair(eng_mod, "map0_like", None, cases.DOOR, 0.0, 0)
    for _ in range(2):                                          # the second roadmap continues the sample stream
        e.grow_prm((0.0, 0.0), 0.1, 5.0, 1500)
        o.grow_prm((0.0, 0.0), 0.1, 5.0, 1500)
        assert_same_roadmap(e, o)
    xy = np.random.default_rng(5).uniform(-1.0, 1.0, size=(800, 2))
    xy[100] = xy[50]                                            # a repeated sample: distance 0, still a separate node
    e.set_samples(xy)
    o.set_samples(xy)
    e.grow_prm((0.0, 0.0), 0.1, 5.0, 800)
    o.grow_prm((0.0, 0.0), 0.1, 5.0, 800)
    assert_same_roadmap(e, o)
    assert np.array_equal(e.tree()[0][1:], xy)


def test_prm_then_other_planners_on_the_same_context(eng_mod):
    """a roadmap, then an RRT* tree, then a roadmap again: the context's buffers are shared"""
    case = cases.cfg2(4000)
    e = cases.configure(eng_mod.Engine(), case)
    o = cases.configure(orc.Oracle(), case)
    e.grow_prm((0.0, -0.8), 0.1, 2.0, 3000)
    o.grow_prm((0.0, -0.8), 0.1, 2.0, 3000)
    assert_same_roadmap(e, o)
    cases.grow(e, case, K=256)
    cases.grow(o, case, K=256, algo=orc.ALGO_BATCHED_KD)
    assert np.array_equal(e.tree()[1], o.tree()[1])
    e.grow_prm((0.0, -0.8), 0.1, 2.0, 2000)
    o.grow_prm((0.0, -0.8), 0.1, 2.0, 2000)
    assert_same_roadmap(e, o)


def test_prm_errors(eng_mod):
    e = eng_mod.Engine()
    e.set_sampler((-1.0, -1.0), (1.0, 1.0), 0)
    with pytest.raises(RuntimeError):
        e.grow_prm((0.0, 0.0), 0.1, 5.0, 100)                   # no grid


def test_prm_plan_path_equals_oracle(eng_mod):
    """PRM::plan_path (prm.rs:111-123): nearest nodes, dijkstra from the goal, extract_path"""
    e, o = pair(eng_mod, "map_benchmark_like", None, cases.SHELF, 0.0, 9)
    e.grow_prm((0.0, -0.8), 0.1, 2.0, 8000)
    o.grow_prm((0.0, -0.8), 0.1, 2.0, 8000)
    for start, goal in (((0.0, -0.8), (0.9, 0.0)), ((-0.7, 0.7), (0.7, -0.7)), ((0.3, 0.3), (0.3, 0.3)), ((0.0, -0.8), (2.0, 2.0))):
        pe, po = e.prm_plan_path(start, goal), o.prm_plan_path(start, goal)
        assert pe.shape == po.shape and np.array_equal(pe.view(np.uint64), po.view(np.uint64))
    assert len(e.prm_plan_path((0.0, -0.8), (0.9, 0.0))) > 5
    # a roadmap whose samples are walled off from the start: no path
    e2, o2 = pair(eng_mod, "door_map_like", "door_map_like_zone_ids", cases.DOOR, 0.3, 3)
    e2.grow_prm((0.5, -0.6), 0.05, 5.0, 300)
    o2.grow_prm((0.5, -0.6), 0.05, 5.0, 300)
    pe, po = e2.prm_plan_path((0.5, -0.6), (-0.5, 0.6)), o2.prm_plan_path((0.5, -0.6), (-0.5, 0.6))
    assert pe.shape == po.shape and np.array_equal(pe, po)
    with pytest.raises(RuntimeError):
        cases.configure(eng_mod.Engine(), cases.cfg2(100)).prm_plan_path((0.0, 0.0), (0.5, 0.5))      # no roadmap


def test_edge_order_with_ranks_made_on_the_device(eng_mod):
    """option host_ranks = 0: the kd pre-order ranks that order a node's neighbours come from a kd-tree built on the device (k_kd1_*: a
    level per launch) instead of the host's -- the same edge lists, on a roadmap (shallow tree) and on a belief-space graph (deep tree)"""
    grid, zones, domain, vis, max_step, search_radius, n_iter, seed = PRM_CASES["benchmark_like_large_radius"]
    e, o = pair(eng_mod, grid, zones, domain, vis, seed)
    e.set_option("host_ranks", 0)
    e.grow_prm((0.0, -0.8), max_step, search_radius, n_iter)
    o.grow_prm((0.0, -0.8), max_step, search_radius, n_iter)
    assert_same_roadmap(e, o)
    case = cases.cfg3(3000, 3000)
    res = []
    for host in (1, 0):
        g = cases.configure(eng_mod.Engine(), case)
        g.set_option("host_ranks", host)
        cases.grow(g, case, K=64)
        res.append(g.edges())
    assert all(np.array_equal(a, b) for a, b in zip(*res)) and len(res[0][0]) > 10000
