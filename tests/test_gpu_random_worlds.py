"""Randomised end-to-end parity: random obstacle maps, random shelf / door layouts, random priors.  For every seed the
whole chain -- growth, belief-space expansion, expected costs, policy, PRM roadmap and path -- must equal the oracle's."""
import numpy as np
import pytest

import cases
import make_maps
from oracle import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    from po_rrt_amd import build
    build.build()
    import po_rrt_amd
    return po_rrt_amd


def random_shelf_world(rng):
    """free space with a few walls (high obstacles) and 3..6 low shelves, each a zone"""
    a = np.full((200, 200), 255, np.uint8)
    z = np.full((200, 200), 255, np.uint8)
    for _ in range(int(rng.integers(1, 4))):
        x, y = rng.uniform(-0.8, 0.6, 2)
        if rng.random() < 0.5:
            make_maps.rect(a, x, y, x + rng.uniform(0.2, 0.6), y + 0.04, 0)
        else:
            make_maps.rect(a, x, y, x + 0.04, y + rng.uniform(0.2, 0.6), 0)
    goals, k = [], 0
    for _ in range(int(rng.integers(3, 7))):
        x, y = rng.uniform(-0.8, 0.7, 2)
        if np.any(z[make_maps.to_pixel(x, y + 0.13)[0]:make_maps.to_pixel(x, y + 0.05)[0] + 1] != 255):
            continue                                        # keep the shelves apart (rows are enough for the test)
        make_maps.rect(a, x - 0.1, y + 0.07, x + 0.1, y + 0.13, 200)
        make_maps.rect(z, x - 0.03, y + 0.08, x + 0.03, y + 0.12, k)
        goals.append((x, y))
        k += 1
    make_maps.clear_disk(a, 0.0, -0.9, 0.08)
    return a, z, goals


def configure_pair(eng_mod, a, z, domain, goals, vis, seed):
    out = []
    for mk in (eng_mod.Engine, orc.Oracle):
        x = mk()
        x.set_grid(a, (-1.0, -1.0), (1.0, 1.0), domain)
        x.set_zones(z, vis)
        x.set_sampler((-1.0, -1.0), (1.0, 1.0), seed)
        n = len(goals)
        x.set_square_goal(np.array(goals, dtype=np.float64), np.array([1 << k for k in range(n)], dtype=np.uint64), 0.06)
        out.append(x)
    return out


@pytest.mark.parametrize("seed", range(6))
def test_random_shelf_world_whole_chain(eng_mod, seed):
    rng = np.random.default_rng(1000 + seed)
    a, z, goals = random_shelf_world(rng)
    if len(goals) < 2:
        pytest.skip("degenerate layout")
    e, o = configure_pair(eng_mod, a, z, cases.SHELF, goals, float(rng.uniform(0.25, 0.6)), seed)
    n_iter, K = int(rng.integers(1500, 4000)), int(rng.choice([64, 256]))
    start = (0.0, -0.9)
    re = e.grow(start, 0.05, 5.0, n_iter, n_iter, batch_K=K, mode=cases.PTO)
    ro = o.grow(start, 0.05, 5.0, n_iter, n_iter, batch_K=K, mode=cases.PTO, algo=orc.ALGO_BATCHED_KD)
    assert re == ro and np.array_equal(e.tree()[1], o.tree()[1]) and np.array_equal(e.reach(), o.reach())
    prior = rng.dirichlet(np.ones(len(goals)))
    if rng.random() < 0.5:
        prior[int(rng.integers(len(goals)))] = 0.0           # one shelf known to be empty
        prior = prior / prior.sum()
    prior = list(prior / prior.sum())
    e.build_belief_graph(prior)
    o.build_belief_graph(prior)
    be, te, (ceo, ce), (peo, pe) = e.belief_graph()
    bo, to, (coo, co), (poo, po) = o.belief_graph()
    assert np.array_equal(be.view(np.uint64), bo.view(np.uint64)) and np.array_equal(te, to)
    assert np.array_equal(ceo, coo) and np.array_equal(ce, co) and np.array_equal(peo, poo) and np.array_equal(pe, po)
    e.compute_expected_costs()
    de, do = e.expected_costs(), o.expected_costs()
    assert np.array_equal(de.view(np.uint64), do.view(np.uint64))
    if np.isfinite(do[0]):
        (oid, par, leaf), cost = e.extract_policy()
        oo, po2, lo = o.extract_policy(do)
        assert cost == do[0] and np.array_equal(oid, oo) and np.array_equal(par, po2) and np.array_equal(leaf, lo)
    # a roadmap on the same world
    n_prm = int(rng.integers(1000, 5000))
    e.grow_prm(start, 0.1, 5.0, n_prm)
    o.grow_prm(start, 0.1, 5.0, n_prm)
    assert np.array_equal(e.tree()[0].view(np.uint64), o.tree()[0].view(np.uint64))
    for x, y in zip(e.edges(), o.edges()):
        assert np.array_equal(x, y)
    goal = tuple(rng.uniform(-0.9, 0.9, 2))
    pe2, po3 = e.prm_plan_path(start, goal), o.prm_plan_path(start, goal)
    assert pe2.shape == po3.shape and np.array_equal(pe2.view(np.uint64), po3.view(np.uint64))
