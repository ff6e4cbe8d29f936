"""GPU parity of the multi-modal PRM growth (porrt_grow_mm_prm, through the C ABI) against the oracle's literal loop
(oracle/mmprm.c: MapShelfDomainTampPRM::grow_mm_prm, src/map_shelves_tamp_prm.rs:328-393, one kd-tree PRM per mode)."""
import numpy as np
import pytest

import cases
from oracle import orc

pytestmark = pytest.mark.gpu


def both(case, belief, n_iter_per_belief, max_step=0.1, search_radius=2.0, seed=0):
    import po_rrt_amd
    e = cases.configure(po_rrt_amd.Engine(), cases.Case(case, seed=seed))
    o = cases.configure(orc.Oracle(), cases.Case(case, seed=seed))
    e.set_discrete_seed(seed)
    o.set_discrete_seed(seed)
    return e, o, e.grow_mm_prm(case.start, belief, max_step, search_radius, n_iter_per_belief), o.grow_mm_prm(case.start, belief, max_step, search_radius, n_iter_per_belief)


def assert_same(ge, go):
    assert ge["n_beliefs"] == go["n_beliefs"] and len(ge["modes"]) == len(go["modes"]) and len(ge["transitions"]) == len(go["transitions"])
    for k, (me, mo) in enumerate(zip(ge["modes"], go["modes"])):
        assert np.array_equal(me["belief"].view(np.uint64), mo["belief"].view(np.uint64)), k
        assert me["reaching_probability"] == mo["reaching_probability"]
        assert np.array_equal(me["xy"].view(np.uint64), mo["xy"].view(np.uint64)), "nodes of mode %d" % k
        assert np.array_equal(me["edges"][0], mo["edges"][0]) and np.array_equal(me["edges"][1], mo["edges"][1]), "edges of mode %d" % k
        assert np.array_equal(me["finals"], mo["finals"])
    for te, to in zip(ge["transitions"], go["transitions"]):
        assert (te["zone"], te["from_mode"], te["to_mode"], te["observation"]) == (to["zone"], to["from_mode"], to["to_mode"], to["observation"])
        assert np.array_equal(te["pairs"], to["pairs"])


@pytest.mark.parametrize("seed", [0, 3])
def test_two_shelves(seed):
    case = cases.cfg3(1500, 1500)
    e, o, ge, go = both(case, [0.5, 0.5], 2000, seed=seed)
    assert len(ge["modes"]) == 3 and sum(len(m["xy"]) for m in ge["modes"]) > 5000
    assert_same(ge, go)
    # the continuous sampler is cloned, not advanced; the discrete one moves on: a second call differs from the first, the same on both sides
    ge2, go2 = e.grow_mm_prm(case.start, [0.3, 0.7], 0.1, 2.0, 800), o.grow_mm_prm(case.start, [0.3, 0.7], 0.1, 2.0, 800)
    assert_same(ge2, go2)
    assert e.mm_seconds()["device_s"] > 0


def test_twelve_shelves_mode_tree():
    """the uniform 12-shelf prior: 4095 reachable beliefs, more than a thousand modes created on demand"""
    case = cases.cfg4(1500, 1500)
    e, o, ge, go = both(case, [1.0 / 12] * 12, 20, max_step=0.05, search_radius=5.0)
    assert ge["n_beliefs"] == 4095 and len(ge["modes"]) > 1000
    assert_same(ge, go)


def test_all_modes_at_once_equals_one_by_one(monkeypatch):
    """the batched roadmap pass (k_mm_connect / k_mm_order: every mode's nodes end to end, one launch sequence) against the first
    version, one porrt_grow_prm launch sequence per mode with the grid-binned kernels (developer switch PORRT_MM_ONE_BY_ONE)"""
    import po_rrt_amd
    case = cases.cfg3(1500, 1500)
    res = []
    for one_by_one in (False, True):
        if one_by_one:
            monkeypatch.setenv("PORRT_MM_ONE_BY_ONE", "1")
        e = cases.configure(po_rrt_amd.Engine(), cases.Case(case, seed=1))
        e.set_discrete_seed(5)
        res.append(e.grow_mm_prm(case.start, [0.3, 0.7], 0.1, 2.0, 3000))
    assert sum(len(m["edges"][0]) for m in res[0]["modes"]) > 20000
    assert_same(res[0], res[1])


def test_errors():
    import po_rrt_amd
    case = cases.cfg3(1500, 1500)
    e = cases.configure(po_rrt_amd.Engine(), case)
    with pytest.raises(RuntimeError):
        e.grow_mm_prm(case.start, [0.5, 0.6], 0.1, 2.0, 100)              # check_belief_state
    with pytest.raises(RuntimeError):
        e.grow_mm_prm(case.start, [1.0], 0.1, 2.0, 100)
    d = cases.configure(po_rrt_amd.Engine(), cases.cfg_door())
    with pytest.raises(RuntimeError):
        d.grow_mm_prm((0.5, -0.6), [0.25] * 4, 0.1, 2.0, 100)             # MapShelfDomain only
    g = e.grow_mm_prm(case.start, [1.0, 0.0], 0.1, 2.0, 400)               # a final prior: one mode, no transitions
    assert len(g["modes"]) == 1 and not g["transitions"]
    with pytest.raises(RuntimeError):
        e.tree()                                                           # the single-graph getters have no results now
