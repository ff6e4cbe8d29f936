"""Frozen answers (tests/golden/trees, written by tools/make_golden_trees.py from the CPU oracle -- oracle-generated, not
reference-generated): the oracle still reproduces them (CPU), and the HIP path reproduces them without the oracle in the
loop (GPU)."""
import os

import numpy as np
import pytest

import cases
import make_golden_trees as mg
from oracle import orc

SPECS = {name: (case, K, algo) for name, case, K, algo in mg.specs()}


def load(name):
    return np.load(os.path.join(mg.OUT, name + ".npz"))


@pytest.mark.parametrize("name", sorted(SPECS))
def test_oracle_reproduces_the_frozen_answers(name):
    case, K, algo = SPECS[name]
    rec, gold = mg.build(name, case, K, algo), load(name)
    assert sorted(rec) == sorted(gold.files)
    for k in gold.files:
        assert np.array_equal(np.asarray(rec[k]), gold[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SPECS))
def test_hip_path_reproduces_the_frozen_answers(name):
    from po_rrt_amd import build
    build.build()
    import po_rrt_amd
    case, K, algo = SPECS[name]
    gold = load(name)
    e = cases.configure(po_rrt_amd.Engine(), case)
    cases.grow(e, case, K=K)
    xy, parent, dist = e.tree()
    if name.startswith("pto_cfg4"):
        # configs[3] at bench size, real parameters: graph, 255-belief expansion, expected costs and the policy against the frozen
        # answers of the oracle (which takes ~35 s for them)
        assert len(xy) == int(gold["n_nodes"])
        assert mg.digest(xy, e.reach(), e.node_validity(), e.final_ids().astype(np.uint64)) == str(gold["node_digest"])
        e.build_belief_graph(mg.CFG4_PRIOR)
        beliefs, types, (coff, cid), (poff, pid) = e.belief_graph()
        assert len(cid) == int(gold["n_belief_edges"]) and mg.digest(beliefs, types, coff, cid, poff, pid) == str(gold["belief_digest"])
        e.compute_expected_costs()
        d = e.expected_costs()
        assert mg.digest(d) == str(gold["cost_digest"]) and np.array_equal(d[:1].view(np.uint64), gold["root_cost_bits"])
        (oid, par, leaf), cost = e.extract_policy()
        assert np.array_equal(oid, gold["policy_ids"]) and np.array_equal(par, gold["policy_parents"]) and np.array_equal(leaf, gold["policy_leaf"])
        assert cost == d[0] and leaf.sum() == 8                       # one leaf per possible world
        return
    assert np.array_equal(xy.view(np.uint64), gold["xy_bits"]) and np.array_equal(parent.astype(np.int32), gold["parent"])
    assert np.array_equal(dist.view(np.uint64), gold["dist_bits"]) and np.array_equal(e.final_ids().astype(np.uint64), gold["final_ids"])
    if case.mode == cases.PTO:
        assert np.array_equal(e.reach(), gold["reach"])
        f, t, v = e.edges()
        if algo == orc.ALGO_SEQ:                       # the sequential oracle logs edges in the reference's order, as the engine returns them
            assert mg.digest(f, t, v) == str(gold["edge_digest"])
        if "belief_digest" in gold.files:
            e.build_belief_graph([0.5, 0.5])
            beliefs, types, (coff, cid), (poff, pid) = e.belief_graph()
            assert mg.digest(beliefs, types, coff, cid, poff, pid) == str(gold["belief_digest"])
            e.compute_expected_costs()
            d = e.expected_costs()
            assert mg.digest(d) == str(gold["cost_digest"]) and np.array_equal(d[:1].view(np.uint64), gold["root_cost_bits"])
            (oid, par, leaf), _ = e.extract_policy()
            assert np.array_equal(oid, gold["policy_ids"]) and np.array_equal(par, gold["policy_parents"])
