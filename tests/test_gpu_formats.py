"""porrt_graph_save_json on a grown graph: the adjacency lists in the file are the reference's (pto.rs:111-120 push order),
rebuilt here from the oracle's literal loop."""
import numpy as np
import pytest

import cases
from oracle import orc
from po_rrt_amd import engine

pytestmark = pytest.mark.gpu


def test_saved_pto_graph_has_the_reference_adjacency(tmp_path):
    import po_rrt_amd
    case = cases.cfg3(800, 3000)
    e = cases.configure(po_rrt_amd.Engine(), case)
    cases.grow(e, case, K=1)
    o = cases.configure(orc.Oracle(), case)
    cases.grow(o, case, K=1, algo=orc.ALGO_SEQ)
    path = str(tmp_path / "graph.json")
    e.save_graph_json(path)
    g = engine.graph_load_json(path)
    xo, _, _ = o.tree()
    n = len(xo)
    assert np.array_equal(g["xy"].view(np.uint64), xo.view(np.uint64))                     # ryu digits round-trip bit for bit
    assert g["node_validity"].tolist() == o.node_validity().tolist()
    fo, to, vo = o.edges()                                                                   # forward edges in the literal push order
    children, parents = [[] for _ in range(n)], [[] for _ in range(n)]
    for node in np.unique(to):                                                               # pto.rs:111-114 then :117-120, per new node
        sel = np.nonzero(to == node)[0]
        for i in sel:
            children[fo[i]].append((int(to[i]), int(vo[i]))); parents[to[i]].append((int(fo[i]), int(vo[i])))
        for i in sel:
            children[to[i]].append((int(fo[i]), int(vo[i]))); parents[fo[i]].append((int(to[i]), int(vo[i])))
    for key, lists in (("children", children), ("parents", parents)):
        off, ids, vals = g[key]
        for node in range(n):
            got = list(zip(ids[off[node]:off[node + 1]].tolist(), vals[off[node]:off[node + 1]].tolist()))
            assert got == lists[node], (key, node)
    val = e.validities()
    assert g["validities"].shape == (len(val), e.n_worlds())
    for v, mask in enumerate(val):
        assert g["validities"][v].tolist() == [bool((int(mask) >> w) & 1) for w in range(e.n_worlds())]
