"""The oracle's multi-modal PRM (oracle/mmprm.c: MapShelfDomainTampPRM::grow_mm_prm, src/map_shelves_tamp_prm.rs:328-393) against
what the reference's text says must hold; the reference's own tests of this planner need its LFS rasters (:511-597), so the row is
parity unpinned against reference vectors."""
import numpy as np

import cases
from oracle import orc


def test_two_shelves_mode_tree_and_roadmaps():
    case = cases.cfg3(1500, 1500)
    o = cases.configure(orc.Oracle(), case)
    g = o.grow_mm_prm(case.start, [0.5, 0.5], 0.1, 2.0, 600)
    zones = o.zone_positions()
    assert g["n_beliefs"] == 3                                              # map_shelves_io.rs:642-652: 3 reachable beliefs for 2 shelves
    beliefs = [m["belief"].tolist() for m in g["modes"]]
    assert beliefs[0] == [0.5, 0.5] and sorted(beliefs[1:]) == [[0.0, 1.0], [1.0, 0.0]]
    assert [m["reaching_probability"] for m in g["modes"]] == [1.0, 0.5, 0.5]
    assert np.array_equal(g["modes"][0]["xy"][0], np.array(case.start)) and len(g["modes"][0]["finals"]) == 0
    for m in g["modes"][1:]:                                                # a final mode starts with its goal: the shelf that holds the object
        z = int(np.argmax(m["belief"]))
        assert m["finals"].tolist() == [0] and np.array_equal(m["xy"][0], zones[z])
    # 600 * 3 / 200 = 9 batches of 190 samples + 10 observation samples; every batch goes to ONE mode, observation samples also to the successors
    assert sum(len(m["xy"]) for m in g["modes"]) >= 1 + 2 + 9 * 200
    for t in g["transitions"]:
        assert t["observation"] == 1                                        # sic (:227, :268)
        a, b = g["modes"][t["from_mode"]]["xy"], g["modes"][t["to_mode"]]["xy"]
        for i, j in t["pairs"]:
            assert np.array_equal(a[i], b[j])                               # one observation sample, added to both roadmaps
            assert abs(np.hypot(*(a[i] - zones[t["zone"]])) - case.visibility) < 1e-9 or (np.abs(a[i]) >= 0.9998).any()     # on the visibility circle unless clamped
    # every mode draws the same sample sequence (clones of one never-advanced sampler): the first draw of the stream shows up in all three
    first = cases.configure(orc.Oracle(), case).sample()
    for m in g["modes"]:
        assert (m["xy"] == first).all(axis=1).any()
    # a mode's roadmap is PRM::add_sample's: edge j -> i (j < i) iff within heuristic_radius(i + 1) and the segment is free
    m = g["modes"][1]
    ef, et = m["edges"]
    edges = set(zip(ef.tolist(), et.tolist()))
    assert all(f < t for f, t in edges)
    for i in (5, 50, len(m["xy"]) - 1):
        r = orc.lib().orc_heuristic_radius(i + 1, 0.1, 2.0, 2)
        for j in range(i):
            d = float(np.sqrt(((m["xy"][j] - m["xy"][i]) ** 2).sum()))
            assert ((j, i) in edges) == (d <= r and o.traversed_class(m["xy"][j], m["xy"][i]) == 0), (j, i)      # shelf domain: valid iff Free
