"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar: integer data (parents, final ids, reach masks, validity ids, edges, iteration counts) bit-exact;
f64 node coordinates and dist_root bit-exact as well (the contract only asks for 1e-9).
"""
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


def run_gpu(eng_mod, case, K, **opts):
    e = eng_mod.Engine()
    for k, v in opts.items():
        e.set_option(k, v)
    cases.configure(e, case)
    rc = cases.grow(e, case, K=K)
    return e, rc


def run_orc(case, K, algo=orc.ALGO_BATCHED_KD):
    o = cases.configure(orc.Oracle(), case)
    rc = cases.grow(o, case, K=K, algo=algo)
    return o, rc


def kd_preorder_rank(xy):
    """pre-order position of every node in the reference's kd-tree (oracle kdtree.c, pinned by nearest_neighbor.rs KATs)"""
    import ctypes as C
    lib = orc.lib()
    xy = np.ascontiguousarray(xy, dtype=np.float64)
    kd = lib.orc_kd_new(xy[0], 0)
    for j in range(1, len(xy)):
        lib.orc_kd_add(kd, xy[j], j)
    out = np.zeros(len(xy), dtype=np.uint64)
    n = lib.orc_kd_radius(kd, xy[0], 1e300, out, len(out))
    lib.orc_kd_free(kd)
    assert n == len(xy)
    rank = np.zeros(len(xy), dtype=np.int64)
    rank[out.astype(np.int64)] = np.arange(len(xy))
    return rank


def assert_same(e, o, pto=False):
    assert e.num_iterations() == o.num_iterations()
    assert e.num_nodes() == o.num_nodes()
    xe, pe, de = e.tree()
    xo, po, do = o.tree()
    assert np.array_equal(xe.view(np.uint64), xo.view(np.uint64)), "node coordinates differ"
    bad = np.nonzero(pe != po)[0]
    assert bad.size == 0, "parents differ at %s" % bad[:8]
    assert np.array_equal(de.view(np.uint64), do.view(np.uint64)), "dist_root differs"
    assert np.array_equal(e.final_ids(), o.final_ids())
    assert np.array_equal(e.final_masks(), o.final_masks())
    assert e.metrics()["n_tie_fallbacks"] == 0
    if pto:
        assert np.array_equal(e.reach(), o.reach())
        assert np.array_equal(e.node_validity(), o.node_validity())
        fe, te, ve = e.edges()
        fo, to, vo = o.edges()
        # adjacency order (pto.rs:103-114): new nodes ascending, neighbours of one node in kd pre-order
        rank = kd_preorder_rank(o.tree()[0])
        order = np.lexsort((rank[fo], to))
        assert np.array_equal(fe, fo[order]) and np.array_equal(te, to[order]) and np.array_equal(ve, vo[order])
        assert e.is_final_set_complete() == o.is_final_set_complete()


def test_device_sqrt_and_divide_are_ieee(eng_mod):
    e = eng_mod.Engine()
    assert e.selftest(1 << 21) == (0, 0)


RRT_SMALL = [cases.empty_space(1000, 10000), cases.cfg1(3000), cases.cfg2(4000), cases.cfg2(3000, seed=3, grid="map_benchmark_like_c"),
             cases.cfg2_obs(1500)]


@pytest.mark.parametrize("K", [1, 64, 1024])
@pytest.mark.parametrize("case", RRT_SMALL, ids=lambda c: c.name)
def test_rrt_matches_oracle(eng_mod, case, K):
    if K == 1:
        case = cases.Case(case)
        case.update(n_iter_min=min(case.n_iter_min, 600), n_iter_max=min(case.n_iter_max, 1200))
    e, _ = run_gpu(eng_mod, case, K)
    o, _ = run_orc(case, K)
    assert e.num_nodes() > 50
    assert_same(e, o)
    be, bo = e.best_solution(), o.best_solution()
    assert (be is None) == (bo is None)
    if be is not None:
        assert np.array_equal(be[0], bo[0]) and be[1] == bo[1]


@pytest.mark.parametrize("gl", [16, 32, 64])
@pytest.mark.parametrize("K", [64, 1024])
@pytest.mark.parametrize("case", RRT_SMALL, ids=lambda c: c.name)
def test_rrt_group_kernels_match_oracle(eng_mod, case, K, gl):
    """the step kernels with several samples per wave (k_nn2 / k_conn2: what porrt_grow_batch uses from 8 queries on), forced
    for a single query, at every group size"""
    e, _ = run_gpu(eng_mod, case, K, group_lanes=gl)
    o, _ = run_orc(case, K)
    assert e.num_nodes() > 50
    assert_same(e, o)


def test_rrt_group_kernels_full_size_and_eager(eng_mod):
    """configs[1] at full size through k_nn2 / k_conn2 (hipGraph replay), and a shorter run launched eagerly"""
    case = cases.cfg2(125000)
    e, _ = run_gpu(eng_mod, case, 1024, group_lanes=16)
    o, _ = run_orc(case, 1024)
    assert e.num_nodes() > 90000
    assert_same(e, o)
    case = cases.cfg2(30000, seed=5)
    e, _ = run_gpu(eng_mod, case, 1024, group_lanes=16, graph=0)
    o, _ = run_orc(case, 1024)
    assert_same(e, o)


def test_rrt_k1_is_the_reference_loop(eng_mod):
    """K = 1 against the literal sequential restatement (kd-tree and all)."""
    case = cases.cfg2(1500)
    e, _ = run_gpu(eng_mod, case, 1)
    o, _ = run_orc(case, 1, algo=orc.ALGO_SEQ)
    assert_same(e, o)


PTO_SMALL = [cases.cfg3(1500, 20000), cases.cfg4(1500, 4000), cases.cfg_door(1200, 20000), cases.cfg_door(1500, 6000, paper=True)]


@pytest.mark.parametrize("K", [1, 16, 256])
@pytest.mark.parametrize("case", PTO_SMALL, ids=lambda c: c.name)
def test_pto_matches_oracle(eng_mod, case, K):
    if K == 1:
        case = cases.Case(case)
        case.update(n_iter_min=min(case.n_iter_min, 500), n_iter_max=min(case.n_iter_max, 1500))
    e, rce = run_gpu(eng_mod, case, K)
    o, rco = run_orc(case, K)
    assert rce == rco
    assert_same(e, o, pto=True)


def test_pto_k1_edge_order_is_the_reference_loop(eng_mod):
    """K = 1 against the literal loop: the edges come back in exactly the order the reference adds them."""
    case = PTO_SMALL[0]
    e, _ = run_gpu(eng_mod, case, 1)
    o, _ = run_orc(case, 1, algo=orc.ALGO_SEQ)
    fe, te, ve = e.edges()
    fo, to, vo = o.edges()
    assert np.array_equal(fe, fo) and np.array_equal(te, to) and np.array_equal(ve, vo)


def test_pto_k1_is_the_reference_loop(eng_mod):
    case = cases.cfg3(800, 3000)
    e, rce = run_gpu(eng_mod, case, 1)
    o, rco = run_orc(case, 1, algo=orc.ALGO_SEQ)
    assert rce == rco
    # ref_seq lists neighbours in kd pre-order; both sides are compared sorted by (to, from)
    assert_same(e, o, pto=True)


def test_headline_config_full_size(eng_mod):
    """BASELINE.json configs[1]: map_benchmark-like, K=1024, ~100k-node tree, against the oracle."""
    case = cases.cfg2(125000)
    e, _ = run_gpu(eng_mod, case, 1024)
    o, _ = run_orc(case, 1024)
    assert e.num_nodes() > 90000
    assert_same(e, o)


@pytest.mark.parametrize("opts", [dict(graph=0), dict(kd_group=1), dict(kd_group=4), dict(graph=0, kd_group=3), dict(profile=1), dict(pipeline=0),
                                  dict(pipeline=0, graph=0), dict(pipeline=0, kd_group=1)],
                         ids=lambda o: ",".join("%s=%s" % kv for kv in o.items()))
def test_launch_modes_do_not_change_results(eng_mod, opts):
    """hipGraph replay vs eager launches, how many steps' nodes enter the kd tie-order structure together (the
    structure lags the steps; equal-cost parents that need it are settled later), and pipelined steps (k_step_rrt, the
    default) vs one kernel after the other (pipeline=0), must give the same tree."""
    case = cases.cfg2(30000)
    e0, _ = run_gpu(eng_mod, case, 1024)
    e1, _ = run_gpu(eng_mod, case, 1024, **opts)
    assert_same(e1, e0)
    o, _ = run_orc(case, 1024)
    assert_same(e1, o)


@pytest.mark.parametrize("K", [1, 64, 333, 1024])
def test_pipelined_steps(eng_mod, K):
    """option pipeline: connect(b) and search(b + 1) in one launch (k_step_rrt), filing and rewire phase 2 between two of them
    (k_file_commit) -- small cases, odd batch sizes, the stepwise tail after n_iter_min, a batch of contexts, and a context
    that goes back and forth between the two forms"""
    c = cases.cfg2(50)
    c.update(n_iter_min=50, n_iter_max=30000)                  # stops on the goal: steps beyond n_iter_min launched one by one
    for pipe in (1, 0):
        for case in RRT_SMALL + [c]:
            e, _ = run_gpu(eng_mod, case, K, pipeline=pipe)
            o, _ = run_orc(case, K)
            assert_same(e, o)
    if K >= 64:
        cs = [cases.cfg2(6000, seed=s) for s in (0, 1, 2)]
        engs = [cases.configure(eng_mod.Engine(), cc) for cc in cs]
        engs[0].set_option("pipeline", 1)
        eng_mod.Engine.grow_batch(engs, [cc.start for cc in cs], cs[0].max_step, cs[0].search_radius, cs[0].n_iter_min, K)
        for e, cc in zip(engs, cs):
            o, _ = run_orc(cc, K)
            assert_same(e, o)
        e = engs[1]                                             # was a member of a pipelined batch; now alone, off, on
        for pipe in (0, 1):
            e.set_option("pipeline", pipe)
            e.set_sampler((-1.0, -1.0), (1.0, 1.0), cs[1].seed)
            cases.grow(e, cs[1], K=K)
            o, _ = run_orc(cs[1], K)
            assert_same(e, o)


@pytest.mark.parametrize("graph", [1, 0])
def test_grow_batch_equals_separate_grows(eng_mod, graph):
    """porrt_grow_batch (one launch sequence, one grid row per context) == the same contexts grown one by one."""
    cs = [cases.cfg2(20000, seed=s) for s in (0, 1, 2)] + [cases.cfg2(20000, seed=3, grid="map_benchmark_like_c")]
    single = []
    for c in cs:
        e, _ = run_gpu(eng_mod, c, 1024)
        single.append(e)
    engs = []
    for c in cs:
        e = eng_mod.Engine()
        e.set_option("graph", graph)
        engs.append(cases.configure(e, c))
    for rep in range(2):                    # second round: sampler states moved on, the cached graph is replayed
        eng_mod.Engine.grow_batch(engs, [c.start for c in cs], cs[0].max_step, cs[0].search_radius, cs[0].n_iter_min, 1024)
        costs = eng_mod.Engine.best_cost_batch(engs)          # one launch for the batch's members ...
        for e, c in zip(engs, costs):
            sol = e.best_solution()
            assert (sol is None and np.isinf(c)) or (sol is not None and c == sol[1] == e.best_cost())
        if rep == 0:
            mixed = eng_mod.Engine.best_cost_batch([engs[2], single[0]])      # ... one by one for any other set
            assert mixed[0] == costs[2] and mixed[1] == costs[0]
            for e, s in zip(engs, single):
                assert_same(e, s)
        else:
            for e, c in zip(engs, cs):
                o = cases.configure(orc.Oracle(), c)
                cases.grow(o, c, K=1024, algo=orc.ALGO_BATCHED_KD)
                cases.grow(o, c, K=1024, algo=orc.ALGO_BATCHED_KD)          # the oracle's sampler moves on the same way
                assert_same(e, o)


def test_grow_batch_of_eight_uses_the_group_kernels(eng_mod):
    """from 8 contexts on porrt_grow_batch picks k_nn2 / k_conn2 by itself (XCD-aware rows need a multiple of 8; 9 do not have one)"""
    for n in (8, 9):
        cs = [cases.cfg2(12000, seed=s, grid="map_benchmark_like_%s" % "abcdefghi"[s]) for s in range(n)]
        engs = [cases.configure(eng_mod.Engine(), c) for c in cs]
        eng_mod.Engine.grow_batch(engs, [c.start for c in cs], cs[0].max_step, cs[0].search_radius, cs[0].n_iter_min, 1024)
        for e, c in zip(engs, cs):
            o = cases.configure(orc.Oracle(), c)
            cases.grow(o, c, K=1024, algo=orc.ALGO_BATCHED_KD)
            assert_same(e, o)


def test_grow_batch_in_sub_batches(eng_mod):
    """batch_streams: the contexts advance as several launch sequences side by side (host thread, streams and hipGraph per
    sub-batch); nothing in a tree may depend on it, and porrt_best_cost_batch serves the sub-batches' members run by run."""
    cs = [cases.cfg2(9000, seed=s, grid="map_benchmark_like_%s" % "abcdefghi"[s % 9]) for s in range(7)]
    ref = []
    for c in cs:
        o = cases.configure(orc.Oracle(), c)
        cases.grow(o, c, K=1024, algo=orc.ALGO_BATCHED_KD)
        ref.append(o)
    for streams in (2, 3, 8):
        engs = [cases.configure(eng_mod.Engine(), c) for c in cs]
        engs[0].set_option("batch_streams", streams)
        engs[0].set_option("group_lanes", 16)
        for rep in range(2):
            for j, e in enumerate(engs):
                e.set_sampler((-1.0, -1.0), (1.0, 1.0), cs[j].seed)
            eng_mod.Engine.grow_batch(engs, [c.start for c in cs], cs[0].max_step, cs[0].search_radius, cs[0].n_iter_min, 1024)
            costs = eng_mod.Engine.best_cost_batch(engs)
            for e, o, c in zip(engs, ref, costs):
                assert_same(e, o)
                sol = o.best_solution()
                assert (sol is None and np.isinf(c)) or (sol is not None and c == sol[1])


def test_get_trees_equals_get_tree(eng_mod):
    """porrt_get_trees (worker threads, pinned staging, more contexts than workers) == porrt_get_tree per context"""
    cs = [cases.cfg2(3000 + 700 * s, seed=s) for s in range(11)]
    engs = []
    for c in cs:
        e, _ = run_gpu(eng_mod, c, 256)
        engs.append(e)
    for rep in range(2):                       # the second call reuses the staging
        got = eng_mod.Engine.trees(engs)
        for e, (xy, parent, dist) in zip(engs, got):
            rxy, rparent, rdist = e.tree()
            assert np.array_equal(xy.view(np.uint64), rxy.view(np.uint64)) and np.array_equal(parent, rparent)
            assert np.array_equal(dist.view(np.uint64), rdist.view(np.uint64))
    assert len(eng_mod.Engine.trees(engs[:1])[0][1]) == engs[0].num_nodes()
    bufs = [(np.zeros((12000, 2)), np.zeros(12000, dtype=np.int64), np.zeros(12000)) for _ in engs]       # the caller's own memory
    for e, (xy, parent, dist) in zip(engs, eng_mod.Engine.trees(engs, bufs)):
        rxy, rparent, rdist = e.tree()
        assert np.array_equal(xy, rxy) and np.array_equal(parent, rparent) and np.array_equal(dist, rdist) and xy.base is not None


def test_grow_batch_splits_by_itself_from_32_contexts(eng_mod):
    """33 RRT* contexts (sub-batches of 16 and 17 rows, group kernels) and 32 belief-space contexts (sub-batches of the PTO kernels):
    the default batch_streams, nothing set"""
    cs = [cases.cfg2(4000, seed=s, grid="map_benchmark_like_%s" % "abcdefghi"[s % 9]) for s in range(33)]
    engs = [cases.configure(eng_mod.Engine(), c) for c in cs]
    eng_mod.Engine.grow_batch(engs, [c.start for c in cs], cs[0].max_step, cs[0].search_radius, cs[0].n_iter_min, 512)
    costs = eng_mod.Engine.best_cost_batch(engs)
    for e, c, bc in zip(engs, cs, costs):
        o, _ = run_orc(c, 512)
        assert_same(e, o)
        sol = o.best_solution()
        assert (sol is None and np.isinf(bc)) or (sol is not None and bc == sol[1])
    ps = [cases.cfg3(1500, 1500, seed=s) for s in range(32)]
    pengs = [cases.configure(eng_mod.Engine(), c) for c in ps]
    eng_mod.Engine.grow_batch(pengs, [c.start for c in ps], ps[0].max_step, ps[0].search_radius, 1500, 128, mode=cases.PTO)
    for e, c in zip(pengs, ps):
        o, _ = run_orc(c, 128)
        assert_same(e, o, pto=True)


def test_grow_batch_pto(eng_mod):
    cs = [cases.cfg3(6000, 6000, seed=s) for s in (0, 1)]
    engs = [cases.configure(eng_mod.Engine(), c) for c in cs]
    eng_mod.Engine.grow_batch(engs, [c.start for c in cs], cs[0].max_step, cs[0].search_radius, 6000, 256, mode=cases.PTO)
    for e, c in zip(engs, cs):
        o, _ = run_orc(c, 256)
        assert_same(e, o, pto=True)


def test_injected_samples_equal_seeded_stream(eng_mod):
    case = cases.cfg1(2000)
    e1, _ = run_gpu(eng_mod, case, 256)
    s = cases.configure(orc.Oracle(), case)
    xy = np.array([s.sample() for _ in range(2000 - 20)])
    e2 = eng_mod.Engine()
    cases.configure(e2, case)
    e2.set_samples(xy)
    cases.grow(e2, case, K=256)
    for a, b in zip(e1.tree(), e2.tree()):
        assert np.array_equal(a, b)


def _injected(case, xy, K, eng_mod, **opts):
    e = eng_mod.Engine()
    for k, v in opts.items():
        e.set_option(k, v)
    cases.configure(e, case)
    e.set_samples(xy)
    cases.grow(e, case, K=K)
    o = cases.configure(orc.Oracle(), case)
    o.set_samples(xy)
    cases.grow(o, case, K=K, algo=orc.ALGO_BATCHED_KD)
    return e, o


def test_dense_cluster_and_exact_duplicates(eng_mod):
    """Every sample inside one small square: one region of the page grid takes all nodes (page chains through the
    directory), neighbour lists are long, and a third of the samples are exact copies of earlier ones (kd chains of
    identical points, equal-cost parents everywhere)."""
    rng = np.random.default_rng(7)
    n = 6000
    xy = np.stack([rng.uniform(-0.02, 0.02, n), rng.uniform(-0.92, -0.88, n)], axis=1)
    xy[::3] = xy[(np.arange(0, n, 3) // 7) * 2 + 1]          # exact duplicates of other samples
    case = cases.cfg2(n - n // 100 - 5)
    for K in (1024, 4096):
        e, o = _injected(case, xy, K, eng_mod)
        assert_same(e, o)
        assert e.num_nodes() > 3000


@pytest.mark.parametrize("K", [7, 100, 333])
def test_odd_batch_sizes_and_early_termination(eng_mod, K):
    """batch_K that is no multiple of the wave size, n_iter_min below one batch, and a loop condition (rrt.rs:109,
    pto.rs:67) that ends the growth between n_iter_min and n_iter_max: the steps beyond n_iter_min are launched one by
    one with the condition re-evaluated in between, as ref_batched does."""
    c = cases.cfg2(50)
    c.update(n_iter_min=50, n_iter_max=30000)
    e, _ = run_gpu(eng_mod, c, K)
    o, _ = run_orc(c, K)
    assert_same(e, o)
    assert e.num_iterations() < 30000, "the case is meant to stop on the goal, not on the budget"
    p = cases.cfg3(40, 9000)
    e, _ = run_gpu(eng_mod, p, K)
    o, _ = run_orc(p, K)
    assert_same(e, o, pto=True)


def test_best_cost_on_the_device_equals_the_host_walk(eng_mod):
    """porrt_best_cost (no tree download) against porrt_best_solution and the oracle: bit-identical cost."""
    for case in (cases.cfg2(30000), cases.cfg1(3000), cases.empty_space(1000, 10000)):
        e, _ = run_gpu(eng_mod, case, 1024)
        bc = e.best_cost()
        sol = e.best_solution()
        o, _ = run_orc(case, 1024)
        so = o.best_solution()
        if sol is None:
            assert bc is None and so is None
        else:
            assert bc == sol[1] == so[1]
    c = cases.cfg2(300)                       # far too few iterations to reach the goal
    e, _ = run_gpu(eng_mod, c, 64)
    assert e.best_cost() is None and e.best_solution() is None


def test_max_batch_and_ragged_tail(eng_mod):
    """batch_K = 4096 (the kd claim kernel's capacity per launch) with an iteration count that is no multiple of it."""
    case = cases.cfg2(4096 * 3 + 17)
    e, _ = run_gpu(eng_mod, case, 4096)
    o, _ = run_orc(case, 4096)
    assert_same(e, o)


def test_points_outside_the_sampler_box(eng_mod):
    """The goal (re-sampled every 100th iteration) lies outside the sampler's box, and so do the nodes steered towards
    it: they fall into the clamped border cells of the region grid, the bound pyramid and the kd hint grid."""
    case = cases.cfg2(12000)
    for K in (256, 1024):
        e = eng_mod.Engine()
        cases.configure(e, case)
        e.set_sampler((-0.5, -1.0), (0.5, 0.2), 3)
        cases.grow(e, case, K=K)
        o = cases.configure(orc.Oracle(), case)
        o.set_sampler((-0.5, -1.0), (0.5, 0.2), 3)
        cases.grow(o, case, K=K, algo=orc.ALGO_BATCHED_KD)
        assert_same(e, o)
        xy = e.tree()[0]
        assert (xy[:, 0] > 0.5).any(), "the case is meant to put nodes outside the box"


def test_sampler_state_persists_across_grows(eng_mod):
    """tamp_rrt.rs:196-232: one RRT object (one RNG stream) serves many plans."""
    case = cases.cfg1(700)
    e = cases.configure(eng_mod.Engine(), case)
    o = cases.configure(orc.Oracle(), case)
    for _ in range(3):
        cases.grow(e, case, K=128)
        cases.grow(o, case, K=128, algo=orc.ALGO_BATCHED_KD)
        assert_same(e, o)


def test_errors(eng_mod):
    from po_rrt_amd import PorrtError
    case = cases.cfg3(100, 100)
    case.update(start=(-0.22, 0.0))            # inside a wall: pto.rs:61
    e = cases.configure(eng_mod.Engine(), case)
    with pytest.raises(PorrtError) as ei:
        cases.grow(e, case, K=16)
    assert ei.value.code == -2
    e = eng_mod.Engine()
    with pytest.raises(PorrtError):
        e.grow((0.0, 0.0), 0.1, 1.0, 10, 10, batch_K=0)
    with pytest.raises(PorrtError):
        e.set_square_goal([[0, 0], [1, 1]], [1, 1], 0.1)     # overlapping validities (common.rs:321)
    with pytest.raises(PorrtError):
        e.tree()                                             # nothing grown yet


def test_neighbour_list_regrowth(eng_mod):
    """A tiny initial capacity forces the overflow -> regrow -> replay path; results are unchanged."""
    case = cases.cfg1(2500)
    e1, _ = run_gpu(eng_mod, case, 256)
    e2, _ = run_gpu(eng_mod, case, 256, cand_cap=64)
    for a, b in zip(e1.tree(), e2.tree()):
        assert np.array_equal(a, b)


def test_size_independent_properties(eng_mod):
    """Properties that hold at any size: every parent exists, no cycles, dist_root of a node is at least
    the straight-line distance to the root, every edge of the tree is collision-free on the grid except
    the reference's unchecked fallback edge."""
    case = cases.cfg2(30000)
    e, _ = run_gpu(eng_mod, case, 1024)
    xy, parent, dist = e.tree()
    n = len(parent)
    assert parent[0] == -1 and (parent[1:] >= 0).all() and (parent[1:] < n).all()
    hops = np.zeros(n, dtype=np.int64)
    cur = parent.copy()
    for _ in range(n):
        live = cur >= 0
        if not live.any():
            break
        hops[live] += 1
        cur[live] = parent[cur[live]]
    assert not (cur >= 0).any(), "cycle in parents"
    straight = np.sqrt(((xy - xy[0]) ** 2).sum(axis=1))
    assert (dist + 1e-9 >= straight).all()


def test_cfg5_maps_and_seeds(eng_mod):
    """BASELINE.json configs[4]: the nine map_benchmark_{a..i} stand-ins x several seeds, one context reused
    for all queries of a map (the reference reuses one RRT object, tamp_rrt.rs:196-232)."""
    for letter in "abcdefghi":
        case = cases.cfg2(6000, grid="map_benchmark_like_%s" % letter)
        e = cases.configure(eng_mod.Engine(), case)
        o = cases.configure(orc.Oracle(), case)
        for seed in (0, 7):
            e.set_sampler((-1.0, -1.0), (1.0, 1.0), seed)
            o.set_sampler((-1.0, -1.0), (1.0, 1.0), seed)
            cases.grow(e, case, K=1024)
            cases.grow(o, case, K=1024, algo=orc.ALGO_BATCHED_KD)
            assert_same(e, o)
