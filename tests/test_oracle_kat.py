"""Known-answer tests that PIN the CPU oracle.

Each test restates a data-free unit test of the reference (file:line given) with the
same inputs and expected values, or a published vector of a third-party algorithm the
reference calls.  They run on CPU (`-m "not gpu"`).
"""
import itertools
import math

import numpy as np
import pytest

from oracle import orc

# ---------------------------------------------------------------- kd-tree
# src/nearest_neighbor.rs:142-167 create_tree (geeksforgeeks 7-point example)
NODES = [[3.0, 6.0], [17.0, 15.0], [13.0, 15.0], [6.0, 12.0], [9.0, 1.0], [2.0, 7.0], [10.0, 19.0]]
CENTERS = [[17.0, 15.0], [9.1, 1.0], [2.0, 8.0], [15.0, 13.0], [3.0, 5.0], [13.0, 7.0]]


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


@pytest.fixture()
def kd(oracle_lib):
    L = orc.lib()
    t = L.orc_kd_new(f64(NODES[0]), 0)
    for i, n in enumerate(NODES[1:]):
        L.orc_kd_add(t, f64(n), i + 1)
    yield t
    L.orc_kd_free(t)


def test_kdtree_creation(oracle_lib):
    # nearest_neighbor.rs:169-176
    L = orc.lib()
    t = L.orc_kd_new(f64([3.0, 6.0]), 0)
    assert L.orc_kd_child(t, 0, 0) == -1 and L.orc_kd_child(t, 0, 1) == -1
    L.orc_kd_free(t)


def test_add_second_level(oracle_lib):
    # nearest_neighbor.rs:178-194
    L = orc.lib()
    t = L.orc_kd_new(f64([3.0, 6.0]), 0)
    L.orc_kd_add(t, f64([2.0, 7.0]), 1)
    assert L.orc_kd_child(t, 0, 1) == -1 and L.orc_kd_child(t, 0, 0) == 1
    L.orc_kd_free(t)
    t = L.orc_kd_new(f64([3.0, 6.0]), 0)
    L.orc_kd_add(t, f64([17.0, 15.0]), 1)
    assert L.orc_kd_child(t, 0, 0) == -1 and L.orc_kd_child(t, 0, 1) == 1
    L.orc_kd_free(t)


def test_full_tree(kd):
    # nearest_neighbor.rs:196-234: shape of the 7-point tree
    L = orc.lib()
    assert L.orc_kd_child(kd, 0, 0) == 5
    assert L.orc_kd_child(kd, 0, 1) == 1
    assert L.orc_kd_child(kd, 1, 0) == 3
    assert L.orc_kd_child(kd, 1, 1) == 2
    assert L.orc_kd_child(kd, 3, 1) == 4
    assert L.orc_kd_child(kd, 2, 0) == 6
    s = np.zeros(2)
    L.orc_kd_state(kd, 4, s)
    assert list(s) == [9.0, 1.0]


def test_nearest_neighbor_and_radius(kd):
    # nearest_neighbor.rs:237-265: NN and radius sets equal brute force
    L = orc.lib()
    out = np.zeros(16, dtype=np.uint64)
    for c in CENTERS:
        d = [math.sqrt((n[0] - c[0]) ** 2 + (n[1] - c[1]) ** 2) for n in NODES]
        order = sorted(range(len(NODES)), key=lambda i: d[i])
        assert L.orc_kd_nearest(kd, f64(c), None, 0) == order[0]
        for radius in range(1, 10):
            expect = sorted(i for i in order if d[i] <= radius)
            n = L.orc_kd_radius(kd, f64(c), float(radius), out, 16)
            assert sorted(int(x) for x in out[:n]) == expect


def test_nearest_neighbor_with_filter(kd):
    # nearest_neighbor.rs:267-311: exclusion order from two query points
    L = orc.lib()

    def nn(q, excl):
        e = np.array(excl, dtype=np.uint64)
        return L.orc_kd_nearest_excluding(kd, f64(q), e if len(excl) else np.zeros(1, dtype=np.uint64), len(excl))

    expect0 = [0, 5, 3, 4, 2, 6, 1]
    for k in range(7):
        assert nn([3.1, 6.0], expect0[:k]) == expect0[k]
    expect2 = [2, 1, 6, 3]
    for k in range(4):
        assert nn([13.0, 15.1], expect2[:k]) == expect2[k]


def test_filtered_nn_returns_root_when_nothing_passes(kd):
    # nearest_neighbor.rs:90: `nearest` starts as the root
    L = orc.lib()
    assert L.orc_kd_nearest_excluding(kd, f64([10.0, 10.0]), np.arange(7, dtype=np.uint64), 7) == 0


# ---------------------------------------------------------- reachability
def bits(*b):
    """bitvec![b0, b1, ...] -> mask with bit i = b_i"""
    return sum(int(v) << i for i, v in enumerate(b))


def make_reach(oracle_lib, root, nodes, n_worlds=2):
    L = orc.lib()
    r = L.orc_reach_new()
    L.orc_reach_set_root(r, root, n_worlds)
    for n in nodes:
        L.orc_reach_add_node(r, n)
    return L, r


def test_reachability_chain(oracle_lib):
    # pto_reachability.rs:109-135
    L, r = make_reach(oracle_lib, bits(1, 1), [bits(1, 0), bits(1, 0), bits(0, 1)])
    L.orc_reach_add_edge(r, 0, 1, bits(1, 0))
    L.orc_reach_add_edge(r, 1, 2, bits(1, 0))
    L.orc_reach_add_edge(r, 1, 3, bits(0, 1))
    assert [L.orc_reach_get(r, i) for i in range(4)] == [bits(1, 1), bits(1, 0), bits(1, 0), bits(0, 0)]
    L.orc_reach_free(r)


def test_reachability_diamond(oracle_lib):
    # pto_reachability.rs:137-164
    L, r = make_reach(oracle_lib, bits(1, 1), [bits(1, 0), bits(0, 1), bits(1, 1)])
    L.orc_reach_add_edge(r, 0, 1, bits(1, 0))
    L.orc_reach_add_edge(r, 0, 2, bits(0, 1))
    L.orc_reach_add_edge(r, 1, 3, bits(1, 1))
    L.orc_reach_add_edge(r, 2, 3, bits(1, 1))
    assert [L.orc_reach_get(r, i) for i in range(4)] == [bits(1, 1), bits(1, 0), bits(0, 1), bits(1, 1)]
    L.orc_reach_free(r)


def test_final_nodes_completeness(oracle_lib):
    # pto_reachability.rs:166-196
    L, r = make_reach(oracle_lib, bits(1, 1), [bits(1, 1), bits(1, 0), bits(0, 1)])
    L.orc_reach_add_edge(r, 0, 1, bits(1, 1))
    L.orc_reach_add_edge(r, 1, 2, bits(1, 0))
    L.orc_reach_add_edge(r, 1, 3, bits(0, 1))
    assert L.orc_reach_is_final_set_complete(r) == 0
    L.orc_reach_add_final_node(r, 2, bits(1, 1))
    assert L.orc_reach_is_final_set_complete(r) == 0
    L.orc_reach_add_final_node(r, 3, bits(1, 1))
    assert L.orc_reach_is_final_set_complete(r) == 1
    out = np.zeros(4, dtype=np.uint64)
    assert L.orc_reach_final_nodes_for_world(r, 0, out, 4) == 1 and out[0] == 2
    assert L.orc_reach_final_nodes_for_world(r, 1, out, 4) == 1 and out[0] == 3
    L.orc_reach_free(r)


def test_final_nodes_two_goals_two_worlds(oracle_lib):
    # pto_reachability.rs:198-230
    L, r = make_reach(oracle_lib, bits(1, 1), [bits(1, 1), bits(1, 1), bits(1, 1)])
    L.orc_reach_add_edge(r, 0, 1, bits(1, 1))
    L.orc_reach_add_edge(r, 1, 2, bits(1, 1))
    L.orc_reach_add_edge(r, 1, 3, bits(1, 1))
    L.orc_reach_add_final_node(r, 2, bits(1, 0))
    assert L.orc_reach_is_final_set_complete(r) == 0
    L.orc_reach_add_final_node(r, 3, bits(0, 1))
    assert L.orc_reach_is_final_set_complete(r) == 1
    out = np.zeros(4, dtype=np.uint64)
    assert L.orc_reach_final_nodes_for_world(r, 0, out, 4) == 1 and out[0] == 2
    assert L.orc_reach_final_nodes_for_world(r, 1, out, 4) == 1 and out[0] == 3
    L.orc_reach_free(r)


# ------------------------------------------------------------ SquareGoal
def test_square_goal(oracle_lib):
    # common.rs:401-411 test_goal, same inputs and expected values
    o = orc.Oracle()
    o.set_square_goal([[0.1, 0.1], [0.9, 0.9]], [bits(1, 0), bits(0, 1)], 0.1)
    assert o.goal([0.11, 0.11]) == bits(1, 0)
    assert o.goal([0.5, 0.5]) is None
    assert o.goal([0.91, 0.91]) == bits(0, 1)
    assert list(o.goal_example(0)) == [0.1, 0.1]
    assert list(o.goal_example(1)) == [0.9, 0.9]
    # the ball is L1 and open (common.rs:338: norm1(state, goal) < max_dist)
    assert o.goal([0.14, 0.15]) == bits(1, 0)
    assert o.goal([0.16, 0.15]) is None


def test_square_goal_overlap_rejected(oracle_lib):
    # common.rs:321: assert!(!world_has_goal[world])
    o = orc.Oracle()
    with pytest.raises(RuntimeError):
        o.set_square_goal([[1.0, 1.0], [2.0, 2.0]], [1, 1], 0.1)


# ----------------------------------------------------- geometry primitives
def test_norms_and_steer(oracle_lib):
    L = orc.lib()
    a, b = f64([0.0, 0.0]), f64([3.0, -4.0])
    assert L.orc_norm1(a, b) == 7.0 and L.orc_norm2(a, b) == 5.0
    to = f64([3.0, -4.0])
    L.orc_steer(a, to, 0.7)      # common.rs:215-225: the step is the L1 norm -> lambda = 0.1
    lam = 0.7 / 7.0
    assert to[0] == 0.0 + 3.0 * lam and to[1] == 0.0 + (-4.0) * lam
    to = f64([0.3, 0.3])
    L.orc_steer(a, to, 0.7)      # inside the step: untouched
    assert list(to) == [0.3, 0.3]


def test_heuristic_radius(oracle_lib):
    # common.rs:357-369 (the reference's own test only prints, common.rs:490-497)
    L = orc.lib()
    assert L.orc_heuristic_radius(1, 0.1, 2.0, 2) == 0.0
    for n in (2, 10, 3233, 5000, 100000):
        s = 2.0 * math.pow(math.log(float(n)) / float(n), 0.5)
        assert L.orc_heuristic_radius(n, 0.1, 2.0, 2) == (s if s < 0.1 else 0.1)
    assert L.orc_heuristic_radius(3232, 0.1, 2.0, 2) == 0.1
    assert L.orc_heuristic_radius(3233, 0.1, 2.0, 2) < 0.1


def test_rust_as_u32(oracle_lib):
    L = orc.lib()
    assert L.orc_f64_as_u32(-0.5) == 0 and L.orc_f64_as_u32(-7.0) == 0
    assert L.orc_f64_as_u32(float("nan")) == 0
    assert L.orc_f64_as_u32(199.99) == 199
    assert L.orc_f64_as_u32(1e20) == 2 ** 32 - 1


# --------------------------------------------------------- third-party KATs
def test_bresenham_documented_example(oracle_lib):
    # line_drawing's documented example: Bresenham::new((0, 0), (5, 6))
    L = orc.lib()
    out = np.zeros(64, dtype=np.int32)
    n = L.orc_bresenham(0, 0, 5, 6, out, 32)
    pts = [tuple(out[2 * k:2 * k + 2]) for k in range(n)]
    assert pts == [(0, 0), (0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6)]


def test_bresenham_properties(oracle_lib):
    # both end points inclusive, max(|dx|,|dy|)+1 points, unit steps, all octants
    L = orc.lib()
    out = np.zeros(256, dtype=np.int32)
    for (x0, y0, x1, y1) in itertools.product((-3, 0, 7), (-5, 0, 4), (-6, 0, 9), (-2, 0, 8)):
        n = L.orc_bresenham(x0, y0, x1, y1, out, 128)
        pts = out[:2 * n].reshape(-1, 2)
        assert n == max(abs(x1 - x0), abs(y1 - y0)) + 1
        assert tuple(pts[0]) == (x0, y0) and tuple(pts[-1]) == (x1, y1)
        steps = np.abs(np.diff(pts, axis=0))
        assert steps.max(initial=0) <= 1


def test_pcg64_published_vectors(oracle_lib):
    # PCG XSL-RR 128/64 reference vectors (pcg64 demo: state 42, stream 54) and the
    # rand_pcg from_seed([1..=32]) vector
    r = orc.Pcg64.new(42, 54)
    assert [r.next_u64() for _ in range(6)] == [0x86b1da1d72062b68, 0x1304aa46c9853d39, 0xa3670e9e0dd50358,
                                                0xf9090e529a7dae00, 0xc85b9fd837996f2c, 0x606121f8e3919196]
    r = orc.Pcg64.from_seed(range(1, 33))
    assert r.next_u64() == 8740028313290271629


def test_sampler_bounds(oracle_lib):
    # sample_space.rs:75-113: the only thing the reference pins is the bounds
    o = orc.Oracle()
    o.set_sampler((-1.0, -1.0), (1.0, 1.0), 0)
    for _ in range(100):
        s = o.sample()
        assert (-1.0 <= s).all() and (s < 1.0).all()
    for _ in range(100):
        assert o.sample_discrete(10) < 10


def test_gen_range_usize_matches_definition(oracle_lib):
    # widening-multiply rejection: zone = (n << lz(n)) - 1
    for n in (1, 2, 3, 10, 12, 16):
        r, q = orc.Pcg64.seed_from_u64(5), orc.Pcg64.seed_from_u64(5)
        zone = ((n << (64 - n.bit_length())) - 1) & (2 ** 64 - 1)
        for _ in range(50):
            while True:
                m = q.next_u64() * n
                if (m & (2 ** 64 - 1)) <= zone:
                    break
            assert r.gen_range_usize(0, n) == m >> 64
