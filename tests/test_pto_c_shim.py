"""libpo_rrt.so: the reference's own C symbols (src/pto_c.rs:63-270) on top of the engine (include/po_rrt_c.h).
CPU part: the library loads, exports every symbol the reference exports, keeps the reference's argument checks, and
refuses callbacks with an error code.  GPU part: plan() equals the same pipeline driven through the engine's C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_SYMBOLS = ["new_planning_problem", "delete_planning_problem", "set_problem_dimensions", "set_lower_sampling_bound",
                     "set_upper_sampling_bound", "set_world_validities", "set_state_validity_callback", "set_transition_validity_callback",
                     "set_cost_evaluator_callback", "set_observer_callback", "set_start_belief_state", "set_goal_callback",
                     "set_goal_example_callback", "set_search_parameters", "set_refine_parameters", "plan", "get_planning_metrics",
                     "get_paths_info", "get_paths_variable"]              # every #[no_mangle] pub extern "C" fn of pto_c.rs


def shim():
    from po_rrt_amd import build
    build.build()
    L = C.CDLL(os.path.join(ROOT, "po_rrt_amd", "libpo_rrt.so"))
    L.new_planning_problem.restype = C.c_void_p
    L.po_rrt_last_error.restype = C.c_char_p
    for name in REFERENCE_SYMBOLS + ["po_rrt_set_grid_domain", "po_rrt_set_square_goals", "po_rrt_set_seed", "po_rrt_set_device", "po_rrt_last_error"]:
        assert hasattr(L, name), name
    return L


def dbl(a):
    return (C.c_double * len(a))(*a)


def configure(L, case, belief, seed):
    p = C.c_void_p(L.new_planning_problem())
    n_worlds = len(belief)
    assert L.set_problem_dimensions(p, C.c_size_t(2), C.c_size_t(n_worlds)) == 0
    assert L.set_lower_sampling_bound(p, dbl([-1.0, -1.0]), C.c_size_t(2)) == 0
    assert L.set_upper_sampling_bound(p, dbl([1.0, 1.0]), C.c_size_t(2)) == 0
    occ, zones = cases.load_map(case.grid), cases.load_map(case.zones)
    h, w = occ.shape
    assert L.po_rrt_set_grid_domain(p, occ.ctypes.data_as(C.c_void_p), w, h, case.domain, zones.ctypes.data_as(C.c_void_p), C.c_double(case.visibility)) == 0
    g = np.ascontiguousarray(case.goals, dtype=np.float64)
    m = np.ascontiguousarray(case.masks, dtype=np.uint64)
    assert L.po_rrt_set_square_goals(p, g.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), len(m), C.c_double(case.l1)) == 0
    assert L.set_start_belief_state(p, dbl(belief), C.c_size_t(n_worlds), None, C.c_size_t(0)) == 0
    assert L.set_search_parameters(p, C.c_size_t(case.n_iter_min), C.c_size_t(case.n_iter_max), C.c_double(case.max_step), C.c_double(case.search_radius)) == 0
    assert L.set_refine_parameters(p, C.c_size_t(0)) == 0
    assert L.po_rrt_set_seed(p, C.c_uint64(seed), C.c_uint64(seed)) == 0
    return p


def test_symbols_checks_and_callbacks_without_a_gpu():
    L = shim()
    p = C.c_void_p(L.new_planning_problem())
    assert L.set_problem_dimensions(p, C.c_size_t(3), C.c_size_t(2)) == -1 and b"state_dim" in L.po_rrt_last_error(p)
    assert L.set_problem_dimensions(p, C.c_size_t(2), C.c_size_t(2)) == 0
    assert L.set_lower_sampling_bound(p, dbl([0.0] * 3), C.c_size_t(3)) == -1          # the reference: assert_eq!(state_dim, low_size)
    assert L.set_start_belief_state(p, dbl([1.0]), C.c_size_t(1), None, C.c_size_t(0)) == -1
    cb = C.CFUNCTYPE(C.c_int64, C.POINTER(C.c_double), C.c_size_t)(lambda s, n: 0)
    assert L.set_state_validity_callback(p, cb) == -1 and b"callbacks cannot run on the GPU" in L.po_rrt_last_error(p)
    assert L.plan(p, dbl([0.0, 0.0]), C.c_size_t(2)) == -1                                # and plan() says so again
    n, lens, cost = C.c_size_t(7), C.POINTER(C.c_size_t)(), C.c_double(-1)
    assert L.get_paths_info(p, C.byref(n), C.byref(lens), C.byref(cost)) == 0 and n.value == 0
    L.delete_planning_problem(p)
    L.delete_planning_problem(None)


@pytest.mark.gpu
def test_plan_equals_the_engine_pipeline():
    import po_rrt_amd
    L = shim()
    case = cases.cfg3_near(1500)
    case.update(n_iter_max=60000)                   # the reference's plan() panics unless the final set is complete (pto_c.rs:215)
    belief = [0.5, 0.5]
    p = configure(L, case, belief, seed=0)
    start = dbl(list(case.start))
    assert L.plan(p, start, C.c_size_t(2)) == 0, L.po_rrt_last_error(p)
    assert list(start) == list(case.start)                                                # plan() left the caller's buffer alone
    # the same through the engine's own ABI
    e = cases.configure(po_rrt_amd.Engine(), case)
    e.set_discrete_seed(0)
    cases.grow(e, case, K=256)
    e.build_belief_graph(belief)
    e.compute_expected_costs()
    (oid, par, leaf), cost = e.extract_policy()
    xy, _, _ = e.tree()
    B = len(e.belief_graph(lists=False)[0])
    n, lens, ecost = C.c_size_t(0), C.POINTER(C.c_size_t)(), C.c_double(0)
    assert L.get_paths_info(p, C.byref(n), C.byref(lens), C.byref(ecost)) == 0
    leaves = np.nonzero(leaf)[0]
    assert n.value == len(leaves) and ecost.value == cost and np.isfinite(cost)
    for i, k in enumerate(leaves):
        path = []
        while k >= 0:
            path.append(xy[oid[k] // B])
            k = par[k]
        path = path[::-1]
        assert lens[i] == len(path)
        for s, st in enumerate(path):
            ptr, size = C.POINTER(C.c_double)(), C.c_size_t(0)
            assert L.get_paths_variable(p, C.c_size_t(i), C.c_size_t(s), C.byref(ptr), C.byref(size)) == 0 and size.value == 2
            assert ptr[0] == st[0] and ptr[1] == st[1]
        assert tuple(path[0]) == tuple(case.start)
    it, g, b, d, r, t = C.c_size_t(0), C.c_double(0), C.c_double(0), C.c_double(0), C.c_double(-1), C.c_double(0)
    assert L.get_planning_metrics(p, C.byref(it), C.byref(g), C.byref(b), C.byref(d), C.byref(r), C.byref(t)) == 0
    assert it.value == e.num_iterations() and g.value > 0 and b.value > 0 and d.value > 0 and r.value == 0.0 and t.value >= g.value
    ptr, size = C.POINTER(C.c_double)(), C.c_size_t(0)
    assert L.get_paths_variable(p, C.c_size_t(99), C.c_size_t(0), C.byref(ptr), C.byref(size)) == -1
    # the reference's default: OS entropy -- two unseeded plans differ, both succeed or report an incomplete graph
    p2 = configure(L, case, belief, seed=0)
    L.delete_planning_problem(p)
    L.delete_planning_problem(p2)
