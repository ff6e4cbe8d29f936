"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg -- never by the product package po_rrt_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_RRT, MODE_PTO = 0, 1
ALGO_SEQ, ALGO_BATCHED, ALGO_BATCHED_KD = 0, 1, 2
DOMAIN_SHELF, DOMAIN_DOOR = 0, 1
FREE, LOW_OBSTACLE, HIGH_OBSTACLE, ZONE_BASE = 0, 1, 2, 16

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build():
    """(Re)build liboracle.so and orc_bench with the committed Makefile."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp = C.c_void_p

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_create", vp)
    sig("orc_destroy", None, vp)
    sig("orc_last_error", C.c_char_p, vp)
    sig("orc_set_grid", C.c_int, vp, _u8p, C.c_uint32, C.c_uint32, _f64p, _f64p, C.c_int)
    sig("orc_set_zones", C.c_int, vp, _u8p, C.c_double)
    sig("orc_set_sampler", C.c_int, vp, _f64p, _f64p, C.c_uint64)
    sig("orc_set_discrete_seed", C.c_int, vp, C.c_uint64)
    sig("orc_set_samples", C.c_int, vp, _f64p, C.c_size_t)
    sig("orc_set_worlds", C.c_int, vp, _u32p, C.c_size_t)
    sig("orc_set_square_goal", C.c_int, vp, _f64p, _u64p, C.c_uint32, C.c_double)
    sig("orc_set_observation_goal", C.c_int, vp, C.c_uint32)
    sig("orc_to_pixel", C.c_int, vp, _f64p, _u32p)
    sig("orc_state_class", C.c_int, vp, _f64p)
    sig("orc_prm_grow", C.c_int, vp, _f64p, C.c_double, C.c_double, C.c_uint64)
    sig("orc_prm_plan_path", C.c_int64, vp, _f64p, _f64p, _f64p, C.c_uint64)
    sig("orc_mm_prm_grow", vp, vp, _f64p, _f64p, C.c_double, C.c_double, C.c_uint64)
    sig("orc_mm_free", None, vp)
    for nm in ("modes", "transitions", "beliefs"):
        sig("orc_mm_num_" + nm, C.c_uint64, vp)
    sig("orc_mm_mode_info", None, vp, C.c_uint64, _f64p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
    sig("orc_mm_mode_graph", None, vp, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    sig("orc_mm_transition", None, vp, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_uint64))
    sig("orc_mm_transition_pairs", None, vp, C.c_uint64, C.c_void_p)
    sig("orc_belief_hash", C.c_uint64, _f64p, C.c_uint32)
    sig("orc_belief_successors", C.c_int, vp, _f64p, C.c_uint32, _f64p)
    sig("orc_zone_observable", C.c_int, vp, _f64p, C.c_uint32)
    sig("orc_observe", C.c_int64, vp, _f64p, _f64p, _f64p, C.c_size_t)
    sig("orc_reachable_beliefs", C.c_int64, vp, _f64p, C.c_void_p, C.c_size_t)
    sig("orc_build_belief_graph", C.c_int, vp, _f64p)
    sig("orc_conditional_dijkstra", C.c_int, C.c_uint64, _f64p, _u32p, _f64p, C.c_uint32, _u8p, _u64p, _u32p, _u64p, _u32p, _u64p, C.c_uint64, _f64p)
    sig("orc_extract_policy", C.c_int64, C.c_uint64, _f64p, _u32p, _u32p, _f64p, C.c_uint32, _u64p, _u32p, _f64p, _u64p, _i64p, _u8p, C.c_uint64)
    sig("orc_bg_expected_costs", C.c_int, vp, _f64p)
    sig("orc_bg_extract_policy", C.c_int64, vp, _f64p, _u64p, _i64p, _u8p, C.c_uint64)
    sig("orc_bg_num_beliefs", C.c_uint64, vp)
    sig("orc_bg_num_nodes", C.c_uint64, vp)
    sig("orc_bg_num_edges", C.c_uint64, vp)
    sig("orc_bg_get_beliefs", C.c_int, vp, _f64p)
    sig("orc_bg_get_types", C.c_int, vp, _u8p)
    sig("orc_bg_get_children", C.c_int, vp, _u64p, _u32p)
    sig("orc_bg_get_parents", C.c_int, vp, _u64p, _u32p)
    sig("orc_traversed_class", C.c_int, vp, _f64p, _f64p)
    sig("orc_n_zones", C.c_int, vp)
    sig("orc_n_worlds", C.c_int, vp)
    sig("orc_n_validities", C.c_int, vp)
    sig("orc_get_validities", C.c_int, vp, _u64p)
    sig("orc_get_zone_positions", C.c_int, vp, _f64p)
    sig("orc_goal", C.c_int, vp, _f64p, _u64p)
    sig("orc_goal_example", C.c_int, vp, C.c_uint32, _f64p)
    sig("orc_sample", C.c_int, vp, _f64p)
    sig("orc_sample_discrete", C.c_uint64, vp, C.c_uint64)
    sig("orc_grow", C.c_int, vp, _f64p, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int)
    sig("orc_num_nodes", C.c_uint64, vp)
    sig("orc_num_iterations", C.c_uint64, vp)
    sig("orc_get_tree", C.c_int, vp, _f64p, _i64p, _f64p)
    sig("orc_num_final", C.c_uint64, vp)
    sig("orc_get_final_ids", C.c_int, vp, _u64p)
    sig("orc_get_final_masks", C.c_int, vp, _u64p)
    sig("orc_get_reach", C.c_int, vp, _u64p)
    sig("orc_get_node_validity", C.c_int, vp, _u32p)
    sig("orc_num_edges", C.c_uint64, vp)
    sig("orc_get_edges", C.c_int, vp, _u32p, _u32p, _u32p)
    sig("orc_is_final_set_complete", C.c_int, vp)
    sig("orc_best_solution", C.c_uint64, vp, vp, C.c_uint64, C.POINTER(C.c_double))
    sig("orc_firstly_final_ids", C.c_uint64, vp, vp, C.c_uint64)
    # primitives
    sig("orc_norm1", C.c_double, _f64p, _f64p)
    sig("orc_norm2", C.c_double, _f64p, _f64p)
    sig("orc_steer", None, _f64p, _f64p, C.c_double)
    sig("orc_heuristic_radius", C.c_double, C.c_uint64, C.c_double, C.c_double, C.c_uint64)
    sig("orc_f64_as_u32", C.c_uint32, C.c_double)
    sig("orc_bresenham", C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, C.c_size_t)
    # pcg
    sig("orc_pcg64_new_u64", None, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64)
    sig("orc_pcg64_from_seed", None, vp, _u8p)
    sig("orc_pcg64_seed_from_u64", None, vp, C.c_uint64)
    sig("orc_pcg64_next_u64", C.c_uint64, vp)
    sig("orc_pcg64_get_state", None, vp, _u64p)
    sig("orc_gen_range_f64", C.c_double, vp, C.c_double, C.c_double)
    sig("orc_gen_range_usize", C.c_uint64, vp, C.c_uint64, C.c_uint64)
    # kd-tree
    sig("orc_kd_new", vp, _f64p, C.c_uint64)
    sig("orc_kd_free", None, vp)
    sig("orc_kd_add", None, vp, _f64p, C.c_uint64)
    sig("orc_kd_nearest", C.c_uint64, vp, _f64p, vp, C.c_uint32)
    sig("orc_kd_nearest_excluding", C.c_uint64, vp, _f64p, _u64p, C.c_size_t)
    sig("orc_kd_radius", C.c_size_t, vp, _f64p, C.c_double, _u64p, C.c_size_t)
    sig("orc_kd_child", C.c_int64, vp, C.c_uint64, C.c_int)
    sig("orc_kd_state", C.c_int, vp, C.c_uint64, _f64p)
    # reachability
    sig("orc_reach_new", vp)
    sig("orc_reach_free", None, vp)
    sig("orc_reach_set_root", None, vp, C.c_uint64, C.c_uint32)
    sig("orc_reach_add_node", None, vp, C.c_uint64)
    sig("orc_reach_add_final_node", None, vp, C.c_uint64, C.c_uint64)
    sig("orc_reach_add_edge", None, vp, C.c_uint64, C.c_uint64, C.c_uint64)
    sig("orc_reach_get", C.c_uint64, vp, C.c_uint64)
    sig("orc_reach_is_final_set_complete", C.c_int, vp)
    sig("orc_reach_final_nodes_for_world", C.c_size_t, vp, C.c_uint32, _u64p, C.c_size_t)
    _LIB = L
    return L


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Pcg64:
    """rand_pcg 0.3 Pcg64 (test helper)."""

    def __init__(self):
        self._buf = (C.c_uint8 * 32)()
        self._p = C.cast(self._buf, C.c_void_p)

    @classmethod
    def new(cls, state, stream):
        r = cls()
        m = (1 << 64) - 1
        lib().orc_pcg64_new_u64(r._p, state & m, state >> 64, stream & m, stream >> 64)
        return r

    @classmethod
    def from_seed(cls, seed_bytes):
        r = cls()
        lib().orc_pcg64_from_seed(r._p, np.frombuffer(bytes(seed_bytes), dtype=np.uint8).copy())
        return r

    @classmethod
    def seed_from_u64(cls, seed):
        r = cls()
        lib().orc_pcg64_seed_from_u64(r._p, seed)
        return r

    def next_u64(self):
        return lib().orc_pcg64_next_u64(self._p)

    def gen_range_f64(self, lo, hi):
        return lib().orc_gen_range_f64(self._p, lo, hi)

    def gen_range_usize(self, lo, hi):
        return lib().orc_gen_range_usize(self._p, lo, hi)

    def state(self):
        out = np.zeros(4, dtype=np.uint64)
        lib().orc_pcg64_get_state(self._p, out)
        return out


class Oracle:
    """One oracle context; mirrors the product's po_rrt_amd.Engine method names."""

    def __init__(self):
        self._l = lib()
        self._c = self._l.orc_create()

    def close(self):
        if self._c:
            self._l.orc_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise RuntimeError(self._l.orc_last_error(self._c).decode())
        return rc

    def set_grid(self, occ, low=(-1.0, -1.0), up=(1.0, 1.0), domain=DOMAIN_SHELF):
        occ = np.ascontiguousarray(occ, dtype=np.uint8)
        H, W = occ.shape
        self._chk(self._l.orc_set_grid(self._c, occ, W, H, _f64(low), _f64(up), domain))

    def set_zones(self, zone_ids, visibility):
        self._chk(self._l.orc_set_zones(self._c, np.ascontiguousarray(zone_ids, dtype=np.uint8), visibility))

    def set_sampler(self, low=(-1.0, -1.0), up=(1.0, 1.0), seed=0):
        self._chk(self._l.orc_set_sampler(self._c, _f64(low), _f64(up), seed))

    def set_discrete_seed(self, seed):
        self._chk(self._l.orc_set_discrete_seed(self._c, seed))

    def set_samples(self, xy):
        xy = _f64(xy).reshape(-1, 2)
        self._chk(self._l.orc_set_samples(self._c, xy, xy.shape[0]))

    def set_worlds(self, worlds):
        w = np.ascontiguousarray(worlds, dtype=np.uint32)
        self._chk(self._l.orc_set_worlds(self._c, w, w.size))

    def set_square_goal(self, centers, masks, l1_radius):
        centers = _f64(centers).reshape(-1, 2)
        masks = np.ascontiguousarray(masks, dtype=np.uint64)
        self._chk(self._l.orc_set_square_goal(self._c, centers, masks, centers.shape[0], l1_radius))

    def set_observation_goal(self, zone_id):
        self._chk(self._l.orc_set_observation_goal(self._c, zone_id))

    # domain probes
    def to_pixel(self, xy):
        ij = np.zeros(2, dtype=np.uint32)
        self._l.orc_to_pixel(self._c, _f64(xy), ij)
        return int(ij[0]), int(ij[1])

    def state_class(self, xy):
        return self._l.orc_state_class(self._c, _f64(xy))

    def traversed_class(self, a, b):
        return self._l.orc_traversed_class(self._c, _f64(a), _f64(b))

    def n_zones(self):
        return self._l.orc_n_zones(self._c)

    def n_worlds(self):
        return self._l.orc_n_worlds(self._c)

    def validities(self):
        out = np.zeros(65, dtype=np.uint64)
        n = self._l.orc_get_validities(self._c, out)
        return out[:n].copy()

    def zone_positions(self):
        out = np.zeros((64, 2), dtype=np.float64)
        n = self._l.orc_get_zone_positions(self._c, out)
        return out[:n].copy()

    def goal(self, xy):
        m = np.zeros(1, dtype=np.uint64)
        hit = self._l.orc_goal(self._c, _f64(xy), m)
        return int(m[0]) if hit else None

    def goal_example(self, world):
        xy = np.zeros(2)
        self._l.orc_goal_example(self._c, world, xy)
        return xy

    def sample(self):
        xy = np.zeros(2)
        self._chk(self._l.orc_sample(self._c, xy))
        return xy

    def sample_discrete(self, n):
        return self._l.orc_sample_discrete(self._c, n)

    def grow(self, start, max_step, search_radius, n_iter_min, n_iter_max, batch_K=1, mode=MODE_RRT, algo=ALGO_SEQ):
        return self._chk(self._l.orc_grow(self._c, _f64(start), max_step, search_radius, n_iter_min, n_iter_max,
                                          batch_K, mode, algo))

    def num_nodes(self):
        return self._l.orc_num_nodes(self._c)

    def num_iterations(self):
        return self._l.orc_num_iterations(self._c)

    def tree(self):
        n = self.num_nodes()
        xy = np.zeros((n, 2))
        parent = np.zeros(n, dtype=np.int64)
        dist = np.zeros(n)
        self._l.orc_get_tree(self._c, xy, parent, dist)
        return xy, parent, dist

    def final_ids(self):
        n = self._l.orc_num_final(self._c)
        ids = np.zeros(n, dtype=np.uint64)
        if n:
            self._l.orc_get_final_ids(self._c, ids)
        return ids

    def final_masks(self):
        n = self._l.orc_num_final(self._c)
        m = np.zeros(n, dtype=np.uint64)
        if n:
            self._l.orc_get_final_masks(self._c, m)
        return m

    def reach(self):
        m = np.zeros(self.num_nodes(), dtype=np.uint64)
        self._l.orc_get_reach(self._c, m)
        return m

    def node_validity(self):
        v = np.zeros(self.num_nodes(), dtype=np.uint32)
        self._l.orc_get_node_validity(self._c, v)
        return v

    def edges(self):
        n = self._l.orc_num_edges(self._c)
        f = np.zeros(n, dtype=np.uint32)
        t = np.zeros(n, dtype=np.uint32)
        v = np.zeros(n, dtype=np.uint32)
        if n:
            self._l.orc_get_edges(self._c, f, t, v)
        return f, t, v

    def grow_prm(self, start, max_step, search_radius, n_iter):
        """PRM::init + PRM::grow_graph (prm.rs:33-109)"""
        return self._chk(self._l.orc_prm_grow(self._c, _f64(start), max_step, search_radius, n_iter))

    def grow_mm_prm(self, start, belief, max_step, search_radius, n_iter_per_belief):
        """MapShelfDomainTampPRM::grow_mm_prm (map_shelves_tamp_prm.rs:328-393): dict(n_beliefs, modes=[dict(belief, reaching_probability,
        xy, edges=(from, to), finals)], transitions=[dict(zone, from_mode, to_mode, observation, pairs)])"""
        L = self._l
        g = L.orc_mm_prm_grow(self._c, _f64(start), _f64(belief), max_step, search_radius, n_iter_per_belief)
        if not g:
            raise RuntimeError(L.orc_last_error(self._c).decode())
        try:
            nw = self.n_worlds()
            modes, trs = [], []
            for m in range(L.orc_mm_num_modes(g)):
                b, rp = np.zeros(nw), C.c_double(0)
                nn, ne, nf = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
                L.orc_mm_mode_info(g, m, b, C.byref(rp), C.byref(nn), C.byref(ne), C.byref(nf))
                xy = np.zeros((nn.value, 2))
                ef, et, fin = np.zeros(ne.value, dtype=np.uint64), np.zeros(ne.value, dtype=np.uint64), np.zeros(nf.value, dtype=np.uint64)
                p = lambda a: a.ctypes.data_as(C.c_void_p)
                L.orc_mm_mode_graph(g, m, p(xy), p(ef), p(et), p(fin))
                modes.append(dict(belief=b, reaching_probability=rp.value, xy=xy, edges=(ef, et), finals=fin))
            for t in range(L.orc_mm_num_transitions(g)):
                z, f, to, ob, n = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_int(0), C.c_uint64(0)
                L.orc_mm_transition(g, t, C.byref(z), C.byref(f), C.byref(to), C.byref(ob), C.byref(n))
                pairs = np.zeros((n.value, 2), dtype=np.uint64)
                L.orc_mm_transition_pairs(g, t, pairs.ctypes.data_as(C.c_void_p))
                trs.append(dict(zone=z.value, from_mode=f.value, to_mode=to.value, observation=ob.value, pairs=pairs))
            return dict(n_beliefs=L.orc_mm_num_beliefs(g), modes=modes, transitions=trs)
        finally:
            L.orc_mm_free(g)

    def prm_plan_path(self, start, goal):
        """PRM::plan_path (prm.rs:111-123): list of states, empty when the goal is not connected"""
        cap = self.num_nodes() + 1
        out = np.zeros((cap, 2))
        n = self._l.orc_prm_plan_path(self._c, _f64(start), _f64(goal), out, cap)
        if n < 0:
            raise RuntimeError("plan_path failed")
        return out[:n]

    # ---- belief space (belief.c; pto.rs:185-259)
    def belief_hash(self, b):
        b = _f64(b)
        return self._l.orc_belief_hash(b, len(b))

    def belief_successors(self, belief, zone):
        nw = self.n_worlds()
        out = np.zeros((2, nw))
        k = self._l.orc_belief_successors(self._c, _f64(belief), zone, out)
        return out[:k]

    def zone_observable(self, xy, zone):
        return self._l.orc_zone_observable(self._c, _f64(xy), zone)

    def observe(self, xy, belief, cap=4096):
        nw = self.n_worlds()
        out = np.zeros((cap, nw))
        k = self._l.orc_observe(self._c, _f64(xy), _f64(belief), out, cap)
        if k < 0:
            raise RuntimeError("observe failed (%d)" % k)
        return out[:k]

    def reachable_beliefs(self, start):
        nw = self.n_worlds()
        n = self._l.orc_reachable_beliefs(self._c, _f64(start), None, 0)
        out = np.zeros((n, nw))
        self._l.orc_reachable_beliefs(self._c, _f64(start), out.ctypes.data_as(C.c_void_p), n)
        return out

    def build_belief_graph(self, start_belief):
        self._chk(self._l.orc_build_belief_graph(self._c, _f64(start_belief)))

    def belief_graph(self):
        """beliefs [B, n_worlds], node types [N*B], (child_off, child_ids), (parent_off, parent_ids)"""
        B, NB, E = (self._l.orc_bg_num_beliefs(self._c), self._l.orc_bg_num_nodes(self._c), self._l.orc_bg_num_edges(self._c))
        beliefs = np.zeros((B, self.n_worlds()))
        types = np.zeros(NB, dtype=np.uint8)
        coff, poff = np.zeros(NB + 1, dtype=np.uint64), np.zeros(NB + 1, dtype=np.uint64)
        cid, pid = np.zeros(max(E, 1), dtype=np.uint32), np.zeros(max(E, 1), dtype=np.uint32)
        self._chk(self._l.orc_bg_get_beliefs(self._c, beliefs))
        self._chk(self._l.orc_bg_get_types(self._c, types))
        self._chk(self._l.orc_bg_get_children(self._c, coff, cid))
        self._chk(self._l.orc_bg_get_parents(self._c, poff, pid))
        return beliefs, types, (coff, cid[:E]), (poff, pid[:E])

    def expected_costs(self):
        """PTO::compute_expected_costs_to_goals on the last belief graph"""
        d = np.zeros(self._l.orc_bg_num_nodes(self._c))
        self._chk(self._l.orc_bg_expected_costs(self._c, d))
        return d

    def extract_policy(self, dist, cap=1 << 20):
        """PTO::extract_policy: (original belief node id, parent policy node or -1, is_leaf) per policy node"""
        oid, par, leaf = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.uint8)
        n = self._l.orc_bg_extract_policy(self._c, np.ascontiguousarray(dist, dtype=np.float64), oid, par, leaf, cap)
        if n < 0:
            raise RuntimeError("extract_policy failed (%d)" % n)
        return oid[:n], par[:n], leaf[:n]

    def is_final_set_complete(self):
        return bool(self._l.orc_is_final_set_complete(self._c))

    def best_solution(self):
        cost = C.c_double(0.0)
        n = self._l.orc_best_solution(self._c, None, 0, C.byref(cost))
        if n == 0:
            return None
        path = np.zeros((n, 2))
        self._l.orc_best_solution(self._c, path.ctypes.data_as(C.c_void_p), n, C.byref(cost))
        return path, cost.value

    def firstly_final_ids(self):
        n = self._l.orc_firstly_final_ids(self._c, None, 0)
        ids = np.zeros(n, dtype=np.uint64)
        if n:
            self._l.orc_firstly_final_ids(self._c, ids.ctypes.data_as(C.c_void_p), n)
        return ids


def conditional_dijkstra(xy, belief_vec, beliefs, types, children, parents, finals):
    """conditional_dijkstra (belief_graph.rs:89-175) on an explicit graph; children / parents are lists of lists;
    belief_vec[i] = row of `beliefs` that node i carries"""
    L = lib()
    n = len(types)

    def csr(lists):
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in lists])
        ids = np.array([v for x in lists for v in x] + [0], dtype=np.uint32)
        return off, ids
    coff, cid = csr(children)
    poff, pid = csr(parents)
    beliefs = np.ascontiguousarray(beliefs, dtype=np.float64)
    dist = np.zeros(n)
    rc = L.orc_conditional_dijkstra(n, np.ascontiguousarray(xy, dtype=np.float64), np.ascontiguousarray(belief_vec, dtype=np.uint32), beliefs,
                                    beliefs.shape[1], np.ascontiguousarray(types, dtype=np.uint8), coff, cid, poff, pid,
                                    np.ascontiguousarray(finals, dtype=np.uint64), len(finals), dist)
    if rc:
        raise RuntimeError("conditional_dijkstra failed (%d)" % rc)
    return dist, (coff, cid), (poff, pid)


def extract_policy(xy, belief_id, belief_vec, beliefs, children_csr, dist, cap=4096):
    L = lib()
    n = len(dist)
    beliefs = np.ascontiguousarray(beliefs, dtype=np.float64)
    oid, par, leaf = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.int64), np.zeros(cap, dtype=np.uint8)
    k = L.orc_extract_policy(n, np.ascontiguousarray(xy, dtype=np.float64), np.ascontiguousarray(belief_id, dtype=np.uint32),
                             np.ascontiguousarray(belief_vec, dtype=np.uint32), beliefs,
                             beliefs.shape[1], children_csr[0], children_csr[1], np.ascontiguousarray(dist, dtype=np.float64), oid, par, leaf, cap)
    if k < 0:
        raise RuntimeError("extract_policy failed (%d)" % k)
    return oid[:k], par[:k], leaf[:k]
