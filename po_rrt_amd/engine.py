"""ctypes binding of libporrt_hip.so -- the C ABI declared in include/porrt_hip.h.

This is plumbing only: every method forwards to one exported symbol.  There is no CPU
fallback; constructing an Engine without the HIP library or without a GPU raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libporrt_hip.so")

MODE_RRT, MODE_PTO = 0, 1
DOMAIN_SHELF, DOMAIN_DOOR = 0, 1
OK, INCOMPLETE = 0, 1

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

# every symbol include/porrt_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "porrt_create", "porrt_destroy", "porrt_last_error", "porrt_set_grid", "porrt_set_zones", "porrt_set_sampler",
    "porrt_set_discrete_seed", "porrt_set_samples", "porrt_set_worlds", "porrt_set_square_goal",
    "porrt_set_observation_goal", "porrt_grow", "porrt_grow_batch", "porrt_grow_batch_each", "porrt_grow_prm", "porrt_prm_plan_path", "porrt_num_nodes", "porrt_num_iterations", "porrt_get_tree", "porrt_get_trees",
    "porrt_num_final", "porrt_get_final_ids", "porrt_get_final_masks", "porrt_get_reach", "porrt_get_node_validity",
    "porrt_num_edges", "porrt_get_edges", "porrt_is_final_set_complete", "porrt_n_worlds", "porrt_get_validities",
    "porrt_get_zone_positions", "porrt_best_solution", "porrt_best_cost", "porrt_best_cost_batch", "porrt_get_metrics", "porrt_set_option", "porrt_get_option", "porrt_selftest",
    "porrt_build_belief_graph", "porrt_bg_num_beliefs", "porrt_bg_num_nodes", "porrt_bg_num_edges", "porrt_bg_get_beliefs",
    "porrt_bg_get_observable_zones", "porrt_bg_get_node_types", "porrt_bg_get_children", "porrt_bg_get_parents", "porrt_bg_get_seconds",
    "porrt_bg_compute_expected_costs", "porrt_bg_get_expected_costs", "porrt_bg_expected_cost_of", "porrt_bg_get_dp_info", "porrt_bg_get_dp_sweep_rows", "porrt_bg_extract_policy", "porrt_conditional_dijkstra",
    "porrt_comm_unique_id", "porrt_comm_create", "porrt_comm_destroy", "porrt_comm_last_error", "porrt_exchange_best", "porrt_exchange_num_nodes",
    "porrt_exchange_get_tree", "porrt_exchange_decide", "porrt_exchange_agree", "porrt_tree_device",
    "porrt_host_pin", "porrt_host_unpin", "porrt_exchange_tables", "porrt_comm_test_new_ops", "porrt_comm_usable", "porrt_comm_set_timeout_ms", "porrt_comm_test_new", "porrt_comm_test_fail", "porrt_comm_test_aborts",
    "porrt_grow_mm_prm", "porrt_mm_num_modes", "porrt_mm_num_transitions", "porrt_mm_num_beliefs", "porrt_mm_get_mode", "porrt_mm_get_mode_graph",
    "porrt_mm_get_transition", "porrt_mm_get_transition_pairs", "porrt_mm_get_seconds",
    "porrt_read_pgm", "porrt_read_pgm_mem", "porrt_graph_write_json", "porrt_graph_save_json", "porrt_graph_load_json", "porrt_graph_file_free",
    "porrt_graph_file_num_nodes", "porrt_graph_file_num_children", "porrt_graph_file_num_parents", "porrt_graph_file_num_validities",
    "porrt_graph_file_num_worlds", "porrt_graph_file_get",
]


class Metrics(C.Structure):
    _fields_ = [("n_iter", C.c_uint64), ("n_nodes", C.c_uint64), ("n_steps", C.c_uint64),
                ("n_tie_fallbacks", C.c_uint64), ("total_s", C.c_double), ("setup_s", C.c_double),
                ("device_s", C.c_double), ("scan_s", C.c_double), ("scan_launches", C.c_uint64),
                ("scan_pairs", C.c_double), ("scan_bytes", C.c_double), ("connect_s", C.c_double)]


class TreeDeviceView(C.Structure):
    _fields_ = [("nx", C.c_void_p), ("ny", C.c_void_p), ("dist_root", C.c_void_p), ("parent", C.c_void_p), ("n_nodes", C.c_uint64)]


BEST_ENTRY = np.dtype([("cost", np.float64), ("rank", np.int32), ("n_nodes", np.int32)])      # porrt_best_entry, 16 bytes
UNIQUE_ID_BYTES = 128


class PorrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("porrt error %d: %s" % (code, msg))
        self.code = code


_LIB = None


def load_library():
    """dlopen the in-tree HIP library; fails loudly when it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("porrt_create", vp, C.c_int)
    sig("porrt_destroy", None, vp)
    sig("porrt_last_error", C.c_char_p, vp)
    sig("porrt_set_grid", C.c_int, vp, _u8p, C.c_uint32, C.c_uint32, _f64p, _f64p, C.c_int)
    sig("porrt_set_zones", C.c_int, vp, _u8p, C.c_double)
    sig("porrt_set_sampler", C.c_int, vp, _f64p, _f64p, C.c_uint64)
    sig("porrt_set_discrete_seed", C.c_int, vp, C.c_uint64)
    sig("porrt_set_samples", C.c_int, vp, _f64p, C.c_size_t)
    sig("porrt_set_worlds", C.c_int, vp, _u32p, C.c_size_t)
    sig("porrt_set_square_goal", C.c_int, vp, _f64p, _u64p, C.c_uint32, C.c_double)
    sig("porrt_set_observation_goal", C.c_int, vp, C.c_uint32)
    sig("porrt_grow", C.c_int, vp, _f64p, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int)
    sig("porrt_grow_batch", C.c_int, C.POINTER(C.c_void_p), C.c_uint32, _f64p, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int)
    sig("porrt_grow_batch_each", C.c_int, C.POINTER(C.c_void_p), C.c_uint32, _f64p, C.c_double, C.c_double, _u64p, _u64p, C.c_uint32, C.c_int)
    sig("porrt_num_nodes", C.c_uint64, vp)
    sig("porrt_num_iterations", C.c_uint64, vp)
    sig("porrt_get_tree", C.c_int, vp, _f64p, _i64p, _f64p)
    sig("porrt_get_trees", C.c_int, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
    sig("porrt_num_final", C.c_uint64, vp)
    sig("porrt_get_final_ids", C.c_int, vp, _u64p)
    sig("porrt_get_final_masks", C.c_int, vp, _u64p)
    sig("porrt_get_reach", C.c_int, vp, _u64p)
    sig("porrt_get_node_validity", C.c_int, vp, _u32p)
    sig("porrt_num_edges", C.c_uint64, vp)
    sig("porrt_get_edges", C.c_int, vp, _u32p, _u32p, _u32p)
    sig("porrt_is_final_set_complete", C.c_int, vp)
    sig("porrt_n_worlds", C.c_int, vp)
    sig("porrt_get_validities", C.c_int, vp, _u64p)
    sig("porrt_get_zone_positions", C.c_int, vp, _f64p)
    sig("porrt_best_solution", C.c_uint64, vp, vp, C.c_uint64, C.POINTER(C.c_double))
    sig("porrt_best_cost", C.c_int, vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64))
    sig("porrt_grow_prm", C.c_int, vp, _f64p, C.c_double, C.c_double, C.c_uint64)
    sig("porrt_prm_plan_path", C.c_int64, vp, _f64p, _f64p, C.c_void_p, C.c_uint64)
    sig("porrt_build_belief_graph", C.c_int, vp, _f64p, C.c_uint32)
    sig("porrt_bg_num_beliefs", C.c_uint64, vp)
    sig("porrt_bg_num_nodes", C.c_uint64, vp)
    sig("porrt_bg_num_edges", C.c_uint64, vp)
    sig("porrt_bg_get_beliefs", C.c_int, vp, _f64p)
    sig("porrt_bg_get_observable_zones", C.c_int, vp, _u64p)
    sig("porrt_bg_get_node_types", C.c_int, vp, _u8p)
    sig("porrt_bg_get_children", C.c_int, vp, _u64p, C.c_void_p)
    sig("porrt_bg_get_parents", C.c_int, vp, _u64p, C.c_void_p)
    sig("porrt_bg_get_seconds", C.c_int, vp, _f64p, C.c_uint32)
    sig("porrt_bg_compute_expected_costs", C.c_int, vp)
    sig("porrt_bg_get_expected_costs", C.c_int, vp, _f64p)
    sig("porrt_bg_expected_cost_of", C.c_int, vp, C.c_uint64, C.POINTER(C.c_double))
    sig("porrt_bg_get_dp_info", C.c_int, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint32))
    sig("porrt_bg_get_dp_sweep_rows", C.c_uint64, vp)
    sig("porrt_bg_extract_policy", C.c_int64, vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_double))
    sig("porrt_conditional_dijkstra", C.c_int, C.c_int, C.c_uint64, _f64p, _u32p, _f64p, C.c_uint32, C.c_uint32, _u8p, _u64p, _u32p, _u64p, _u32p,
        _u64p, C.c_uint64, _f64p)
    sig("porrt_best_cost_batch", C.c_int, C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_double))
    sig("porrt_get_metrics", C.c_int, vp, C.POINTER(Metrics))
    sig("porrt_set_option", C.c_int, vp, C.c_char_p, C.c_int64)
    sig("porrt_get_option", C.c_int, vp, C.c_char_p, C.POINTER(C.c_int64))
    sig("porrt_selftest", C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
    sig("porrt_comm_unique_id", C.c_int, C.c_void_p)
    sig("porrt_comm_create", vp, C.c_int, C.c_int, C.c_int, C.c_void_p)
    sig("porrt_comm_destroy", None, vp)
    sig("porrt_comm_last_error", C.c_char_p, vp)
    sig("porrt_exchange_best", C.c_int, vp, C.POINTER(C.c_void_p), C.c_uint32, _u32p, C.c_uint32, C.c_void_p)
    sig("porrt_exchange_num_nodes", C.c_uint64, vp, C.c_uint32)
    sig("porrt_exchange_get_tree", C.c_int, vp, C.c_uint32, _f64p, _i64p, _f64p)
    sig("porrt_exchange_decide", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p)
    sig("porrt_exchange_agree", C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32))
    sig("porrt_tree_device", TreeDeviceView, vp)
    sig("porrt_host_pin", C.c_int, vp, C.c_size_t)
    sig("porrt_host_unpin", C.c_int, vp)
    sig("porrt_exchange_tables", C.c_int, vp, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p)
    sig("porrt_comm_test_new_ops", vp, C.c_int, C.c_int, C.c_void_p)
    sig("porrt_comm_usable", C.c_int, vp)
    sig("porrt_comm_set_timeout_ms", C.c_int, vp, C.c_int)
    sig("porrt_comm_test_new", vp, C.c_int, C.c_int)
    sig("porrt_comm_test_fail", C.c_int, vp, C.c_int, C.c_int)
    sig("porrt_comm_test_aborts", C.c_int, vp)
    sig("porrt_grow_mm_prm", C.c_int, vp, _f64p, _f64p, C.c_uint32, C.c_double, C.c_double, C.c_uint64)
    for nm in ("modes", "transitions", "beliefs"):
        sig("porrt_mm_num_" + nm, C.c_uint64, vp)
    sig("porrt_mm_get_mode", C.c_int, vp, C.c_uint64, _f64p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))
    sig("porrt_mm_get_mode_graph", C.c_int, vp, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    sig("porrt_mm_get_transition", C.c_int, vp, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_uint64))
    sig("porrt_mm_get_transition_pairs", C.c_int, vp, C.c_uint64, C.c_void_p)
    sig("porrt_mm_get_seconds", C.c_int, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
    sig("porrt_read_pgm", C.c_int, C.c_char_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
    sig("porrt_read_pgm_mem", C.c_int, C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
    sig("porrt_graph_write_json", C.c_int, C.c_char_p, C.c_uint64, _f64p, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p, C.c_uint64, C.c_uint64, _u8p)
    sig("porrt_graph_save_json", C.c_int, vp, C.c_char_p)
    sig("porrt_graph_load_json", vp, C.c_char_p, C.c_char_p, C.c_size_t)
    sig("porrt_graph_file_free", None, vp)
    for nm in ("nodes", "children", "parents", "validities", "worlds"):
        sig("porrt_graph_file_num_" + nm, C.c_uint64, vp)
    sig("porrt_graph_file_get", C.c_int, vp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    _LIB = L
    return L


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Engine:
    """One porrt_ctx on one GPU (mirrors the `&mut self` RRT / PTO object of the reference)."""

    def __init__(self, device=0):
        self._l = load_library()
        self._c = self._l.porrt_create(device)
        if not self._c:
            raise PorrtError(-6, "porrt_create failed: no usable HIP device %d" % device)

    def close(self):
        if getattr(self, "_c", None):
            self._l.porrt_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise PorrtError(rc, self._l.porrt_last_error(self._c).decode())
        return rc

    def set_option(self, name, value):
        self._chk(self._l.porrt_set_option(self._c, name.encode(), int(value)))

    def set_grid(self, occ, low=(-1.0, -1.0), up=(1.0, 1.0), domain=DOMAIN_SHELF):
        occ = np.ascontiguousarray(occ, dtype=np.uint8)
        H, W = occ.shape
        self._chk(self._l.porrt_set_grid(self._c, occ, W, H, _f64(low), _f64(up), domain))

    def set_zones(self, zone_ids, visibility):
        self._chk(self._l.porrt_set_zones(self._c, np.ascontiguousarray(zone_ids, dtype=np.uint8), visibility))

    def set_sampler(self, low=(-1.0, -1.0), up=(1.0, 1.0), seed=0):
        self._chk(self._l.porrt_set_sampler(self._c, _f64(low), _f64(up), seed))

    def set_discrete_seed(self, seed):
        self._chk(self._l.porrt_set_discrete_seed(self._c, seed))

    def set_samples(self, xy):
        xy = _f64(xy).reshape(-1, 2)
        self._chk(self._l.porrt_set_samples(self._c, xy, xy.shape[0]))

    def set_worlds(self, worlds):
        w = np.ascontiguousarray(worlds, dtype=np.uint32)
        self._chk(self._l.porrt_set_worlds(self._c, w, w.size))

    def set_square_goal(self, centers, masks, l1_radius):
        centers = _f64(centers).reshape(-1, 2)
        masks = np.ascontiguousarray(masks, dtype=np.uint64)
        self._chk(self._l.porrt_set_square_goal(self._c, centers, masks, centers.shape[0], l1_radius))

    def set_observation_goal(self, zone_id):
        self._chk(self._l.porrt_set_observation_goal(self._c, zone_id))

    def n_worlds(self):
        return self._l.porrt_n_worlds(self._c)

    def validities(self):
        out = np.zeros(65, dtype=np.uint64)
        n = self._chk(self._l.porrt_get_validities(self._c, out))
        return out[:n].copy()

    def zone_positions(self):
        out = np.zeros((64, 2), dtype=np.float64)
        n = self._chk(self._l.porrt_get_zone_positions(self._c, out))
        return out[:n].copy()

    def grow(self, start, max_step, search_radius, n_iter_min, n_iter_max, batch_K=1024, mode=MODE_RRT):
        """RRT::grow_tree (rrt.rs:102-174) / PTO::grow_graph (pto.rs:55-139) on the GPU."""
        return self._chk(self._l.porrt_grow(self._c, _f64(start), max_step, search_radius, n_iter_min, n_iter_max,
                                            batch_K, mode))

    @staticmethod
    def grow_batch(engines, starts, max_step, search_radius, n_iter_min, batch_K=1024, mode=MODE_RRT, n_iter_max=None):
        """porrt_grow_batch / porrt_grow_batch_each: the same growth for several contexts of one GPU in one launch sequence, each
        running the reference's loop `while i < n_iter_min || (no solution && i < n_iter_max)`.  n_iter_min / n_iter_max: one
        number for all, or one per engine; n_iter_max = None: a fixed budget (= n_iter_min)."""
        engines = list(engines)
        arr = (C.c_void_p * len(engines))(*[e._c for e in engines])
        st = np.ascontiguousarray(np.asarray(starts, dtype=np.float64).reshape(len(engines), 2))
        if n_iter_max is None:
            n_iter_max = n_iter_min
        if np.ndim(n_iter_min) == 0 and np.ndim(n_iter_max) == 0:
            return engines[0]._chk(engines[0]._l.porrt_grow_batch(arr, len(engines), st.reshape(-1), max_step, search_radius,
                                                                   int(n_iter_min), int(n_iter_max), batch_K, mode))
        mn = np.ascontiguousarray(np.broadcast_to(np.asarray(n_iter_min, dtype=np.uint64), (len(engines),)))
        mx = np.ascontiguousarray(np.broadcast_to(np.asarray(n_iter_max, dtype=np.uint64), (len(engines),)))
        return engines[0]._chk(engines[0]._l.porrt_grow_batch_each(arr, len(engines), st.reshape(-1), max_step, search_radius, mn, mx, batch_K, mode))

    def num_nodes(self):
        return self._l.porrt_num_nodes(self._c)

    def num_iterations(self):
        return self._l.porrt_num_iterations(self._c)

    def num_final(self):
        return self._l.porrt_num_final(self._c)

    def get_option(self, name):
        v = C.c_int64(0)
        self._chk(self._l.porrt_get_option(self._c, name.encode(), C.byref(v)))
        return v.value

    def tree(self):
        n = self.num_nodes()
        xy = np.zeros((n, 2))
        parent = np.zeros(n, dtype=np.int64)
        dist = np.zeros(n)
        self._chk(self._l.porrt_get_tree(self._c, xy, parent, dist))
        return xy, parent, dist

    @staticmethod
    def trees(engines, buffers=None):
        """porrt_get_trees: the trees of several contexts of one device in one call -> [(xy, parent, dist_root), ...].
        buffers: per engine (xy [cap, 2] f64, parent [cap] i64, dist [cap] f64) to fill instead of new arrays (a caller that
        fetches trees again and again keeps its memory mapped: fresh pages cost more than the copies); views of them come back."""
        n = len(engines)
        sizes = [e.num_nodes() for e in engines]
        if buffers is None:
            out = [(np.empty((m, 2)), np.empty(m, dtype=np.int64), np.empty(m)) for m in sizes]
        else:
            out = []
            for m, (bxy, bp, bd) in zip(sizes, buffers):
                if len(bxy) < m or len(bp) < m or len(bd) < m or bxy.dtype != np.float64 or bp.dtype != np.int64 or bd.dtype != np.float64:
                    raise ValueError("trees: a buffer is too small or of the wrong type")
                out.append((bxy[:m], bp[:m], bd[:m]))
        arr = (C.c_void_p * n)(*[e._c for e in engines])
        ptrs = [(C.c_void_p * n)(*[o[k].ctypes.data for o in out]) for k in range(3)]
        rc = engines[0]._l.porrt_get_trees(arr, n, ptrs[0], ptrs[1], ptrs[2])
        engines[0]._chk(rc)
        return out

    @staticmethod
    def pin_buffers(buffers):
        """porrt_host_pin for the arrays of `buffers` (as Engine.trees takes them): a later trees(engines, buffers) lets one kernel write
        every tree straight into them.  The arrays must stay alive until unpin_buffers(buffers)."""
        L = load_library()
        for arrs in buffers:
            for a in arrs:
                rc = L.porrt_host_pin(a.ctypes.data, a.nbytes)
                if rc:
                    raise PorrtError(rc, "porrt_host_pin (%d bytes)" % a.nbytes)

    @staticmethod
    def unpin_buffers(buffers):
        L = load_library()
        for arrs in buffers:
            for a in arrs:
                L.porrt_host_unpin(a.ctypes.data)

    def final_ids(self):
        n = self._l.porrt_num_final(self._c)
        ids = np.zeros(n, dtype=np.uint64)
        if n:
            self._chk(self._l.porrt_get_final_ids(self._c, ids))
        return ids

    def final_masks(self):
        n = self._l.porrt_num_final(self._c)
        m = np.zeros(n, dtype=np.uint64)
        if n:
            self._chk(self._l.porrt_get_final_masks(self._c, m))
        return m

    def reach(self):
        m = np.zeros(self.num_nodes(), dtype=np.uint64)
        self._chk(self._l.porrt_get_reach(self._c, m))
        return m

    def node_validity(self):
        v = np.zeros(self.num_nodes(), dtype=np.uint32)
        self._chk(self._l.porrt_get_node_validity(self._c, v))
        return v

    def edges(self):
        n = self._l.porrt_num_edges(self._c)
        f = np.zeros(n, dtype=np.uint32)
        t = np.zeros(n, dtype=np.uint32)
        v = np.zeros(n, dtype=np.uint32)
        if n:
            self._chk(self._l.porrt_get_edges(self._c, f, t, v))
        return f, t, v

    def is_final_set_complete(self):
        return bool(self._l.porrt_is_final_set_complete(self._c))

    def best_solution(self):
        cost = C.c_double(0.0)
        n = self._l.porrt_best_solution(self._c, None, 0, C.byref(cost))
        if n == 0:
            return None
        path = np.zeros((n, 2))
        self._l.porrt_best_solution(self._c, path.ctypes.data_as(C.c_void_p), n, C.byref(cost))
        return path, cost.value

    def best_cost(self):
        """cost of best_solution()'s path, evaluated on the device (no tree download); None = no solution"""
        cost, fid = C.c_double(0.0), C.c_uint64(0)
        r = self._chk(self._l.porrt_best_cost(self._c, C.byref(cost), C.byref(fid)))
        return cost.value if r else None

    @staticmethod
    def best_cost_batch(engines):
        """best_cost() of every engine (inf = no solution); one launch when they are the last grow_batch's members"""
        n = len(engines)
        arr = (C.c_void_p * n)(*[e._c for e in engines])
        costs = np.zeros(n)
        rc = engines[0]._l.porrt_best_cost_batch(arr, n, costs.ctypes.data_as(C.POINTER(C.c_double)))
        if rc < 0:
            engines[0]._chk(rc)
        return costs

    def grow_prm(self, start, max_step, search_radius, n_iter):
        """PRM::init + PRM::grow_graph (prm.rs:33-109); nodes and edges through tree() / edges()"""
        return self._chk(self._l.porrt_grow_prm(self._c, np.ascontiguousarray(start, dtype=np.float64), max_step, search_radius, n_iter))

    def prm_plan_path(self, start, goal):
        """PRM::plan_path (prm.rs:111-123): array of states, empty when start and goal are not connected"""
        a, b = np.ascontiguousarray(start, dtype=np.float64), np.ascontiguousarray(goal, dtype=np.float64)
        n = self._l.porrt_prm_plan_path(self._c, a, b, None, 0)
        if n < 0:
            self._chk(int(n))
        out = np.zeros((n, 2))
        if n:
            self._l.porrt_prm_plan_path(self._c, a, b, out.ctypes.data_as(C.c_void_p), n)
        return out

    # ---- belief-space expansion (PTO::build_belief_graph, pto.rs:185-259)
    def build_belief_graph(self, start_belief):
        b = np.ascontiguousarray(start_belief, dtype=np.float64)
        self._chk(self._l.porrt_build_belief_graph(self._c, b, len(b)))

    def belief_graph(self, lists=True):
        """beliefs [B, n_worlds], node types [N*B], (child_off, child_ids), (parent_off, parent_ids)"""
        B, NB, E = (self._l.porrt_bg_num_beliefs(self._c), self._l.porrt_bg_num_nodes(self._c), self._l.porrt_bg_num_edges(self._c))
        beliefs = np.zeros((B, self.n_worlds()))
        types = np.zeros(NB, dtype=np.uint8)
        coff, poff = np.zeros(NB + 1, dtype=np.uint64), np.zeros(NB + 1, dtype=np.uint64)
        cid, pid = np.zeros(max(E, 1) if lists else 1, dtype=np.uint32), np.zeros(max(E, 1) if lists else 1, dtype=np.uint32)
        self._chk(self._l.porrt_bg_get_beliefs(self._c, beliefs))
        self._chk(self._l.porrt_bg_get_node_types(self._c, types))
        self._chk(self._l.porrt_bg_get_children(self._c, coff, cid.ctypes.data_as(C.c_void_p) if lists else None))
        self._chk(self._l.porrt_bg_get_parents(self._c, poff, pid.ctypes.data_as(C.c_void_p) if lists else None))
        return beliefs, types, (coff, cid[:E] if lists else None), (poff, pid[:E] if lists else None)

    def bg_num_edges(self):
        return self._l.porrt_bg_num_edges(self._c)

    def observable_zones(self):
        m = np.zeros(self.num_nodes(), dtype=np.uint64)
        self._chk(self._l.porrt_bg_get_observable_zones(self._c, m))
        return m

    def bg_seconds(self):
        v = np.zeros(8)
        self._chk(self._l.porrt_bg_get_seconds(self._c, v, 8))
        keys = ("total_s", "device_s", "host_tables_s", "reach_s", "fold_table_s", "adjacency_s", "alloc_upload_s", "edges_fetch_s")
        return dict(zip(keys, v.tolist()))

    def compute_expected_costs(self):
        """PTO::compute_expected_costs_to_goals on the device (pto.rs:261-275)"""
        self._chk(self._l.porrt_bg_compute_expected_costs(self._c))

    def expected_costs(self):
        d = np.zeros(self._l.porrt_bg_num_nodes(self._c))
        self._chk(self._l.porrt_bg_get_expected_costs(self._c, d))
        return d

    def expected_cost_of(self, belief_node=0):
        v = C.c_double(0.0)
        self._chk(self._l.porrt_bg_expected_cost_of(self._c, belief_node, C.byref(v)))
        return v.value

    def extract_policy(self):
        """PTO::extract_policy: (original belief node ids, parents (-1 = root), leaf flags), expected cost of the root"""
        cost = C.c_double(0.0)
        n = self._l.porrt_bg_extract_policy(self._c, None, None, None, 0, C.byref(cost))
        if n < 0:
            self._chk(int(n))
        oid, par, leaf = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.uint8)
        self._l.porrt_bg_extract_policy(self._c, oid.ctypes.data_as(C.c_void_p), par.ctypes.data_as(C.c_void_p), leaf.ctypes.data_as(C.c_void_p), n, C.byref(cost))
        return (oid, par, leaf), cost.value

    def dp_info(self):
        a, b, n = C.c_double(0), C.c_double(0), C.c_uint32(0)
        self._chk(self._l.porrt_bg_get_dp_info(self._c, C.byref(a), C.byref(b), C.byref(n)))
        return dict(total_s=a.value, device_s=b.value, sweeps=n.value, sweep_rows=int(self._l.porrt_bg_get_dp_sweep_rows(self._c)))

    def selftest(self, n=1 << 20):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._l.porrt_selftest(self._c, n, C.byref(a), C.byref(b)))
        return a.value, b.value

    def grow_mm_prm(self, start, belief, max_step, search_radius, n_iter_per_belief):
        """MapShelfDomainTampPRM::grow_mm_prm (map_shelves_tamp_prm.rs:328-393); same dict as the oracle's"""
        L = self._l
        b = _f64(belief)
        self._chk(L.porrt_grow_mm_prm(self._c, _f64(start), b, len(b), max_step, search_radius, n_iter_per_belief))
        nw = len(b)
        modes, trs = [], []
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        for m in range(L.porrt_mm_num_modes(self._c)):
            bb, rp = np.zeros(nw), C.c_double(0)
            nn, ne, nf = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
            self._chk(L.porrt_mm_get_mode(self._c, m, bb, C.byref(rp), C.byref(nn), C.byref(ne), C.byref(nf)))
            xy = np.zeros((nn.value, 2))
            ef, et, fin = np.zeros(ne.value, dtype=np.uint32), np.zeros(ne.value, dtype=np.uint32), np.zeros(nf.value, dtype=np.uint64)
            self._chk(L.porrt_mm_get_mode_graph(self._c, m, p(xy), p(ef), p(et), p(fin)))
            modes.append(dict(belief=bb, reaching_probability=rp.value, xy=xy, edges=(ef.astype(np.uint64), et.astype(np.uint64)), finals=fin))
        for t in range(L.porrt_mm_num_transitions(self._c)):
            z, f, to, ob, n = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0), C.c_int(0), C.c_uint64(0)
            self._chk(L.porrt_mm_get_transition(self._c, t, C.byref(z), C.byref(f), C.byref(to), C.byref(ob), C.byref(n)))
            pairs = np.zeros((n.value, 2), dtype=np.uint64)
            self._chk(L.porrt_mm_get_transition_pairs(self._c, t, p(pairs)))
            trs.append(dict(zone=z.value, from_mode=f.value, to_mode=to.value, observation=ob.value, pairs=pairs))
        return dict(n_beliefs=L.porrt_mm_num_beliefs(self._c), modes=modes, transitions=trs)

    def mm_seconds(self):
        h, r, d = C.c_double(0), C.c_double(0), C.c_double(0)
        self._chk(self._l.porrt_mm_get_seconds(self._c, C.byref(h), C.byref(r), C.byref(d)))
        return dict(host_s=h.value, roadmap_s=r.value, device_s=d.value)

    def save_graph_json(self, path):
        """the PTO graph / PRM roadmap of the last grow as the reference's PTOGraph JSON (pto_graph.rs:22-118)"""
        self._chk(self._l.porrt_graph_save_json(self._c, os.fsencode(path)))

    def metrics(self):
        m = Metrics()
        self._chk(self._l.porrt_get_metrics(self._c, C.byref(m)))
        return {k: getattr(m, k) for k, _ in Metrics._fields_}


def conditional_dijkstra(xy, belief_row, beliefs, types, children, parents, finals, device=0):
    """porrt_conditional_dijkstra: conditional_dijkstra (belief_graph.rs:89-175) of an explicit graph on the GPU;
    children / parents are lists of lists in add_edge order"""
    L = load_library()
    n = len(types)

    def csr(lists):
        off = np.zeros(n + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(x) for x in lists])
        return off, np.array([v for x in lists for v in x] + [0], dtype=np.uint32)
    coff, cid = csr(children)
    poff, pid = csr(parents)
    beliefs = np.ascontiguousarray(beliefs, dtype=np.float64)
    dist = np.zeros(n)
    rc = L.porrt_conditional_dijkstra(device, n, np.ascontiguousarray(xy, dtype=np.float64), np.ascontiguousarray(belief_row, dtype=np.uint32),
                                      beliefs, beliefs.shape[0], beliefs.shape[1], np.ascontiguousarray(types, dtype=np.uint8), coff, cid, poff, pid,
                                      np.ascontiguousarray(finals, dtype=np.uint64), len(finals), dist)
    if rc < 0:
        raise RuntimeError("porrt_conditional_dijkstra failed (%d)" % rc)
    return dist


def exchange_decide(entries):
    """porrt_exchange_decide: entries[rank, map] (BEST_ENTRY) -> winning rank per map (-1 = nobody solved it); host code only"""
    L = load_library()
    e = np.ascontiguousarray(entries, dtype=BEST_ENTRY)
    world, n_maps = e.shape
    win = np.zeros(n_maps, dtype=np.int32)
    rc = L.porrt_exchange_decide(e.ctypes.data_as(C.c_void_p), world, n_maps, win.ctypes.data_as(C.c_void_p))
    if rc:
        raise PorrtError(rc, "porrt_exchange_decide")
    return win


def exchange_agree(words, my_rank):
    """porrt_exchange_agree: words[rank] = (status code, n_maps) -> (what rank my_rank must return, first failing rank or -1); host code only"""
    L = load_library()
    w = np.ascontiguousarray(words, dtype=np.int32).reshape(-1, 2)
    bad = C.c_int32(-1)
    rc = L.porrt_exchange_agree(w.ctypes.data_as(C.c_void_p), len(w), my_rank, C.byref(bad))
    return rc, bad.value


class Comm:
    """The one exchange of a query-sharded job (porrt_comm_*, porrt_exchange_*): RCCL over xGMI, device to device."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        rc = load_library().porrt_comm_unique_id(buf)
        if rc:
            raise PorrtError(rc, "porrt_comm_unique_id (RCCL)")
        return bytes(buf)

    def __init__(self, device, rank, world, unique_id):
        self._l = load_library()
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._c = self._l.porrt_comm_create(device, rank, world, buf)
        if not self._c:
            raise PorrtError(-4, "porrt_comm_create failed (device %d, rank %d of %d)" % (device, rank, world))
        self.rank, self.world = rank, world

    def close(self):
        if getattr(self, "_c", None):
            self._l.porrt_comm_destroy(self._c)
            self._c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def usable(self):
        """False once a collective step failed on this rank (the communicator was aborted: make a new one)"""
        return bool(self._c) and bool(self._l.porrt_comm_usable(self._c))

    def set_timeout_ms(self, ms):
        rc = self._l.porrt_comm_set_timeout_ms(self._c, int(ms))
        if rc:
            raise PorrtError(rc, "porrt_comm_set_timeout_ms")

    def exchange_best(self, engines, map_ids, n_maps):
        """collective: winners[m] (BEST_ENTRY) identical on every rank; the winning trees stay on the device (tree(m))"""
        n = len(engines)
        arr = (C.c_void_p * max(n, 1))(*[e._c for e in engines])
        ids = np.ascontiguousarray(map_ids, dtype=np.uint32)
        win = np.zeros(n_maps, dtype=BEST_ENTRY)
        rc = self._l.porrt_exchange_best(self._c, arr, n, ids, n_maps, win.ctypes.data_as(C.c_void_p))
        if rc:
            raise PorrtError(rc, self._l.porrt_comm_last_error(self._c).decode())
        return win

    def tree(self, m):
        n = self._l.porrt_exchange_num_nodes(self._c, m)
        xy, parent, dist = np.zeros((n, 2)), np.zeros(n, dtype=np.int64), np.zeros(n)
        rc = self._l.porrt_exchange_get_tree(self._c, m, xy, parent, dist)
        if rc:
            raise PorrtError(rc, self._l.porrt_comm_last_error(self._c).decode())
        return xy, parent, dist


# ---- on-disk formats (host code of the library: no GPU needed)
def read_pgm(path=None, data=None):
    """porrt_read_pgm / porrt_read_pgm_mem: the gray raster the reference's domains open (uint8 [H, W])"""
    L = load_library()
    W, H = C.c_uint32(0), C.c_uint32(0)
    if data is not None:
        call = lambda out: L.porrt_read_pgm_mem(data, len(data), out, C.byref(W), C.byref(H))
    else:
        call = lambda out: L.porrt_read_pgm(os.fsencode(path), out, C.byref(W), C.byref(H))
    rc = call(None)
    if rc:
        raise PorrtError(rc, "porrt_read_pgm: %s" % ("cannot open the file" if rc == -7 else "not an 8-bit gray PNM (the reference: \"Wrong image format!\")"))
    out = np.zeros((H.value, W.value), dtype=np.uint8)
    rc = call(out.ctypes.data_as(C.c_void_p))
    if rc:
        raise PorrtError(rc, "porrt_read_pgm")
    return out


def _u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def graph_write_json(path, xy, node_validity, children, parents, validities):
    """porrt_graph_write_json; children / parents = (offsets, ids, validity ids); validities = bool [n_validities, n_worlds]"""
    L = load_library()
    xy = _f64(xy).reshape(-1, 2)
    v = np.ascontiguousarray(np.asarray(validities, dtype=np.uint8).reshape(len(validities), -1))
    (co, ci, cv), (po, pi, pv) = children, parents
    rc = L.porrt_graph_write_json(os.fsencode(path), len(xy), xy, _u64(node_validity), _u64(co), _u64(ci), _u64(cv), _u64(po), _u64(pi), _u64(pv),
                                  v.shape[0], v.shape[1] if v.size else 0, v)
    if rc:
        raise PorrtError(rc, "porrt_graph_write_json")


def graph_load_json(path):
    """porrt_graph_load_json: dict(xy, node_validity, children=(off, ids, validity), parents=(...), validities bool [n, n_worlds])"""
    L = load_library()
    err = C.create_string_buffer(256)
    g = L.porrt_graph_load_json(os.fsencode(path), err, 256)
    if not g:
        raise PorrtError(-1, "porrt_graph_load_json: " + err.value.decode())
    try:
        n, nc, npar = L.porrt_graph_file_num_nodes(g), L.porrt_graph_file_num_children(g), L.porrt_graph_file_num_parents(g)
        nv, nw = L.porrt_graph_file_num_validities(g), L.porrt_graph_file_num_worlds(g)
        xy, val = np.zeros((n, 2)), np.zeros(n, dtype=np.uint64)
        co, ci, cv = np.zeros(n + 1, dtype=np.uint64), np.zeros(nc, dtype=np.uint64), np.zeros(nc, dtype=np.uint64)
        po, pi, pv = np.zeros(n + 1, dtype=np.uint64), np.zeros(npar, dtype=np.uint64), np.zeros(npar, dtype=np.uint64)
        vb = np.zeros((nv, nw), dtype=np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = L.porrt_graph_file_get(g, p(xy), p(val), p(co), p(ci), p(cv), p(po), p(pi), p(pv), p(vb))
        if rc:
            raise PorrtError(rc, "porrt_graph_file_get")
    finally:
        L.porrt_graph_file_free(g)
    return dict(xy=xy, node_validity=val, children=(co, ci, cv), parents=(po, pi, pv), validities=vb.astype(bool))
