"""Planning problems shared by the oracle tests and the GPU parity tests.

The coordinates, radii and iteration counts follow the reference's drivers
(src/main.rs:486-848, src/pto.rs:298-546, src/rrt.rs:254-416); the maps are the
synthetic stand-ins of tools/make_maps.py (the reference's rasters are LFS pointers).
"""
import os

import numpy as np

from make_maps import read_pgm

MAPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maps")
SHELF, DOOR = 0, 1
RRT, PTO = 0, 1


def load_map(name):
    return read_pgm(os.path.join(MAPS, name + ".pgm"))


def onehot(k):
    return np.uint64(1) << np.uint64(k)


class Case(dict):
    __getattr__ = dict.__getitem__


def cfg1(n_iter=5000, seed=0):
    """map0-like single-goal 2D RRT* (BASELINE.json configs[0])."""
    return Case(name="cfg1", grid="map0_like", zones=None, visibility=0.0, domain=SHELF, mode=RRT,
                start=(0.0, 0.0), goals=[(0.0, 0.9)], masks=[1], l1=0.05, obs_zone=None,
                max_step=0.1, search_radius=2.0, n_iter_min=n_iter, n_iter_max=n_iter, seed=seed)


def cfg2(n_iter=125000, seed=0, grid="map_benchmark_like"):
    """map_benchmark-like 2D RRT*, the headline workload (configs[1]; main.rs:532,767)."""
    return Case(name="cfg2", grid=grid, zones=None, visibility=0.0, domain=SHELF, mode=RRT,
                start=(0.0, -1.0), goals=[(0.9, 0.0)], masks=[1], l1=0.05, obs_zone=None,
                max_step=0.1, search_radius=2.0, n_iter_min=n_iter, n_iter_max=n_iter, seed=seed)


def cfg2_obs(n_iter=2500, seed=0):
    """RRT* towards an observation goal (rrt.rs:305-360, tamp_rrt.rs:49-65)."""
    c = cfg2(n_iter, seed)
    c.update(name="cfg2_obs", zones="map_benchmark_like_6_goals_zone_ids", visibility=0.5, obs_zone=2,
             goals=None, n_iter_max=4 * n_iter, start=(0.0, -0.8))
    return c


def cfg3(n_iter_min=2000, n_iter_max=100000, seed=0):
    """2-world belief-space RRG on a shelf map (configs[2]; pto.rs:466-479)."""
    return Case(name="cfg3", grid="map1_2_goals_like", zones="map1_2_goals_like_zone_ids", visibility=0.5,
                domain=SHELF, mode=PTO, start=(-0.8, -0.8), goals=[(0.68, -0.45), (0.68, 0.38)],
                masks=[1, 2], l1=0.05, obs_zone=None, max_step=0.05, search_radius=5.0,
                n_iter_min=n_iter_min, n_iter_max=n_iter_max, seed=seed)


def cfg4(n_iter_min=5000, n_iter_max=100000, seed=0):
    """12-world growth on the 4x3 shelf lattice (configs[3]; main.rs:386-408)."""
    goals = [(x, y) for y in (0.75, 0.25, -0.25) for x in (-0.75, -0.25, 0.25, 0.75)]
    return Case(name="cfg4", grid="map5_like", zones="map5_like_12_goals_zone_ids", visibility=0.2,
                domain=SHELF, mode=PTO, start=(0.0, -0.8), goals=goals, masks=[1 << k for k in range(12)],
                l1=0.05, obs_zone=None, max_step=0.05, search_radius=5.0,
                n_iter_min=n_iter_min, n_iter_max=n_iter_max, seed=seed)


def cfg_door(n_iter_min=2000, n_iter_max=100000, seed=0, paper=False):
    """Door domain (2^n worlds): the paper's map_4 raster (16 worlds, main.rs:207-216) or a
    synthetic two-door map."""
    if paper:
        return Case(name="cfg_door_paper", grid="paper_map_4", zones="paper_map_4_zone_ids", visibility=0.3,
                    domain=DOOR, mode=PTO, start=(0.55, -0.8), goals=[(0.55, 0.9)], masks=[(1 << 16) - 1],
                    l1=0.05, obs_zone=None, max_step=0.05, search_radius=5.0,
                    n_iter_min=n_iter_min, n_iter_max=n_iter_max, seed=seed)
    return Case(name="cfg_door", grid="door_map_like", zones="door_map_like_zone_ids", visibility=0.3,
                domain=DOOR, mode=PTO, start=(0.5, -0.6), goals=[(-0.5, 0.6)], masks=[(1 << 4) - 1],
                l1=0.05, obs_zone=None, max_step=0.05, search_radius=5.0,
                n_iter_min=n_iter_min, n_iter_max=n_iter_max, seed=seed)


def empty_space(n_iter_min=1000, n_iter_max=10000, seed=0):
    """rrt.rs:254-267 test_plan_empty_space."""
    return Case(name="empty", grid=None, zones=None, visibility=0.0, domain=SHELF, mode=RRT,
                start=(0.0, 0.0), goals=[(0.9, 0.9)], masks=[1], l1=0.05, obs_zone=None,
                max_step=0.1, search_radius=1.0, n_iter_min=n_iter_min, n_iter_max=n_iter_max, seed=seed)


def cfg_big(mode=RRT, n_iter=6000, seed=0):
    """400 x 400 raster over [-1, 1)^2 (ppm 200; the reference opens 400 x 400 maps: data/map2_fov.pgm, map_io.rs:699, pto.rs:387-389):
    RRT* with cfg2's parameters, or the 2-world belief-space RRG of cfg3 on it."""
    if mode == RRT:
        return Case(name="cfg_big_rrt", grid="big_shelf_map_400", zones=None, visibility=0.0, domain=SHELF, mode=RRT,
                    start=(-0.8, -0.8), goals=[(0.68, 0.38)], masks=[1], l1=0.05, obs_zone=None,
                    max_step=0.1, search_radius=2.0, n_iter_min=n_iter, n_iter_max=n_iter, seed=seed)
    return Case(name="cfg_big_pto", grid="big_shelf_map_400", zones="big_shelf_map_400_zone_ids", visibility=0.5, domain=SHELF, mode=PTO,
                start=(-0.8, -0.8), goals=[(0.68, -0.45), (0.68, 0.38)], masks=[1, 2], l1=0.05, obs_zone=None,
                max_step=0.05, search_radius=5.0, n_iter_min=n_iter, n_iter_max=n_iter, seed=seed)


def cfg_wide(mode=PTO, n_iter=4000, seed=0):
    """300 x 200 raster over [-1.5, 1.5) x [-1, 1) (W != H; the transform of map_io.rs:176-181 is generic in both): door domain,
    two doors (4 worlds); the sampler draws from the map's box."""
    base = dict(grid="wide_door_map_300x200", low=(-1.5, -1.0), up=(1.5, 1.0), domain=DOOR, l1=0.05, obs_zone=None,
                n_iter_min=n_iter, n_iter_max=n_iter, seed=seed)
    if mode == RRT:      # (RRT* is only ever driven with the shelf adapter in the reference, tamp_rrt.rs:35-47: doors read as low obstacles)
        return Case(name="cfg_wide_rrt", zones=None, visibility=0.0, mode=RRT, start=(-1.2, -0.7), goals=[(1.2, 0.7)], masks=[1],
                    max_step=0.1, search_radius=2.0, **dict(base, domain=SHELF))
    return Case(name="cfg_wide_pto", zones="wide_door_map_300x200_zone_ids", visibility=0.4, mode=PTO, start=(-1.2, -0.7),
                goals=[(1.2, 0.7)], masks=[(1 << 4) - 1], max_step=0.05, search_radius=5.0, **base)


def cfg_map4(n_iter_min=5000, seed=0):
    """The reference's recorded end-to-end problem on its one recoverable raster (main.rs:893-908 test_plan_on_navigation_map4_pomdp;
    results/maps_paper/map_4/costs_and_timings_*.txt): start (0.8, -0.8), goal (-0.8, 0.8) in all 16 worlds, max_step 0.1,
    search_radius 5, visibility 0.25, n_iter_min 5000 / n_iter_max 100000, uniform prior."""
    return Case(name="cfg_map4", grid="paper_map_4", zones="paper_map_4_zone_ids", visibility=0.25, domain=DOOR, mode=PTO,
                start=(0.8, -0.8), goals=[(-0.8, 0.8)], masks=[(1 << 16) - 1], l1=0.05, obs_zone=None,
                max_step=0.1, search_radius=5.0, n_iter_min=n_iter_min, n_iter_max=100000, seed=seed)


def configure(eng, case):
    """Apply a case to an oracle or engine object (same method names on both)."""
    low, up = case.get("low", (-1.0, -1.0)), case.get("up", (1.0, 1.0))
    if case.grid is not None:
        eng.set_grid(load_map(case.grid), low, up, case.domain)
        if case.zones is not None:
            eng.set_zones(load_map(case.zones), case.visibility)
    eng.set_sampler(low, up, case.seed)
    if case.obs_zone is not None:
        eng.set_observation_goal(case.obs_zone)
    else:
        eng.set_square_goal(np.array(case.goals, dtype=np.float64), np.array(case.masks, dtype=np.uint64), case.l1)
    return eng


def grow(eng, case, K=1, **kw):
    return eng.grow(case.start, case.max_step, case.search_radius, case.n_iter_min, case.n_iter_max,
                    batch_K=K, mode=case.mode, **kw)


def cfg3_near(n_iter=1500, seed=0):
    """cfg3 with both goals on the start's side of the wall: a small graph already has a finite expected cost at the
    root (the belief splits at the first shelf seen, one goal per world is reached)."""
    c = cfg3(n_iter, n_iter, seed)
    c.update(name="cfg3_near", goals=[(0.68, -0.45), (0.5, -0.8)], start=(0.2, -0.6))
    return c


def tamp_queries(n, seed0=0):
    """Queries shaped like the TAMP search's (map_shelves_tamp_rrt.rs:224,232,355,367,492,500; main.rs:532): every search edge plans
    twice -- to an ObservationGoal (see the shelf of a zone) and to a SquareGoal (pickup, L1 radius 0.05) -- with starts of their own,
    max_step 0.1, search_radius 2, n_iter_min 2500, n_iter_max 10000, on the benchmark map with its six zones."""
    z6 = [(-0.9, -0.5), (-0.9, 0.5), (-0.5, 0.9), (0.5, 0.9), (0.9, 0.5), (0.9, -0.5)]
    starts = [(0.0, -1.0), (0.0, -0.8), (-0.3, -0.2), (0.3, 0.1), (-0.6, 0.3), (0.6, -0.4)]
    out = []
    for q in range(n):
        c = cfg2(2500, seed=seed0 + q)
        c.update(n_iter_max=10000, start=starts[(q // 2) % len(starts)], zones="map_benchmark_like_6_goals_zone_ids", visibility=0.5)
        if q % 2 == 0:
            c.update(name="tamp_obs", obs_zone=(q // 2) % 6, goals=None)
        else:
            gx, gy = z6[(q // 2) % 6]
            c.update(name="tamp_pick", goals=[(0.92 * gx, 0.92 * gy)])
        out.append(c)
    return out
