"""Route planner on the MI355X: the reference's A* over motion primitives
(main/lib/mp_search_ww_generic.py + main/lib/a_star.py) for a batch of route queries, one wavefront per route
(csrc/planner.inc behind `jsim_plan_routes`).  The (M, 3) [x, y, yaw] trajectories it returns are what `MPC(cx, cy, cyaw, ...)`
/ `BatchedMPC(paths, ...)` take -- exactly the hand-over of main/scenarios/mpc_intersection.py:63-76.

The motion primitives are regenerated from the reference's recipe (main/create_motion_primitives_bicycle_model.py:12-27:
explicit-Euler kinematic bicycle from the origin, 8.3 m/s, nine steering angles, 61 states 0.01 s apart -- 4.98 m, 0.083 m
between points); the reference's pickled primitives are never loaded.  Scenario geometry restates main/envs/intersection.py
and main/envs/intersection_multi_lanes.py (boxes and circles, turned into half-plane sets like Obstacle.to_convex).
There is no CPU fallback: without the HIP library / a HIP device `plan_routes` raises."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _cabi

MP_NAMES = ("straight", "left1", "left2", "left3", "left4", "right1", "right2", "right3", "right4")
MP_STEER = (0.0, 0.1, 0.2, 0.3, 0.4, -0.1, -0.2, -0.3, -0.4)     # main/create_motion_primitives_prius.py:19-29
WH_DEFAULT = (1.0, 2.7, 15.0, 0.0, 0.0)                            # mp_search_ww_generic.py:29-31
WC_DEFAULT = (1.0, 5.0, 0.1, 0.0)                                  # :33


def make_motion_primitives(L: float = 2.86, v: float = 8.3, n_steps: int = 60, dt: float = 0.01):
    """(points [9, 61, 3], total_length [9]) in MP_NAMES order: Bicycle.step (main/bicycle/main.py:28-41) from the origin, the
    state recorded before every step."""
    pts = np.zeros((len(MP_STEER), n_steps + 1, 3))
    for k, delta in enumerate(MP_STEER):
        x = y = th = 0.0
        for i in range(n_steps + 1):
            pts[k, i] = (x, y, th)
            xd = v * np.cos(th); yd = v * np.sin(th); thd = (v / L) * np.tan(delta)
            x += xd * dt; y += yd * dt; th += thd * dt
    length = np.array([np.linalg.norm(p[:-1, :2] - p[1:, :2], axis=1).sum() for p in pts])
    return pts, length


def car_circles(L: float = 2.86, width: float = 2.0, extra_length: float = 0.64):
    """(radius, circle centres [2, 2]) of BicycleModelDimensions (main/lib/car_dimensions.py:52-90)."""
    length = L + extra_length
    off = length / 2 - width / 2
    return width / (2 ** .5), np.array([[L / 2 + off, 0.0], [L / 2 - off, 0.0]])


def collision_points(mp_points: np.ndarray, circle_centers: np.ndarray, radius: float) -> np.ndarray:
    """_create_collision_points for one primitive (mp_search_ww_generic.py:118-136): resample_curve at dl = radius keeping the
    last point (main/lib/trajectories.py:58-86), then each collision circle's trajectory (:11-55); xy only."""
    step = np.append(0.0, np.linalg.norm(mp_points[1:, :2] - mp_points[:-1, :2], axis=1))
    k = np.floor(step.cumsum() / radius).astype(int)
    mask = np.append(True, (k[1:] - k[:-1]) >= 1.0)
    mask[-1] = True
    p = mp_points[mask]
    th = p[:, 2]
    out = [np.vstack([np.cos(th) * cx - np.sin(th) * cy, np.sin(th) * cx + np.cos(th) * cy]).T + p[:, :2] for cx, cy in circle_centers]
    return np.concatenate(out, axis=0)


def box_halfplanes(xy_width, xy_center, margin: float) -> np.ndarray:
    """BoxObstacle.to_convex (main/lib/obstacles.py:80-93)."""
    cx, cy = xy_center
    w, h = xy_width
    x1, y1, x2, y2 = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
    return np.array([[1, 0, -(x2 + margin)], [-1, 0, x1 - margin], [0, 1, -(y2 + margin)], [0, -1, y1 - margin]], dtype=np.float64)


def circle_halfplanes(radius: float, xy_center, margin: float) -> np.ndarray:
    """CircleObstacle.to_convex (main/lib/obstacles.py:127-142): the octagon around the circle."""
    cx, cy = xy_center
    r = radius
    return np.array([[1, 0, -(cx + r + margin)], [-1, 0, cx - r - margin], [0, 1, -(cy + r + margin)], [0, -1, cy - r - margin],
                     [-1, 1, cx - cy - r * np.sqrt(2) - 2 * margin], [1, -1, -cx + cy - r * np.sqrt(2) - 2 * margin],
                     [-1, -1, cx + cy - r * np.sqrt(2) - 2 * margin], [1, 1, -cx - cy - r * np.sqrt(2) - 2 * margin]], dtype=np.float64)


@dataclass
class RouteQuery:
    start: Tuple[float, float, float]
    goal: Tuple[float, float, float]
    goal_box: Tuple[float, float, float, float]     # (x1, y1, x2, y2) of the goal area
    tol: float                                      # allowed_goal_theta_difference
    obstacles: List[np.ndarray]                     # half-plane sets [a, b, c], one array per obstacle (margin included)


def intersection_query(start_pos: int, turn_indicator: int, margin: float, start_lane: int = 1, goal_lane: int = 1,
                       number_of_lanes: int = 0) -> RouteQuery:
    """The reference's scenarios as a route query: number_of_lanes = 0 -> main/envs/intersection.py:10-160 (one lane per
    direction, 4 m road, 2 m island, corner radius 6); number_of_lanes >= 1 -> main/envs/intersection_multi_lanes.py:9-170
    (lane_width 4, median 2, as mpc_intersection_multi_lane.py builds it with number_of_lanes = 2)."""
    pi = np.pi
    if number_of_lanes == 0:
        road, island, pav, length, corner = 4, 2, 5, 30, 6
        dc = corner + road + island
        lat = island / 2 + road / 2
        starts = {1: (lat, -30, 0.5 * pi), 2: (-30, -lat, 0), 3: (-lat, 30, -0.5 * pi), 4: (30, lat, pi)}
        g = (island + road) / 2
        gd = 30
        half_road = island / 2 + road
        corner_r = dc - island / 2 - road
        goal_w = (road * 1.8, road)
    else:
        lane_w, island, pav, length, corner = 4, 2, 5, 30, 6
        dc = corner + lane_w * number_of_lanes + island
        lat = island / 2 + (start_lane - 1) * lane_w + lane_w / 2
        starts = {1: (lat, -30, 0.5 * pi), 2: (-30, -lat, 0), 3: (-lat, 30, -0.5 * pi), 4: (30, lat, pi)}
        g = (island + lane_w) / 2 + (goal_lane - 1) * lane_w
        gd = 30
        half_road = island / 2 + number_of_lanes * lane_w
        corner_r = dc - island / 2 - number_of_lanes * lane_w
        goal_w = (lane_w * 1.8, 1.5)
    W, N_, E, S_ = (-gd, g, -pi), (g, gd, 0.5 * pi), (gd, -g, 0), (-g, -gd, -0.5 * pi)
    goals = {1: {1: W, 2: N_, 3: E}, 2: {1: N_, 2: E, 3: S_}, 3: {1: E, 2: S_, 3: W}, 4: {1: S_, 2: W, 3: N_}}
    start, goal = starts[start_pos], goals[start_pos][turn_indicator]
    if (start_pos in (1, 3) and turn_indicator in (1, 3)) or (start_pos in (2, 4) and turn_indicator in (2, 4)):
        gw = goal_w
    else:
        gw = (goal_w[1], goal_w[0])
    gb = (goal[0] - gw[0] / 2, goal[1] - gw[1] / 2, goal[0] + gw[0] / 2, goal[1] + gw[1] / 2)
    m = margin
    far = length / 2 + dc
    side = half_road + pav / 2
    obs = [
        box_halfplanes((island, length), (0, -far), m), circle_halfplanes(island / 2, (0, -dc), m),          # south median
        box_halfplanes((island, length), (0, far), m), circle_halfplanes(island / 2, (0, dc), m),            # north median
        box_halfplanes((length, island), (-far, 0), m), circle_halfplanes(island / 2, (-dc, 0), m),          # west median
        box_halfplanes((length, island), (far, 0), m), circle_halfplanes(island / 2, (dc, 0), m),            # east median
        circle_halfplanes(corner_r, (-dc, -dc), m), circle_halfplanes(corner_r, (-dc, dc), m),                # corners
        circle_halfplanes(corner_r, (dc, dc), m), circle_halfplanes(corner_r, (dc, -dc), m),
        box_halfplanes((pav, length), (-side, -far), m), box_halfplanes((pav, length), (side, -far), m),       # pavements: south
        box_halfplanes((length, pav), (-far, -side), m), box_halfplanes((length, pav), (-far, side), m),       # west
        box_halfplanes((pav, length), (-side, far), m), box_halfplanes((pav, length), (side, far), m),         # north
        box_halfplanes((length, pav), (far, -side), m), box_halfplanes((length, pav), (far, side), m),         # east
    ]
    # the hidden boxes that close the oncoming lanes of the four arms (envs/intersection.py:150-208, intersection_multi_lanes.py
    # :183-213): lateral sign per (start_pos, arm W / E / S / N); the multi-lane file's east box for start_pos 4 is centred with
    # lane_width instead of number_of_lanes * lane_width (kept as written)
    road_total = road if number_of_lanes == 0 else number_of_lanes * lane_w
    gl_ = (road_total + island) / 2
    sign = {1: (-1, 1, -1, -1), 2: (1, 1, 1, -1), 3: (-1, 1, 1, 1), 4: (-1, -1, 1, -1)}[start_pos]
    ge = (lane_w + island) / 2 if (number_of_lanes and start_pos == 4) else gl_
    obs += [box_halfplanes((length, road_total), (-far, sign[0] * gl_), m), box_halfplanes((length, road_total), (far, sign[1] * ge), m),
            box_halfplanes((road_total, length), (sign[2] * gl_, -far), m), box_halfplanes((road_total, length), (sign[3] * gl_, far), m)]
    return RouteQuery(start=tuple(float(v) for v in start), goal=tuple(float(v) for v in goal), goal_box=tuple(float(v) for v in gb),
                      tol=float(np.pi / 16), obstacles=obs)


@dataclass
class PlannedRoute:
    status: int
    cost: float
    prims: np.ndarray          # primitive index per segment (into MP_NAMES)
    nodes: np.ndarray          # (n_prims + 1, 3) poses of the path
    trajectory: np.ndarray     # (n_prims * 60, 3) [x, y, yaw]: what MPC(cx, cy, cyaw) takes
    n_expanded: int


def plan_routes(queries: Sequence[RouteQuery], L: float = 2.86, wh=WH_DEFAULT, wc=WC_DEFAULT, max_path: int = 32,
                device: int = 0, primitives=None, node_cap: int = 1 << 17, retry_node_cap: int = 1 << 21, circles=None) -> List[PlannedRoute]:
    """All queries in ONE launch (one wavefront per route).  Routes whose search outgrows `node_cap` nodes (status 4) are planned
    again, together, with `retry_node_cap` (0: no second attempt).  primitives = (points [P, n, 3], total_length [P]) and
    circles = (radius, centres [k, 2]) replace the regenerated bicycle-model set / BicycleModelDimensions' circles."""
    out = _plan(queries, L, wh, wc, max_path, device, primitives, node_cap, circles)
    again = [i for i, r in enumerate(out) if r.status == 4]
    if again and retry_node_cap > node_cap:
        for i, r in zip(again, _plan([queries[i] for i in again], L, wh, wc, max_path, device, primitives, retry_node_cap, circles)):
            out[i] = r
    return out


class MotionPrimitiveSearch:
    """Drop-in for the reference's planner class (main/lib/mp_search_ww_generic.py:26-58 constructor, :136-140 run): the same
    arguments -- a scenario (.start, .goal_point, .goal_area with .xy1 / .xy2, .allowed_goal_theta_difference, .obstacles whose
    .to_convex(margin) gives half-planes), car dimensions (.radius, .circle_centers), the motion primitives the CALLER holds (a
    dict name -> object with .points (n, 3) and .total_length; iteration order = expansion order, as in the reference) and the
    nine weights -- and run() -> (cost, path, trajectory) with path a list of (x, y, theta) tuples.  The search itself is
    jsim_plan_routes on the GPU (one route = one wavefront; use plan_routes for many routes at once).  Like the reference,
    run() raises Exception("No solution found.") when the open list runs empty (main/lib/a_star.py:78); debug=True (the
    reference's matplotlib trace of the expansion) is not offered."""

    def __init__(self, scenario, car_dimensions, mps, margin: float,
                 wh_dist: float = 1.0, wh_theta: float = 2.7, wh_steering: float = 15.0, wh_obstacle: float = 0.0, wh_center: float = 0.0,
                 wc_dist: float = 1.0, wc_steering: float = 5.0, wc_obstacle: float = 0.1, wc_center: float = 0.0, device: int = 0,
                 max_path: int = 32, node_cap: int = 1 << 17, retry_node_cap: int = 1 << 21):
        # max_path: longest path in primitives the output arrays hold (status 6 beyond); node_cap / retry_node_cap: node table of
        # the first / second attempt (status 4 beyond) -- the reference's dict and heap grow without bound, these do not
        self.max_path, self.node_cap, self.retry_node_cap = int(max_path), int(node_cap), int(retry_node_cap)
        self._names = list(mps.keys())
        pts = [np.asarray(mps[n].points, dtype=np.float64) for n in self._names]
        if not pts or any(p.shape != pts[0].shape or p.ndim != 2 or p.shape[1] != 3 for p in pts):
            raise ValueError("motion primitives must be (n, 3) arrays of one common length")
        self._primitives = (np.stack(pts), np.array([float(mps[n].total_length) for n in self._names]))
        self._circles = (float(car_dimensions.radius), np.asarray(car_dimensions.circle_centers, dtype=np.float64).reshape(-1, 2))
        ga = scenario.goal_area
        if not (hasattr(ga, "xy1") and hasattr(ga, "xy2")):
            # the reference accepts any Obstacle with distance_to_point as goal area (main/lib/mp_search_ww_generic.py:101-103); every
            # scenario builder it ships uses a BoxObstacle, and the kernel's goal test is the box test
            raise ValueError(f"goal_area must be a box (.xy1 / .xy2), got {type(ga).__name__}: the GPU planner's goal test is the "
                             "reference's BoxObstacle.distance_to_point(...) <= 0")
        (x1, y1), (x2, y2) = ga.xy1, ga.xy2
        self._query = RouteQuery(start=tuple(float(v) for v in scenario.start), goal=tuple(float(v) for v in scenario.goal_point),
                                 goal_box=(float(x1), float(y1), float(x2), float(y2)), tol=float(scenario.allowed_goal_theta_difference),
                                 obstacles=[np.asarray(o.to_convex(margin=margin), dtype=np.float64) for o in scenario.obstacles])
        self._wh = (wh_dist, wh_theta, wh_steering, wh_obstacle, wh_center)
        self._wc = (wc_dist, wc_steering, wc_obstacle, wc_center)
        self._device = device
        self._points_to_mp_names = {}
        self.last = None       # the PlannedRoute of the last run (status, primitive indices, expansion count)

    def run(self, debug: bool = False):
        if debug:
            raise NotImplementedError("debug=True (the reference's expansion trace) is not offered by the GPU planner")
        r = plan_routes([self._query], wh=self._wh, wc=self._wc, primitives=self._primitives, circles=self._circles, device=self._device,
                        max_path=self.max_path, node_cap=self.node_cap, retry_node_cap=self.retry_node_cap)[0]
        self.last = r
        if r.status == 1:
            raise Exception("No solution found.")                    # main/lib/a_star.py:78
        if r.status != 0:
            raise RuntimeError(f"route planner: status {r.status} (4: node table full -- raise node_cap / retry_node_cap; 5 / 6: path "
                               f"longer than max_path = {self.max_path} primitives)")
        path = [tuple(float(v) for v in n) for n in r.nodes]
        for a, b, k in zip(path[:-1], path[1:], r.prims):
            self._points_to_mp_names[a, b] = self._names[int(k)]
        return r.cost, path, r.trajectory


def _plan(queries, L, wh, wc, max_path, device, primitives, node_cap, circles=None) -> List[PlannedRoute]:
    lib = _cabi.load()
    pts, length = primitives if primitives is not None else make_motion_primitives(L=L)
    radius, centres = circles if circles is not None else car_circles(L=L)
    cc = [collision_points(p, centres, radius) for p in pts]
    cc_off = np.concatenate([[0], np.cumsum([len(c) for c in cc])]).astype(np.int32)
    cc_flat = np.ascontiguousarray(np.concatenate(cc, axis=0), dtype=np.float64)
    R = len(queries)
    hp_list, hp_off, r_off = [], [0], [0]
    for q in queries:
        for o in q.obstacles:
            o = np.asarray(o, dtype=np.float64).reshape(-1, 3)
            hp_list.append(o)
            hp_off.append(hp_off[-1] + len(o))
        r_off.append(len(hp_off) - 1)
    hp = np.ascontiguousarray(np.concatenate(hp_list, axis=0) if hp_list else np.zeros((0, 3)), dtype=np.float64)
    hp_off = np.array(hp_off, dtype=np.int32); r_off = np.array(r_off, dtype=np.int32)
    start = np.ascontiguousarray([q.start for q in queries], dtype=np.float64).reshape(R, 3)
    goal = np.ascontiguousarray([q.goal for q in queries], dtype=np.float64).reshape(R, 3)
    box = np.ascontiguousarray([q.goal_box for q in queries], dtype=np.float64).reshape(R, 4)
    tol = np.ascontiguousarray([q.tol for q in queries], dtype=np.float64)
    mp = np.ascontiguousarray(pts, dtype=np.float64); ml = np.ascontiguousarray(length, dtype=np.float64)
    whv = np.ascontiguousarray(wh, dtype=np.float64); wcv = np.ascontiguousarray(wc, dtype=np.float64)
    n_prim, n_pts = mp.shape[0], mp.shape[1]
    seg = n_pts - 1
    status = np.zeros(R, dtype=np.int32); cost = np.zeros(R); n_prims = np.zeros(R, dtype=np.int32)
    prims = np.zeros((R, max_path), dtype=np.int32); nodes = np.zeros((R, max_path + 1, 3)); traj = np.zeros((R, max_path * seg, 3))
    n_exp = np.zeros(R, dtype=np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)
    _cabi.check(lib.jsim_plan_routes(int(device), R, p(start), p(goal), p(box), p(tol), p(hp), p(hp_off), len(hp_off) - 1, p(r_off), p(mp),
                                     p(ml), n_prim, n_pts, p(cc_flat), p(cc_off), p(whv), p(wcv), int(max_path), int(node_cap), p(status), p(cost),
                                     p(n_prims), p(prims), p(nodes), p(traj), p(n_exp)), None, "jsim_plan_routes")
    out = []
    for i in range(R):
        k = int(n_prims[i])
        out.append(PlannedRoute(status=int(status[i]), cost=float(cost[i]), prims=prims[i, :k].copy(), nodes=nodes[i, :k + 1].copy(),
                                trajectory=traj[i, :k * seg].copy(), n_expanded=int(n_exp[i])))
    return out
