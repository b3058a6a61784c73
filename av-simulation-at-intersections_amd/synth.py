"""Seeded synthetic inputs for the batched MPC path.

Routes: the reference's planner output is an (M, 3) float64 array [x, y, yaw] with uniform
spacing dl = 0.083 m; 720 points for the left-turn / straight routes of its 4-arm
`intersection()` scenario and 600 for the right turns.  That planner needs the reference's pickled
motion primitives, which are never loaded here, so the routes below are synthetic arcs of the same
shape, spacing, length and start poses (straight approach, constant-radius turn, straight exit),
with yaw wrapped to (-pi, pi] as the planner emits it (so `smooth_yaw` has work to do).

Egos: state, warm start and per-ego path truncation drawn as SURVEY.md 8d prescribes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

DL = 0.083
DT = 0.2
WHEELBASE = 2.86
SPEED = 30.0 / 3.6


def _wrap(a: np.ndarray) -> np.ndarray:
    return (a + np.pi) % (2.0 * np.pi) - np.pi


def make_route(start_pos: int, turn: int, dl: float = DL, lane_offset: float = 3.0,
               arm: float = 30.0) -> np.ndarray:
    """One route of a 4-arm intersection.  start_pos in 1..4 (south, west, north, east arm),
    turn in 1..3 (left, straight, right).  Returns (M, 3) [x, y, yaw]."""
    M = 600 if turn == 3 else 720
    s = np.arange(M, dtype=np.float64) * dl
    # canonical frame: start at (lane_offset, -arm), heading +y
    if turn == 2:
        x = np.full(M, lane_offset)
        y = -arm + s
        yaw = np.full(M, math.pi / 2)
    else:
        left = turn == 1
        r = 11.0 if left else 5.0
        l1 = 22.0
        arc = r * math.pi / 2
        x = np.empty(M)
        y = np.empty(M)
        yaw = np.empty(M)
        sgn = 1.0 if left else -1.0
        cxo = lane_offset - sgn * r  # turn centre
        cyo = -arm + l1
        for i, si in enumerate(s):
            if si <= l1:
                x[i], y[i], yaw[i] = lane_offset, -arm + si, math.pi / 2
            elif si <= l1 + arc:
                th = (si - l1) / r
                x[i] = cxo + sgn * r * math.cos(th)
                y[i] = cyo + r * math.sin(th)
                yaw[i] = math.pi / 2 + sgn * th
            else:
                rem = si - l1 - arc
                x[i] = cxo - sgn * rem
                y[i] = cyo + r
                yaw[i] = math.pi / 2 + sgn * math.pi / 2
    # rotate the canonical (south-arm) route to the requested arm
    rot = {1: 0.0, 2: -math.pi / 2, 3: math.pi, 4: math.pi / 2}[start_pos]
    c, sn = math.cos(rot), math.sin(rot)
    xr = c * x - sn * y
    yr = sn * x + c * y
    return np.stack([xr, yr, _wrap(yaw + rot)], axis=1)


def _resample_uniform(x: np.ndarray, y: np.ndarray, dl: float) -> np.ndarray:
    """Points at uniform arclength spacing dl along the polyline (x, y), yaw = direction of travel."""
    seg = np.hypot(np.diff(x), np.diff(y))
    s = np.concatenate([[0.0], np.cumsum(seg)])
    M = int(math.floor(s[-1] / dl)) + 1
    si = np.arange(M, dtype=np.float64) * dl
    xi, yi = np.interp(si, s, x), np.interp(si, s, y)
    h = 1e-3
    yaw = np.arctan2(np.interp(si + h, s, y) - np.interp(si - h, s, y), np.interp(si + h, s, x) - np.interp(si - h, s, x))
    return np.stack([xi, yi, yaw], axis=1)


def make_multi_lane_route(start_pos: int, turn: int, start_lane: int, goal_lane: int, dl: float = DL) -> np.ndarray:
    """One route of the reference's two-lane 4-arm intersection (main/envs/intersection_multi_lanes.py:9-60 with
    number_of_lanes = 2, as main/scenarios/mpc_intersection_multi_lane.py:41-46 builds it): lane_width 4, median_width 2,
    start / goal distance 30 -- lane centres 3 m and 7 m from the road axis.  Starts at (lane centre, -30) heading +y in the
    canonical (south-arm) frame; left turns end heading -x at y = +goal lane centre, right turns heading +x at y = -goal lane
    centre, straight routes change from the start lane to the goal lane with a smooth S-curve inside the junction.  Synthetic
    stand-in for the planner's output (same [x, y, yaw] layout, spacing and extent; the planner needs the reference's pickled
    motion primitives, which are never loaded)."""
    lane_w, median, dist = 4.0, 2.0, 30.0
    sx = median / 2 + (start_lane - 1) * lane_w + lane_w / 2
    gl = (median + lane_w) / 2 + (goal_lane - 1) * lane_w
    h = 0.01
    if turn == 2:
        y = np.arange(-dist, dist + h, h)
        blend = np.clip((y + 10.0) / 20.0, 0.0, 1.0)
        x = sx + (gl - sx) * 0.5 * (1.0 - np.cos(np.pi * blend))
    else:
        left = turn == 1
        r = (8.0 + sx) if left else (2.0 + sx)
        sgn = 1.0 if left else -1.0
        y_end = gl if left else -gl
        y_t = y_end - r
        cxo = sx - sgn * r
        th = np.arange(0.0, np.pi / 2, h / r)
        y1 = np.arange(-dist, y_t, h)
        x2 = np.arange(h, dist - abs(cxo) + h, h)
        x = np.concatenate([np.full(len(y1), sx), cxo + sgn * r * np.cos(th), cxo - sgn * x2])
        y = np.concatenate([y1, y_t + r * np.sin(th), np.full(len(x2), y_end)])
    pts = _resample_uniform(x, y, dl)
    rot = {1: 0.0, 2: -math.pi / 2, 3: math.pi, 4: math.pi / 2}[start_pos]
    c, sn = math.cos(rot), math.sin(rot)
    return np.stack([c * pts[:, 0] - sn * pts[:, 1], sn * pts[:, 0] + c * pts[:, 1], _wrap(pts[:, 2] + rot)], axis=1)


def make_route_table(multi_lane: bool = False) -> List[np.ndarray]:
    """The 12 routes (4 arms x 3 turns) of the single-lane intersection, or with multi_lane=True the 48 routes
    (4 arms x 3 turns x 2 start lanes x 2 goal lanes) of mpc_intersection_multi_lane's two-lane geometry."""
    if multi_lane:
        return [make_multi_lane_route(sp, tn, sl, gl) for sp in (1, 2, 3, 4) for tn in (1, 2, 3) for sl in (1, 2)
                for gl in (1, 2)]
    return [make_route(sp, tn, lane_offset=3.0, arm=30.0) for sp in (1, 2, 3, 4) for tn in (1, 2, 3)]


def smooth_yaw_inplace(yaw: np.ndarray) -> np.ndarray:
    """Host-side unwrapping with the reference's semantics (main/lib/mpc.py:46-58): successive
    differences end up in (-pi/2, pi/2); mutates and returns `yaw`."""
    for i in range(len(yaw) - 1):
        dyaw = yaw[i + 1] - yaw[i]
        while dyaw >= math.pi / 2.0:
            yaw[i + 1] -= math.pi * 2.0
            dyaw = yaw[i + 1] - yaw[i]
        while dyaw <= -math.pi / 2.0:
            yaw[i + 1] += math.pi * 2.0
            dyaw = yaw[i + 1] - yaw[i]
    return yaw


def pack_paths(routes: List[np.ndarray]) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Concatenate routes into the C-ABI's flat path table: cx, cy, cyaw, path_off[n_paths+1]."""
    off = np.zeros(len(routes) + 1, dtype=np.int64)
    for i, r in enumerate(routes):
        off[i + 1] = off[i] + r.shape[0]
    cat = np.concatenate(routes, axis=0)
    return (np.ascontiguousarray(cat[:, 0]), np.ascontiguousarray(cat[:, 1]),
            np.ascontiguousarray(cat[:, 2]), off)


@dataclass
class EgoBatch:
    x0: np.ndarray          # [B, 4] (x, y, v, yaw)  -- MPC state order, main/lib/mpc.py:291
    path_id: np.ndarray     # [B] int32
    path_len: np.ndarray    # [B] int32  (M' <= M: truncated path, reference semantics D2)
    target_ind: np.ndarray  # [B] int64
    speed: np.ndarray       # [B]
    oa: np.ndarray          # [B, T] warm start accel
    od: np.ndarray          # [B, T] warm start steer


def make_ego_batch(routes: List[np.ndarray], B: int, T: int, seed: int = 0,
                   truncate: bool = False, near_end_frac: float = 0.1, dt: float = DT,
                   dl: float = DL, max_dsteer: float = math.radians(30.0)) -> EgoBatch:
    """Random-init egos near their routes (routes must already be yaw-smoothed)."""
    rng = np.random.default_rng(seed)
    n_routes = len(routes)
    x0 = np.zeros((B, 4))
    path_id = rng.integers(0, n_routes, size=B).astype(np.int32)
    path_len = np.zeros(B, dtype=np.int32)
    target_ind = np.zeros(B, dtype=np.int64)
    reach = int(math.ceil(T * SPEED * dt / dl))
    for b in range(B):
        r = routes[path_id[b]]
        M = r.shape[0]
        if rng.random() < near_end_frac:
            s = int(rng.integers(max(M - reach, 0), M - 3))
        else:
            s = int(rng.integers(0, max(M - 1 - reach, 1)))
        lat = rng.normal(0.0, 0.3)
        dyaw = rng.normal(0.0, 0.1)
        yaw = r[s, 2]
        x0[b, 0] = r[s, 0] - lat * math.sin(yaw)
        x0[b, 1] = r[s, 1] + lat * math.cos(yaw)
        x0[b, 2] = rng.uniform(0.0, SPEED)
        x0[b, 3] = yaw + dyaw
        # the controller's remembered index trails the true nearest point a little
        target_ind[b] = max(s - int(rng.integers(0, 12)), 0)
        if truncate:
            path_len[b] = int(rng.integers(min(s + 4, M), M + 1))
        else:
            path_len[b] = M
    oa = rng.uniform(-1.0, 2.0, size=(B, T))
    od = np.zeros((B, T))
    od[:, 0] = np.clip(rng.normal(0.0, 0.05, size=B), -0.3, 0.3)
    lim = max_dsteer * dt
    for t in range(1, T):
        od[:, t] = od[:, t - 1] + np.clip(rng.normal(0.0, 0.03, size=B), -lim, lim)
    speed = np.full(B, SPEED)
    return EgoBatch(x0=x0, path_id=path_id, path_len=path_len, target_ind=target_ind, speed=speed,
                    oa=oa, od=od)
