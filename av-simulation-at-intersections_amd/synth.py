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


def make_route_table(multi_lane: bool = False) -> List[np.ndarray]:
    """The 12 routes (4 arms x 3 turns).  multi_lane=True shifts the lane offset (second lane of a
    two-lane arm, lane width 3.5 m) to emulate mpc_intersection_multi_lane geometry."""
    off = 3.0 + 3.5 if multi_lane else 3.0
    arm = 40.0 if multi_lane else 30.0
    return [make_route(sp, tn, lane_offset=off, arm=arm) for sp in (1, 2, 3, 4) for tn in (1, 2, 3)]


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
