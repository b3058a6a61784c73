"""CPU restatement (numpy) of the reference's per-tick loop glue that PRODUCES the truncated path for the MPC
(SURVEY.md 8 row f1).  TEST INFRASTRUCTURE ONLY (see oracle/mpc_oracle.h).

Restates, with explicit index formulas instead of the reference's concatenate/reshape pipeline:
  main/scenarios/mpc_intersection.py:104-140   progress index, ego path resampling, obstacle prediction,
                                               collision check, cut-off with margin
  main/lib/trajectories.py:58-86               resample_curve
  main/lib/trajectories.py:11-55               circle-centre trajectories of a car
  main/lib/moving_obstacles_prediction.py:21-47 constant-acceleration / constant-steer obstacle prediction
  main/lib/collision_avoidance.py:68-124       frame-offset copies, first colliding (frame, circle, copy, circle) row,
                                               first path point whose circle touches that obstacle circle
  main/lib/collision_avoidance.py:168-180      first path index within 1 mm of the hit
  main/lib/car_dimensions.py:62-79,82-90       circle radius / centres of BicycleModelDimensions
Pinned by golden vectors produced with the reference's own functions (tests/golden/make_golden_loop.py).
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

import oracle_py as O

DT = 0.2
TIME_HORIZON = 7.0
FRAME_WINDOW = 10
MAX_SPEED = 30.0 / 3.6
MAX_ACCEL = 2.0


def car_circles(L: float = 2.86, width: float = 2.0, extra_length: float = 0.64):
    """BicycleModelDimensions: bounding box (2.0, L + 0.64); radius = width / sqrt(2); two circle centres on the
    body axis at L/2 +- (length/2 - width/2) from the rear axle."""
    length = L + extra_length
    offset = length / 2 - width / 2
    c = L / 2
    return width / (2 ** .5), (c + offset, c - offset)


def circle_centres(xyyaw: np.ndarray, x_off: float) -> np.ndarray:
    """(n, 2) world positions of one body-axis circle: (cos(yaw) * x_off - sin(yaw) * 0.0) + x, ..."""
    th = xyyaw[:, 2]
    return np.stack([np.cos(th) * x_off - np.sin(th) * 0.0 + xyyaw[:, 0],
                     np.sin(th) * x_off + np.cos(th) * 0.0 + xyyaw[:, 1]], axis=1)


def resample_mask(points_xy: np.ndarray, dl) -> np.ndarray:
    """lib/trajectories.py:58-86: keep point i when floor(cumdist_i / dl_i) steps up; first and last always kept."""
    n = len(points_xy)
    step = np.zeros(n)
    d = points_xy[1:] - points_xy[:-1]
    step[1:] = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
    cum = np.cumsum(step)                       # sequential accumulate
    k = np.floor(cum / dl).astype(np.int64)
    mask = np.ones(n, dtype=bool)
    mask[1:] = (k[1:] - k[:-1]) >= 1
    mask[-1] = True
    return mask


def ego_resample_dl(n: int, v: float, dt: float = DT) -> np.ndarray:
    """mpc_intersection.py:113-120: the ego is predicted to accelerate at MAX_ACCEL up to MAX_SPEED."""
    if v < MAX_SPEED:
        acc = np.cumsum(np.zeros(n) + MAX_ACCEL) + v
        return dt * np.minimum(acc, MAX_SPEED)
    return np.full(n, dt * MAX_SPEED)


def predict_obstacle(x, y, v, yaw, a, steer, dt: float = DT, L: float = 2.86, horizon: float = TIME_HORIZON):
    """moving_obstacles_prediction.py:21-47 -> (n, 3) [x, y, yaw]; note yaw uses the UPDATED speed."""
    n = len(np.arange(0, horizon, dt))
    out = np.zeros((n, 3))
    for i in range(n):
        x += v * math.cos(yaw) * dt
        y += v * math.sin(yaw) * dt
        v += a * dt
        yaw += (v / L) * math.tan(steer) * dt
        out[i] = (x, y, yaw)
    return out


def _pair_geometry(L: float, obst_dims):
    """(min_distance, ego circle offsets, obstacle circle offsets): two cars (collision_avoidance.py:95), or a car and an
    obstacle of another shape, obst_dims = (L, width, extra_length) -- check_collision_moving_bicycle, :137."""
    radius, offs = car_circles(L)
    if obst_dims is None:
        return 2 * radius, offs, offs
    orad, ooffs = car_circles(*obst_dims)
    return radius + orad, offs, ooffs


def first_collision(res: np.ndarray, detailed: np.ndarray, preds: Sequence[np.ndarray], L: float = 2.86,
                    frame_window: int = FRAME_WINDOW, obst_dims=None) -> Optional[Tuple[float, float, int]]:
    """collision_avoidance.py:85-124.  res: resampled ego path (n_res, 3); detailed: the ego path from its progress
    index (n, 3); preds: predicted obstacle trajectories (each (P, 3)).
    Row order of the reference's pair table: frame f, then ego circle a, then obstacle copy c = o * (2w+1) + (off + w)
    with off = -w..w, then obstacle circle b.  Copy (o, off) at frame f shows sample clamp(min(f, P-1) - off, 0, P-1)."""
    if len(preds) == 0:
        return None
    thr, offs, ooffs = _pair_geometry(L, obst_dims)
    P = len(preds[0])
    F = max(len(res), max(len(p) for p in preds))
    ego_cc = [circle_centres(res, xo) for xo in offs]
    obs_cc = [[circle_centres(p, xo) for xo in ooffs] for p in preds]
    hit = None
    for f in range(F):
        fa = min(f, len(res) - 1)
        fo = min(f, P - 1)
        for a in range(2):
            pa = ego_cc[a][fa]
            for o in range(len(preds)):
                for off in range(-frame_window, frame_window + 1):
                    j = min(max(fo - off, 0), P - 1)
                    for b in range(2):
                        po = obs_cc[o][b][j]
                        dx, dy = pa[0] - po[0], pa[1] - po[1]
                        if math.sqrt(dx * dx + dy * dy) <= thr:
                            hit = po
                            break
                    if hit is not None:
                        break
                if hit is not None:
                    break
            if hit is not None:
                break
        if hit is not None:
            break
    if hit is None:
        return None
    # first point of the detailed path whose front circle (then rear circle) touches that obstacle circle
    n = len(detailed)
    first = 0  # np.argmax of an all-False mask is 0
    found = False
    for a in range(2):
        cc = circle_centres(detailed, offs[a])
        dx, dy = hit[0] - cc[:, 0], hit[1] - cc[:, 1]
        m = np.sqrt(dx * dx + dy * dy) <= thr
        if m.any():
            first = int(np.argmax(m))
            found = True
            break
    _ = found
    return float(detailed[first, 0]), float(detailed[first, 1]), first


def first_collision_fast(res, detailed, preds, L=2.86, frame_window=FRAME_WINDOW, obst_dims=None):
    """Vectorised form of first_collision (same row order), used for the bulk of the tests."""
    if len(preds) == 0:
        return None
    thr, offs, ooffs = _pair_geometry(L, obst_dims)
    P = len(preds[0])
    F = max(len(res), max(len(p) for p in preds))
    w = frame_window
    fa = np.minimum(np.arange(F), len(res) - 1)
    fo = np.minimum(np.arange(F), P - 1)
    offv = np.arange(-w, w + 1)
    j = np.clip(fo[:, None] - offv[None, :], 0, P - 1)                       # [F, 2w+1]
    ego = np.stack([circle_centres(res, xo)[fa] for xo in offs], axis=1)      # [F, a, 2]
    obs = np.stack([np.stack([circle_centres(p, xo)[j] for xo in ooffs], axis=2) for p in preds], axis=1)  # [F, o, off, b, 2]
    d = ego[:, :, None, None, None, :] - obs[:, None, :, :, :, :]             # [F, a, o, off, b, 2]
    m = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) <= thr
    flat = m.reshape(-1)
    k = int(np.argmax(flat))
    if not flat[k]:
        return None
    f, a, o, io, b = np.unravel_index(k, m.shape)
    hit = obs[f, o, io, b]
    n = len(detailed)
    cc = np.concatenate([circle_centres(detailed, xo) for xo in offs])
    dx, dy = hit[0] - cc[:, 0], hit[1] - cc[:, 1]
    mm = np.sqrt(dx * dx + dy * dy) <= thr
    first = int(np.argmax(mm)) % n
    return float(detailed[first, 0]), float(detailed[first, 1]), first


def cutoff_index(full: np.ndarray, x: float, y: float, radius: float = 0.001) -> Optional[int]:
    """collision_avoidance.py:168-180 (returns None where the reference returns the array: 'no cut-off')."""
    dx, dy = full[:, 0] - x, full[:, 1] - y
    m = np.sqrt(dx * dx + dy * dy) <= radius
    k = int(np.argmax(m))
    return k if m[k] else None


def extra_cutoff_margin(dl: float, L: float = 2.86) -> int:
    radius, _ = car_circles(L)
    return 4 * int(math.ceil(radius / dl))       # mpc_intersection.py:88-89


def loop_pre_tick(state_xyyawv, traj_agent_idx: int, prev_path_len: Optional[int], full: np.ndarray,
                  obstacles: Sequence[Sequence[float]], dl: float, L: float = 2.86, dt: float = DT, obst_dims=None,
                  margin_factor: int = 4, frame_window: int = FRAME_WINDOW):
    """mpc_intersection.py:104-140 for one ego: returns (status, traj_agent_idx, path_len, collision_xy or None).
    obstacles: (x, y, v, yaw, a, steer) tuples as MovingObstacle*.get() returns them.  obst_dims = (L, width,
    extra_length) of obstacles shaped unlike the ego and margin_factor = 2: the same glue in
    scenarios/overtaking_cyclist_bidirectional_road.py:94-95,122-133,221-240."""
    x, y, yaw, v = state_xyyawv
    M = len(full)
    # :106-109  (rows of a path are distinct points, so "row differs from the last row" == "index differs")
    if prev_path_len is None or traj_agent_idx != prev_path_len - 1:
        st, idx = O.nearest_index_in_direction(x, y, full[:, 0], full[:, 1], traj_agent_idx, True)
        if st != 0:
            return st, traj_agent_idx, prev_path_len if prev_path_len is not None else M, None
        traj_agent_idx = idx
    detailed = full[traj_agent_idx:]
    res = detailed[resample_mask(detailed[:, :2], ego_resample_dl(len(detailed), v, dt))]
    preds = [predict_obstacle(*o, dt=dt, L=L if obst_dims is None else obst_dims[0]) for o in obstacles]
    col = first_collision_fast(res, detailed, preds, L=L, frame_window=frame_window, obst_dims=obst_dims)
    if col is None:
        return 0, traj_agent_idx, M, None
    c = cutoff_index(full, col[0], col[1])
    assert c is not None
    cut = max(traj_agent_idx + 1, c - (extra_cutoff_margin(dl, L) // 4) * margin_factor)
    return 0, traj_agent_idx, cut, (col[0], col[1])
