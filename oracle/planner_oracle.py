"""CPU restatement of the reference's route planner -- A* over motion primitives.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by tests/golden/make_golden_planner.py, which pins it against the
reference's own classes); never by the product package.

Follows, function by function:
  main/lib/mp_search_ww_generic.py:27-58   MotionPrimitiveSearch.__init__ (weights, obstacle half-planes with margin)
                              :60-82      calculate_steering_change_cost
                              :118-136    _create_collision_points (resample_curve at dl = radius, both circle trajectories)
                              :150-155    is_goal
                              :166-190    distance_to_goal (the heuristic)
                              :202-243    neighbor_function (collision check of every primitive, edge cost)
                              :245-257    path_to_full_trajectory
  main/lib/a_star.py:31-78                 AStar.run (binary heap of (g + h, g, node, predecessor), best-predecessor dict)
  main/lib/obstacles.py:80-93,127-142,157-176   BoxObstacle.to_convex / CircleObstacle.to_convex / check_collision
  main/lib/linalg.py:4-25,28-57            create_2d_transform_mtx / transform_2d_pts (2x2 rotation-only matrix when x == y == 0)
  main/lib/maths.py:4-10                   normalize_angle
  main/lib/trajectories.py:11-55,58-86     collision-circle trajectories, resample_curve
  main/create_motion_primitives_bicycle_model.py:12-27 + main/bicycle/main.py:28-41   the primitives' recipe (explicit Euler of the
                                           kinematic bicycle at 8.3 m/s, nine steering angles, 61 points 0.083 m apart) --
                                           regenerated from the recipe; the reference's pickled primitives are never loaded.
Only the default weights' terms carry weight in the reference's scenarios (wh_obstacle = wh_center = wc_center = 0, and the
edge's obstacle term is guarded by the HEURISTIC's obstacle weight, mp_search_ww_generic.py:230, so it is 0 too); the
obstacle / centre terms are restated all the same."""
from __future__ import annotations

import heapq
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

MP_NAMES = ("straight", "left1", "left2", "left3", "left4", "right1", "right2", "right3", "right4")
MP_STEER = (0.0, 0.1, 0.2, 0.3, 0.4, -0.1, -0.2, -0.3, -0.4)   # main/create_motion_primitives_prius.py:19-29


def make_motion_primitives(L: float = 2.86, v: float = 8.3, n_steps: int = 60, dt: float = 0.01):
    """[(name, points (61, 3), total_length)] -- Bicycle.step (main/bicycle/main.py:28-41) from the origin, state recorded
    BEFORE each step (create_motion_primitives_bicycle_model.py:21-23)."""
    out = []
    for name, delta in zip(MP_NAMES, MP_STEER):
        x = y = th = 0.0
        pts = []
        for _ in range(n_steps + 1):
            pts.append((x, y, th))
            xd = v * np.cos(th); yd = v * np.sin(th); thd = (v / L) * np.tan(delta)
            x += xd * dt; y += yd * dt; th += thd * dt
        pts = np.array(pts, dtype=np.float64)
        total = float(np.linalg.norm(pts[:-1, :2] - pts[1:, :2], axis=1).sum())
        out.append((name, pts, total))
    return out


def box_halfplanes(xy_width, xy_center, margin):
    cx, cy = xy_center
    w, h = xy_width
    x1, y1, x2, y2 = cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2
    return np.array([[1, 0, -(x2 + margin)], [-1, 0, x1 - margin], [0, 1, -(y2 + margin)], [0, -1, y1 - margin]], dtype=np.float64)


def circle_halfplanes(radius, xy_center, margin):
    cx, cy = xy_center
    r = radius
    return np.array([[1, 0, -(cx + r + margin)], [-1, 0, cx - r - margin], [0, 1, -(cy + r + margin)], [0, -1, cy - r - margin],
                     [-1, 1, cx - cy - r * np.sqrt(2) - 2 * margin], [1, -1, -cx + cy - r * np.sqrt(2) - 2 * margin],
                     [-1, -1, cx + cy - r * np.sqrt(2) - 2 * margin], [1, 1, -cx - cy - r * np.sqrt(2) - 2 * margin]], dtype=np.float64)


def resample_curve(points, dl, keep_last_point=True):
    step = np.linalg.norm(points[1:, :2] - points[:-1, :2], axis=1)
    step = np.append(0.0, step)
    k = np.floor(step.cumsum() / dl).astype(int)
    mask = np.append(True, (k[1:] - k[:-1]) >= 1.0)
    if keep_last_point:
        mask[-1] = True
    return points[mask].copy()


def collision_points(mp_points, circle_centers, radius):
    pts = resample_curve(mp_points.copy(), dl=radius, keep_last_point=True)
    th = pts[:, 2]
    out = []
    for cx, cy in circle_centers:
        off = np.vstack([np.cos(th) * cx - np.sin(th) * cy, np.sin(th) * cx + np.cos(th) * cy]).T + pts[:, :2]
        out.append(np.append(off, np.atleast_2d(th).T, axis=1))
    return np.concatenate(out, axis=0)


def transform_mtx(x, y, theta):
    if x == 0 and y == 0:
        return np.array([[np.cos(theta), -np.sin(theta)], [np.sin(theta), np.cos(theta)]])
    return np.array([[np.cos(theta), -np.sin(theta), x], [np.sin(theta), np.cos(theta), y], [0, 0, 1]])


def transform_pts(theta, mtx, points):
    p = points[:, :2]
    if mtx.shape == (2, 2):
        p = p @ mtx.T
    else:
        p = (np.append(p, np.ones((points.shape[0], 1)), axis=1) @ mtx.T)[:, :2]
    if points.shape[1] == 3:
        return np.append(p, points[:, 2:] + theta, axis=1)
    return p


def normalize_angle(theta):
    theta = theta % math.tau
    if theta >= math.pi:
        theta -= math.tau
    return theta


def check_collision(hp, pts_xy):
    pts = np.vstack([pts_xy, np.ones((pts_xy.shape[1],))])
    return bool(np.any(np.all((hp @ pts) <= 0, axis=0)))


class PlannerOracle:
    def __init__(self, start, goal_point, goal_box, allowed_theta, obstacles_hp: Sequence[np.ndarray], mps, circle_centers, radius,
                 wh=(1.0, 2.7, 15.0, 0.0, 0.0), wc=(1.0, 5.0, 0.1, 0.0)):
        self.start = tuple(float(v) for v in start)
        self.goal = tuple(float(v) for v in goal_point)
        self.gx1, self.gy1, self.gx2, self.gy2 = goal_box       # BoxObstacle xy1 / xy2 of the goal area
        self.tol = float(allowed_theta)
        self.hp = [np.asarray(h, dtype=np.float64) for h in obstacles_hp]
        self.mps = mps
        self.cc = [collision_points(p, circle_centers, radius) for _, p, _ in mps]
        self.wh_dist, self.wh_theta, self.wh_steer, self.wh_obst, self.wh_center = wh
        self.wc_dist, self.wc_steer, self.wc_obst, self.wc_center = wc
        self.edge_mp: Dict[Tuple, int] = {}
        self.n_expanded = 0
        self.max_open = 0

    @staticmethod
    def steering_change(a, b):
        d = b[2] - a[2]
        d = (d + np.pi) % (2 * np.pi) - np.pi
        return abs(d) * 1.0

    def dist_obstacle(self, node):
        x0, y0 = node[0], node[1]
        best = float("inf")
        for hp in self.hp:
            d = min(abs(a * x0 + b * y0 + c) / (a ** 2 + b ** 2) ** 0.5 for a, b, c in hp)
            best = min(best, d)
        return best

    def is_goal(self, node):
        x, y, th = node
        dx = max(self.gx1 - x, 0, x - self.gx2)
        dy = max(self.gy1 - y, 0, y - self.gy2)
        return bool(np.sqrt(dx * dx + dy * dy) <= 1e-5 and abs(th - self.goal[2]) <= self.tol)

    def heuristic(self, node):
        x, y, th = node
        gx, gy, gth = self.goal
        dxy = np.sqrt((x - gx) ** 2 + (y - gy) ** 2)
        dth = min(abs(th - gth), abs(th - gth) - self.tol / 2)
        steer = self.steering_change(node, self.goal)
        obst = 0.0
        center = 0.0
        if self.wh_obst != 0.0:
            d = self.dist_obstacle(node)
            obst = 1 / d if d else float("inf")
        if self.wh_center != 0.0:
            center = np.sqrt(x ** 2 + y ** 2)
        return self.wh_dist * dxy + self.wh_theta * dth + self.wh_steer * steer + self.wh_obst * obst + self.wh_center * center

    def neighbors(self, node):
        mtx = transform_mtx(*node)
        for k, (name, pts, total) in enumerate(self.mps):
            ccp = transform_pts(node[2], mtx, self.cc[k])
            xy = ccp[:, :2].T
            if any(check_collision(o, xy) for o in self.hp):
                continue
            x, y, th = tuple(np.squeeze(transform_pts(node[2], mtx, np.atleast_2d(pts[-1]))).tolist())
            nb = (x, y, normalize_angle(th))
            self.edge_mp[(node, nb)] = k
            steer = self.steering_change(node, nb)
            obst = 0.0
            center = 0.0
            if self.wh_obst != 0.0:          # (sic: the heuristic's weight guards the edge term, mp_search_ww_generic.py:230)
                d = self.dist_obstacle(nb)
                obst = 1 / d if d else float("inf")
            if self.wc_center != 0.0:
                center = np.linalg.norm([x, y])
            yield self.wc_dist * total + self.wc_steer * steer + self.wc_obst * obst + self.wc_center * center, nb

    def run(self, max_expansions=200000):
        start = self.start
        q = [(0, 0, start, start)]
        pred = {}
        while q:
            self.max_open = max(self.max_open, len(q))
            gh, g, node, p = heapq.heappop(q)
            if node in pred and g >= pred[node][0]:
                continue
            pred[node] = (g, p)
            self.n_expanded += 1
            if self.n_expanded > max_expansions:
                raise RuntimeError("expansion budget exhausted")
            if self.is_goal(node):
                path = [node]
                while node != start:
                    path.append(p)
                    node, p = p, pred[p][1]
                path.reverse()
                return g, path, self.trajectory(path)
            for ev, nb in self.neighbors(node):
                ng = g + ev
                if nb not in pred or ng < pred[nb][0]:
                    heapq.heappush(q, (ng + self.heuristic(nb), ng, nb, node))
        raise Exception("No solution found.")

    def prim_sequence(self, path):
        return [self.edge_mp[(a, b)] for a, b in zip(path[:-1], path[1:])]

    def trajectory(self, path):
        segs = []
        for a, b in zip(path[:-1], path[1:]):
            k = self.edge_mp[(a, b)]
            segs.append(transform_pts(a[2], transform_mtx(*a), self.mps[k][1])[:-1])
        return np.concatenate(segs, axis=0)
