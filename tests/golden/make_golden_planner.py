#!/usr/bin/env python3
"""G1 fixtures for the route planner (row f4): the REFERENCE's own MotionPrimitiveSearch
(main/lib/mp_search_ww_generic.py) run here on its own scenarios -- the 12 routes of envs/intersection.py (4 arms x 3 turns) and
6 routes of envs/intersection_multi_lanes.py with number_of_lanes = 2 -- and written to tests/golden/planner.npz:
obstacle half-planes (with the margin the scenario scripts use, car_dimensions.radius), start, goal point, goal box, angle
tolerance  ->  cost, path nodes, primitive sequence, trajectory (n_primitives * 60, 3).

The motion primitives are REGENERATED from the reference's recipe with the reference's own Bicycle class
(main/create_motion_primitives_bicycle_model.py:12-27: Bicycle.step from the origin at 8.3 m/s, nine steering angles, 61
states 0.01 s apart); the pickled primitives shipped under main/data/ are never loaded (serialized files of the reference).
The oracle's restatement (oracle/planner_oracle.py) is checked against every route right here: identical node tuples,
identical primitive sequence, bit-identical trajectory.

usage (needs /root/reference; from the repo root):  python tests/golden/make_golden_planner.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("JSIM_REFERENCE", "/root/reference/main")
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REF)
os.environ.setdefault("MPLBACKEND", "Agg")

from bicycle.main import Bicycle                                    # noqa: E402  (reference)
from envs.intersection import intersection                          # noqa: E402
from envs.intersection_multi_lanes import intersection as intersection_ml   # noqa: E402
from lib.car_dimensions import BicycleModelDimensions               # noqa: E402
from lib.motion_primitive import MotionPrimitive                    # noqa: E402
from lib.mp_search_ww_generic import MotionPrimitiveSearch          # noqa: E402

import planner_oracle as PO                                         # noqa: E402


def reference_primitives(car):
    mps = {}
    for name, delta in zip(PO.MP_NAMES, PO.MP_STEER):
        model = Bicycle(car_dimensions=car, sample_time=0.01)
        pts = []
        for _ in range(61):
            pts.append(np.array([model.xc, model.yc, model.theta]))
            model.step(8.3, delta)
        pts = np.array(pts, dtype=np.float64)
        mp = MotionPrimitive(name=name, forward_speed=8.3, steering_angle=delta, n_seconds=0.3)
        mp.total_length = np.linalg.norm(pts[:-1, :2] - pts[1:, :2], axis=1).sum()
        mp.points = pts
        mps[name] = mp
    return mps


def main():
    car = BicycleModelDimensions(skip_back_circle_collision_checking=False)
    mps = reference_primitives(car)
    mine = PO.make_motion_primitives()
    for (name, pts, total) in mine:      # the restated recipe equals the reference classes' output bit for bit
        assert np.array_equal(pts, mps[name].points) and total == mps[name].total_length, name
    scen = [("single", sp, tn, 0, 0, intersection(start_pos=sp, turn_indicator=tn)) for sp in (1, 2, 3, 4) for tn in (1, 2, 3)]
    scen += [("multi", 1, tn, sl, gl, intersection_ml(start_pos=1, turn_indicator=tn, start_lane=sl, goal_lane=gl, number_of_lanes=2))
             for tn in (1, 2, 3) for (sl, gl) in ((1, 1), (2, 2))]
    out = {"n_routes": np.int64(len(scen)), "mp_points": np.array([mps[n].points for n in PO.MP_NAMES]),
           "mp_length": np.array([mps[n].total_length for n in PO.MP_NAMES]), "radius": np.float64(car.radius),
           "circle_centers": np.array(car.circle_centers)}
    for i, (kind, sp, tn, sl, gl, sc) in enumerate(scen):
        search = MotionPrimitiveSearch(sc, car, mps, margin=car.radius)
        cost, path, traj = search.run(debug=True)
        names = [search._points_to_mp_names[a, b] for a, b in zip(path[:-1], path[1:])]
        hps = [np.asarray(h, dtype=np.float64) for h in search._obstacles_hp]
        ga = sc.goal_area
        orc = PO.PlannerOracle(sc.start, sc.goal_point, (ga.xy1[0], ga.xy1[1], ga.xy2[0], ga.xy2[1]), sc.allowed_goal_theta_difference,
                               hps, mine, car.circle_centers, car.radius)
        c2, p2, t2 = orc.run()
        assert c2 == cost and p2 == path and np.array_equal(t2, traj), (kind, sp, tn)
        assert [PO.MP_NAMES[k] for k in orc.prim_sequence(p2)] == names
        assert orc.n_expanded == len(search.debug_data)
        hp_flat = np.concatenate(hps, axis=0)
        hp_off = np.concatenate([[0], np.cumsum([len(h) for h in hps])]).astype(np.int64)
        out[f"r{i}_meta"] = np.array([{"single": 0, "multi": 1}[kind], sp, tn, sl, gl], dtype=np.int64)
        out[f"r{i}_hp"] = hp_flat
        out[f"r{i}_hp_off"] = hp_off
        out[f"r{i}_start"] = np.array(sc.start, dtype=np.float64)
        out[f"r{i}_goal"] = np.array(sc.goal_point, dtype=np.float64)
        out[f"r{i}_goal_box"] = np.array([ga.xy1[0], ga.xy1[1], ga.xy2[0], ga.xy2[1]], dtype=np.float64)
        out[f"r{i}_tol"] = np.float64(sc.allowed_goal_theta_difference)
        out[f"r{i}_cost"] = np.float64(cost)
        out[f"r{i}_path"] = np.array(path, dtype=np.float64)
        out[f"r{i}_prims"] = np.array([PO.MP_NAMES.index(n) for n in names], dtype=np.int32)
        out[f"r{i}_traj"] = np.asarray(traj, dtype=np.float64)
        out[f"r{i}_n_expanded"] = np.int64(orc.n_expanded)
        out[f"r{i}_max_open"] = np.int64(orc.max_open)
        print(f"route {i:2d} {kind:6s} start {sp} turn {tn} lanes {sl}->{gl}: cost {cost:8.3f}, {len(path) - 1:2d} primitives, "
              f"trajectory {traj.shape}, {orc.n_expanded} expansions, open list up to {orc.max_open}, {len(hps)} obstacles")
    np.savez_compressed(os.path.join(HERE, "planner.npz"), **out)


if __name__ == "__main__":
    main()
