#!/usr/bin/env python3
"""Golden vectors for the loop glue with an obstacle of another shape than the ego -- the cyclist of
main/scenarios/overtaking_cyclist_bidirectional_road.py:94-95,122-133,221-240 -- produced by the REFERENCE's own functions
(build container only): lib.trajectories.resample_curve, lib.moving_obstacles_prediction.MovingObstaclesPrediction with
car_dimensions = BicycleRealDimensions, lib.collision_avoidance.check_collision_moving_bicycle /
get_cutoff_curve_by_position_idx, lib.car_dimensions.*.  No pickle is loaded (synthetic routes), cvxpy is not needed."""
import importlib.util
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_MAIN = "/root/reference/main"


def main():
    if not os.path.isdir(REF_MAIN):
        raise SystemExit("reference not present")
    sys.path.insert(0, REF_MAIN)
    import matplotlib
    matplotlib.use("Agg")
    from lib.car_dimensions import BicycleModelDimensions, BicycleRealDimensions
    from lib.collision_avoidance import check_collision_moving_bicycle, get_cutoff_curve_by_position_idx
    from lib.moving_obstacles_prediction import MovingObstaclesPrediction
    from lib.simulation import Simulation
    from lib.trajectories import resample_curve
    spec = importlib.util.spec_from_file_location("jsim_synth", os.path.join(REPO, "av-simulation-at-intersections_amd", "synth.py"))
    S = importlib.util.module_from_spec(spec); sys.modules["jsim_synth"] = S; spec.loader.exec_module(S)
    car = BicycleModelDimensions(skip_back_circle_collision_checking=False)
    bike = BicycleRealDimensions(skip_back_circle_collision_checking=False)
    rng = np.random.default_rng(91)
    routes = S.make_route_table()
    for r in routes:
        S.smooth_yaw_inplace(r[:, 2])
    DT, TIME_HORIZON, FRAME_WINDOW, MAX_ACCEL = 0.2, 7.0, 10, 2.0
    dl = float(np.linalg.norm(routes[0][0, :2] - routes[0][1, :2]))
    MARGIN = 2 * int(math.ceil(car.radius / dl))                       # overtaking_cyclist_bidirectional_road.py:94-95
    N, NOBS = 120, 2
    rec = dict(route=[], idx=[], v=[], obst=[], pred=[], col=[], cutoff=[])
    for k in range(N):
        rid = int(rng.integers(0, len(routes)))
        full = routes[rid]
        idx = int(rng.integers(0, len(full) - 5))
        v = float(rng.choice([0.0, rng.uniform(0, 8.3), 30 / 3.6]))
        traj = full[idx:]
        if v < Simulation.MAX_SPEED:
            rdl = np.cumsum(np.zeros((traj.shape[0],)) + MAX_ACCEL) + v
            res = resample_curve(traj, dl=DT * np.minimum(rdl, Simulation.MAX_SPEED))
        else:
            res = resample_curve(traj, dl=DT * Simulation.MAX_SPEED)
        obst = []
        for o in range(NOBS):                                           # slow cyclists near the path ahead
            j = int(min(idx + rng.integers(10, 300), len(full) - 1))
            ang = full[j, 2] + rng.normal(0, 0.3)
            dist = rng.uniform(0, 12)
            ox, oy = full[j, 0] - dist * math.cos(ang) + rng.normal(0, 4.0), full[j, 1] - dist * math.sin(ang) + rng.normal(0, 4.0)
            obst.append((ox, oy, rng.uniform(0, 5), ang, 0.0, rng.choice([0.0, 0.1, -0.15])))
        preds = [np.vstack(MovingObstaclesPrediction(*o, sample_time=DT, car_dimensions=bike).state_prediction(TIME_HORIZON)).T
                 for o in obst]
        col = check_collision_moving_bicycle(car, bike, res, traj, preds, frame_window=FRAME_WINDOW)
        if col is not None:
            cut = get_cutoff_curve_by_position_idx(full, col[0], col[1])
            assert isinstance(cut, (int, np.integer))
            cutoff = max(idx + 1, int(cut) - MARGIN)
            colrow = (1.0, col[0], col[1], float(col[2]))
        else:
            cutoff, colrow = len(full), (0.0, 0.0, 0.0, -1.0)
        rec["route"].append(rid); rec["idx"].append(idx); rec["v"].append(v); rec["obst"].append(obst)
        rec["pred"].append(np.stack([p[:, :3] for p in preds])); rec["col"].append(colrow); rec["cutoff"].append(cutoff)
    np.savez_compressed(os.path.join(HERE, "loop_bicycle.npz"), route=np.array(rec["route"]), idx=np.array(rec["idx"]),
                        v=np.array(rec["v"]), obst=np.array(rec["obst"]), pred=np.array(rec["pred"]), col=np.array(rec["col"]),
                        cutoff=np.array(rec["cutoff"]), margin=np.array(MARGIN), car_radius=np.array(car.radius),
                        bike_radius=np.array(bike.radius), bike_circle_centers=np.array(bike.circle_centers),
                        bike_L=np.array(bike.distance_back_to_front_wheel), bike_box=np.array(bike.bounding_box_size), dl=np.array(dl))
    print("loop_bicycle:", N, "cases,", int(np.array(rec["col"])[:, 0].sum()), "with a collision; bike radius", bike.radius,
          "centres", bike.circle_centers)


if __name__ == "__main__":
    main()
