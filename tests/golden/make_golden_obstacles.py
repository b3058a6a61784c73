#!/usr/bin/env python3
"""Golden vectors for the scripted obstacle vehicles: the reference's own MovingObstacleTIntersection / MovingObstacleRoundabout /
MovingObstacleArterial (main/lib/moving_obstacles.py) stepped 140 times; `get()` recorded before every `step()`, exactly as the
scenario loops call them.  Build container only."""
import contextlib
import io
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_MAIN = "/root/reference/main"


def main():
    if not os.path.isdir(REF_MAIN):
        raise SystemExit("reference not present")
    sys.path.insert(0, REF_MAIN)
    import matplotlib
    matplotlib.use("Agg")
    from lib.car_dimensions import BicycleModelDimensions
    from lib.moving_obstacles import MovingObstacleArterial, MovingObstacleRoundabout, MovingObstacleTIntersection
    car = BicycleModelDimensions(skip_back_circle_collision_checking=False)
    specs = [dict(direction=1, offset=2., turning=False, speed=25 / 3.6), dict(direction=-1, offset=4., turning=True, speed=25 / 3.6),
             dict(direction=1, offset=None, turning=True, speed=20 / 3.6), dict(direction=-1, offset=0., turning=False, speed=30 / 3.6)]
    obs = [MovingObstacleTIntersection(car, dt=0.2, **s) for s in specs]
    rec = []
    for k in range(140):
        rec.append([list(o.get()) for o in obs])
        for o in obs:
            o.step()
    out = dict(get=np.array(rec, dtype=np.float64),
               direction=np.array([s["direction"] for s in specs]), turning=np.array([s["turning"] for s in specs]),
               speed=np.array([s["speed"] for s in specs]), offset=np.array([-1.0 if s["offset"] is None else s["offset"] for s in specs]))
    # roundabout (its steering property prints) and arterial vehicles
    rspecs = [dict(direction=1, turning=True, speed=20 / 3.6, offset=None), dict(direction=-1, turning=True, speed=25 / 3.6, offset=1.0),
              dict(direction=1, turning=False, speed=15 / 3.6, offset=2.0), dict(direction=-1, turning=True, speed=12 / 3.6, offset=None)]
    aspecs = [dict(x_init=1.5, y_init=-30.0, speed=25 / 3.6, initial_speed=5 / 3.6, offset=3.0),
              dict(x_init=-1.5, y_init=-10.0, speed=10 / 3.6, initial_speed=0.0, offset=None)]
    with contextlib.redirect_stdout(io.StringIO()):
        robs = [MovingObstacleRoundabout(car, dt=0.2, **s) for s in rspecs]
        aobs = [MovingObstacleArterial(car, dt=0.2, **s) for s in aspecs]
        rrec, arec = [], []
        for k in range(140):
            rrec.append([list(o.get()) for o in robs])
            arec.append([list(o.get()) for o in aobs])
            for o in robs + aobs:
                o.step()
    out.update(r_get=np.array(rrec, dtype=np.float64), r_direction=np.array([s["direction"] for s in rspecs]),
               r_turning=np.array([s["turning"] for s in rspecs]), r_speed=np.array([s["speed"] for s in rspecs]),
               r_offset=np.array([-1.0 if s["offset"] is None else s["offset"] for s in rspecs]),
               a_get=np.array(arec, dtype=np.float64), a_x=np.array([s["x_init"] for s in aspecs]), a_y=np.array([s["y_init"] for s in aspecs]),
               a_speed=np.array([s["speed"] for s in aspecs]), a_v0=np.array([s["initial_speed"] for s in aspecs]),
               a_offset=np.array([-1.0 if s["offset"] is None else s["offset"] for s in aspecs]))
    np.savez(os.path.join(HERE, "obstacles_scripted.npz"), **out)
    print("obstacles_scripted.npz", out["get"].shape, out["r_get"].shape, out["a_get"].shape,
          "roundabout steer values", np.unique(np.round(out["r_get"][:, :, 5], 6)))


if __name__ == "__main__":
    main()
