#!/usr/bin/env python3
"""Golden vectors for the scripted obstacle vehicles: the reference's own MovingObstacleTIntersection
(main/lib/moving_obstacles.py:166-232) stepped 140 times; `get()` recorded before every `step()`.  Build container only."""
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
    from lib.moving_obstacles import MovingObstacleTIntersection
    car = BicycleModelDimensions(skip_back_circle_collision_checking=False)
    specs = [dict(direction=1, offset=2., turning=False, speed=25 / 3.6), dict(direction=-1, offset=4., turning=True, speed=25 / 3.6),
             dict(direction=1, offset=None, turning=True, speed=20 / 3.6), dict(direction=-1, offset=0., turning=False, speed=30 / 3.6)]
    obs = [MovingObstacleTIntersection(car, dt=0.2, **s) for s in specs]
    rec = []
    for k in range(140):
        rec.append([list(o.get()) for o in obs])
        for o in obs:
            o.step()
    np.savez(os.path.join(HERE, "obstacles_scripted.npz"), get=np.array(rec, dtype=np.float64),
             direction=np.array([s["direction"] for s in specs]), turning=np.array([s["turning"] for s in specs]),
             speed=np.array([s["speed"] for s in specs]), offset=np.array([-1.0 if s["offset"] is None else s["offset"] for s in specs]))
    print("obstacles_scripted.npz", np.array(rec).shape)


if __name__ == "__main__":
    main()
