#!/usr/bin/env python3
"""More G1 fixtures for the route planner (row f4): the REFERENCE's MotionPrimitiveSearch run on its OTHER scenario
builders -- envs/roundabout.py (both sizes) and envs/t_intersection.py -- written to tests/golden/planner_envs.npz
(envs/free_area.py cannot be imported: it asks lib.obstacles for a name that does not exist; envs/arterial_multi_lanes.py
imports cvxpy, which is not installed and stays absent).  Nothing of those builders is restated in the
product: each route is stored as data (obstacle half-planes from the scenario objects' own to_convex(margin = car radius),
start, goal point, goal box, angle tolerance -> cost, node path, primitive sequence, trajectory, expansion count), which is
what the product's drop-in class receives when a scenario script hands it the reference's scenario object.
Motion primitives regenerated from the recipe as in make_golden_planner.py; no pickle is loaded.  The oracle's restatement
is checked against every route here.  A route whose reference search takes longer than `--budget` seconds is skipped.

usage (needs /root/reference; from the repo root):  python tests/golden/make_golden_planner_envs.py"""
import os
import signal
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("JSIM_REFERENCE", "/root/reference/main")
sys.path.insert(0, os.path.join(REPO, "oracle"))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)
os.environ.setdefault("MPLBACKEND", "Agg")

from envs.roundabout import roundabout                              # noqa: E402  (reference)
from envs.t_intersection import t_intersection                      # noqa: E402
from lib.car_dimensions import BicycleModelDimensions               # noqa: E402
from lib.mp_search_ww_generic import MotionPrimitiveSearch          # noqa: E402

import planner_oracle as PO                                         # noqa: E402
from make_golden_planner import reference_primitives                # noqa: E402

BUDGET = float(sys.argv[sys.argv.index("--budget") + 1]) if "--budget" in sys.argv else 120.0


class TooSlow(Exception):
    pass


def _alarm(*_):
    raise TooSlow()


def main():
    car = BicycleModelDimensions(skip_back_circle_collision_checking=False)
    mps = reference_primitives(car)
    mine = PO.make_motion_primitives()
    cand = []
    for size in ("normal", "big"):
        for sp in (1, 2, 3, 4):
            for tn in (1, 2, 3, 4):
                cand.append((f"roundabout/{size}/start{sp}/turn{tn}", lambda sp=sp, tn=tn, size=size: roundabout(start_pos=sp, turn_indicator=tn, size=size)))
    for sp in (1, 2, 3):
        for tn in (1, 2, 3):
            cand.append((f"t_intersection/start{sp}/turn{tn}", lambda sp=sp, tn=tn: t_intersection(turn_indicator=tn, start_pos=sp)))
    out = {"radius": np.float64(car.radius), "circle_centers": np.array(car.circle_centers)}
    names, n = [], 0
    signal.signal(signal.SIGALRM, _alarm)
    for label, build in cand:
        try:
            sc = build()
        except Exception as e:                      # a start / turn combination the builder does not define
            print(f"{label}: not built by the reference ({type(e).__name__}: {e})")
            continue
        t0 = time.time()
        try:
            signal.alarm(int(BUDGET))
            search = MotionPrimitiveSearch(sc, car, mps, margin=car.radius)
            cost, path, traj = search.run(debug=True)
            signal.alarm(0)
        except TooSlow:
            print(f"{label}: reference search over {BUDGET:.0f} s, skipped")
            continue
        except Exception as e:
            signal.alarm(0)
            if str(e) != "No solution found.":
                print(f"{label}: reference raised {type(e).__name__}: {e}")
                continue
            # the reference's open list ran empty (lib/a_star.py:78): a fixture too -- query, number of expansions, no route
            hps = [np.asarray(h, dtype=np.float64) for h in search._obstacles_hp]
            ga = sc.goal_area
            box = (ga.xy1[0], ga.xy1[1], ga.xy2[0], ga.xy2[1])
            orc = PO.PlannerOracle(sc.start, sc.goal_point, box, sc.allowed_goal_theta_difference, hps, mine, car.circle_centers, car.radius)
            try:
                orc.run(max_expansions=10 ** 7)
                raise AssertionError(f"{label}: the oracle found a route where the reference found none")
            except Exception as e2:
                assert str(e2) == "No solution found.", (label, e2)
            assert orc.n_expanded == len(search.debug_data), (label, orc.n_expanded, len(search.debug_data))
            i = n
            out[f"r{i}_hp"] = np.concatenate(hps, axis=0)
            out[f"r{i}_hp_off"] = np.concatenate([[0], np.cumsum([len(h) for h in hps])]).astype(np.int64)
            out[f"r{i}_start"] = np.array(sc.start, dtype=np.float64)
            out[f"r{i}_goal"] = np.array(sc.goal_point, dtype=np.float64)
            out[f"r{i}_goal_box"] = np.array(box, dtype=np.float64)
            out[f"r{i}_tol"] = np.float64(sc.allowed_goal_theta_difference)
            out[f"r{i}_cost"] = np.float64(np.nan)
            out[f"r{i}_path"] = np.zeros((0, 3)); out[f"r{i}_prims"] = np.zeros(0, dtype=np.int32); out[f"r{i}_traj"] = np.zeros((0, 3))
            out[f"r{i}_n_expanded"] = np.int64(orc.n_expanded)
            out[f"r{i}_max_open"] = np.int64(orc.max_open)
            names.append(label)
            n += 1
            print(f"route {i:2d} {label:34s}: NO SOLUTION after {orc.n_expanded} expansions (open list up to {orc.max_open}), "
                  f"{len(hps)} obstacles", flush=True)
            continue
        dt_ref = time.time() - t0
        pnames = [search._points_to_mp_names[a, b] for a, b in zip(path[:-1], path[1:])]
        hps = [np.asarray(h, dtype=np.float64) for h in search._obstacles_hp]
        ga = sc.goal_area
        box = (ga.xy1[0], ga.xy1[1], ga.xy2[0], ga.xy2[1])
        orc = PO.PlannerOracle(sc.start, sc.goal_point, box, sc.allowed_goal_theta_difference, hps, mine, car.circle_centers, car.radius)
        c2, p2, t2 = orc.run(max_expansions=10 ** 7)
        assert c2 == cost and p2 == path and np.array_equal(t2, traj), label
        assert [PO.MP_NAMES[k] for k in orc.prim_sequence(p2)] == pnames and orc.n_expanded == len(search.debug_data), label
        i = n
        out[f"r{i}_hp"] = np.concatenate(hps, axis=0) if hps else np.zeros((0, 3))
        out[f"r{i}_hp_off"] = np.concatenate([[0], np.cumsum([len(h) for h in hps])]).astype(np.int64)
        out[f"r{i}_start"] = np.array(sc.start, dtype=np.float64)
        out[f"r{i}_goal"] = np.array(sc.goal_point, dtype=np.float64)
        out[f"r{i}_goal_box"] = np.array(box, dtype=np.float64)
        out[f"r{i}_tol"] = np.float64(sc.allowed_goal_theta_difference)
        out[f"r{i}_cost"] = np.float64(cost)
        out[f"r{i}_path"] = np.array(path, dtype=np.float64)
        out[f"r{i}_prims"] = np.array([PO.MP_NAMES.index(x) for x in pnames], dtype=np.int32)
        out[f"r{i}_traj"] = np.asarray(traj, dtype=np.float64)
        out[f"r{i}_n_expanded"] = np.int64(orc.n_expanded)
        out[f"r{i}_max_open"] = np.int64(orc.max_open)
        names.append(label)
        n += 1
        print(f"route {i:2d} {label:34s}: cost {cost:8.3f}, {len(path) - 1:2d} primitives, {orc.n_expanded:6d} expansions, open list up to "
              f"{orc.max_open:6d}, {len(hps):2d} obstacles, reference {dt_ref:6.1f} s", flush=True)
    out["n_routes"] = np.int64(n)
    out["labels"] = np.array(names)          # fixed-width unicode array: plain data, no pickling
    np.savez_compressed(os.path.join(HERE, "planner_envs.npz"), **out)


if __name__ == "__main__":
    main()
