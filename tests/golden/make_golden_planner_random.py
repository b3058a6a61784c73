#!/usr/bin/env python3
"""Writes tests/golden/planner_random.npz: route queries on RANDOM obstacle fields (boxes and octagons of random size, a fence
around the world) solved by oracle/planner_oracle.py -- the numpy restatement that reproduces the reference's own planner bit for
bit on its 18 + 36 stored routes (tests/test_planner.py).  These are the ORACLE's outputs, not the reference's: they widen what
the HIP planner is compared on (dead ends, stale pops, open lists of thousands of entries, exhausted searches, obstacle sets
of 7-17 pieces that the intersection scenarios do not have), they do not pin the oracle.

usage: make_golden_planner_random.py [first_seed last_seed max_expansions]   (defaults 0 96 6000; ~10 min on 4 cores)"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import planner_oracle as PO     # noqa: E402

L, WIDTH, EXTRA = 2.86, 2.0, 0.64
RADIUS = WIDTH / (2 ** .5)
_len = L + EXTRA
CENTRES = np.array([[L / 2 + (_len / 2 - WIDTH / 2), 0.0], [L / 2 - (_len / 2 - WIDTH / 2), 0.0]])


def field(seed):
    rng = np.random.default_rng(1000 + seed)
    obs = []
    for _ in range(int(rng.integers(3, 14))):
        c = rng.uniform(-28, 28, 2)
        if np.hypot(*c) < 6:                      # keep the start clear
            continue
        if rng.random() < 0.5:
            obs.append(PO.box_halfplanes(tuple(rng.uniform(1.5, 9.0, 2)), tuple(c), RADIUS))
        else:
            obs.append(PO.circle_halfplanes(float(rng.uniform(0.8, 4.0)), tuple(c), RADIUS))
    for cx, cy, w, h in ((0, 36, 80, 4), (0, -36, 80, 4), (36, 0, 4, 80), (-36, 0, 4, 80)):   # a fence: a hopeless search runs empty
        obs.append(PO.box_halfplanes((w, h), (cx, cy), RADIUS))
    th0 = float(rng.uniform(-np.pi, np.pi))
    ang, dist = float(rng.uniform(-np.pi, np.pi)), float(rng.uniform(10, 24))
    gx, gy = dist * np.cos(ang), dist * np.sin(ang)
    gth = float(rng.uniform(-np.pi, np.pi))
    half = float(rng.uniform(2.0, 4.0))
    tol = float(rng.choice([np.pi / 4, np.pi / 6, np.pi / 8]))
    return dict(start=(0.0, 0.0, th0), goal=(gx, gy, gth), goal_box=(gx - half, gy - half, gx + half, gy + half), tol=tol, obstacles=obs)


def solve(args):
    seed, budget = args
    q = field(seed)
    orc = PO.PlannerOracle(q["start"], q["goal"], q["goal_box"], q["tol"], q["obstacles"], PO.make_motion_primitives(), CENTRES, RADIUS)
    t0 = time.perf_counter()
    try:
        cost, path, traj = orc.run(max_expansions=budget)
        out = dict(status=0, cost=cost, path=np.array(path), traj=traj, prims=np.array(orc.prim_sequence(path), dtype=np.int32))
    except RuntimeError:
        return seed, None, time.perf_counter() - t0
    except Exception as e:
        assert str(e) == "No solution found."
        out = dict(status=1, cost=np.nan, path=np.zeros((0, 3)), traj=np.zeros((0, 3)), prims=np.zeros(0, dtype=np.int32))
    out.update(n_expanded=orc.n_expanded, max_open=orc.max_open, seed=seed, start=np.array(q["start"]), goal=np.array(q["goal"]),
               goal_box=np.array(q["goal_box"]), tol=q["tol"], hp=np.concatenate(q["obstacles"], axis=0),
               hp_off=np.cumsum([0] + [len(o) for o in q["obstacles"]]).astype(np.int32))
    return seed, out, time.perf_counter() - t0


if __name__ == "__main__":
    first, last, budget = (int(v) for v in (sys.argv[1:4] + ["0", "96", "6000"][len(sys.argv) - 1:]))
    arrays, n = {}, 0
    with ProcessPoolExecutor(max_workers=4) as ex:
        for seed, out, dt in ex.map(solve, [(s, budget) for s in range(first, last)]):
            print(seed, "budget exhausted" if out is None else (out["status"], out["n_expanded"], out["max_open"]), f"{dt:.1f}s", flush=True)
            if out is None or (out["status"] == 1 and out["n_expanded"] < 5):
                continue                          # over budget, or walled in at the start: not kept
            for k, v in out.items():
                arrays[f"r{n}_{k}"] = v
            n += 1
    arrays["n_routes"] = n
    arrays["radius"] = RADIUS
    arrays["circle_centers"] = CENTRES
    np.savez_compressed(os.path.join(HERE, "planner_random.npz"), **arrays)
    print(n, "routes kept")
