"""The named workloads of BASELINE.json `configs`: route tables, ego batches and loop objects.

ONE definition shared by bench.py (what is timed) and tests/test_gpu_bench_workloads.py (what is checked against the
oracle), so that parity is shown on the very egos, routes and loop settings the benchmark line is quoted on.

  config 2  256 egos per GPU, horizon 20, the 12 routes of the reference's intersection()              [configs[1], headline]
  config 3  4096 egos, horizon 30, the scenario loop with four scripted obstacle vehicles: prediction -> collision check ->
            path cut-off inside every tick (the reference's dynamic-obstacle mechanism)           [configs[2]]
  config 4  4096 egos per GPU (32768 / 8), horizon 20                                                        [configs[3]]
  config 5  1024 egos per GPU (8192 / 8), horizon 40, the 48 routes of the two-lane scenario                 [configs[4]]
Routes come from the GPU planner (row f4, `planner.plan_routes`: A* over motion primitives on the reference's scenario
geometry) unless `source="synthetic"` (round 1's idealised arcs).  Egos: `synth.make_ego_batch(..., seed=1 + rank)`.
"""
from __future__ import annotations

import time
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import planner as PL
from . import synth as S
from .batched import BatchedMPC
from .closed_loop import ClosedLoop

# config -> (egos per GPU, horizon, route geometry, loop)
CONFIGS: Dict[int, dict] = {
    2: dict(batch=256, horizon=20, multi_lane=False, scenario=False,
            name="BASELINE.json configs[1]: 256-ego batch, kinematic bicycle, horizon N=20, fp64"),
    3: dict(batch=4096, horizon=30, multi_lane=False, scenario=True,
            name="BASELINE.json configs[2]: 4096-ego batch, N=30, dynamic obstacles (four scripted vehicles: prediction, "
                 "collision check and path cut-off inside every tick)"),
    4: dict(batch=4096, horizon=20, multi_lane=False, scenario=False,
            name="BASELINE.json configs[3]: 32768-ego batch over 8 GPUs = 4096 egos per GPU, N=20, fp64"),
    5: dict(batch=1024, horizon=40, multi_lane=True, scenario=False,
            name="BASELINE.json configs[4]: multi-lane geometry, 8192 egos over 8 GPUs = 1024 egos per GPU, N=40"),
}
# the scripted vehicles of config 3 (main/lib/moving_obstacles.py MovingObstacleTIntersection-style entries)
OBSTACLE_SPECS: List[dict] = [dict(direction=1, turning=False, speed=25 / 3.6, offset=None),
                              dict(direction=-1, turning=True, speed=20 / 3.6, offset=6.0),
                              dict(direction=1, turning=True, speed=15 / 3.6, offset=12.0),
                              dict(direction=-1, turning=False, speed=25 / 3.6, offset=3.0)]
MAX_AGE = 400          # ticks after which an ego that has not met MPC.is_goal re-enters (safety respawn)


def route_queries(multi_lane: bool):
    """intersection(start_pos 1..4, turn_indicator 1..3) -- 12 routes; two-lane scenario: x start lane x goal lane -- 48."""
    rad, _ = PL.car_circles()
    if multi_lane:
        return [PL.intersection_query(sp, tn, rad, sl, gl, number_of_lanes=2) for sp in (1, 2, 3, 4) for tn in (1, 2, 3)
                for sl in (1, 2) for gl in (1, 2)]
    return [PL.intersection_query(sp, tn, rad) for sp in (1, 2, 3, 4) for tn in (1, 2, 3)]


def route_table(multi_lane: bool, source: str = "planner", device_index: int = 0) -> Tuple[List[np.ndarray], Optional[dict]]:
    """(routes, planner info or None).  Routes are (M, 3) [x, y, yaw] float64, yaw unwrapped like MPC.__init__ does
    (main/lib/mpc.py:260).  The planner info carries row f4's own measurement (one timed call, untimed first use before it)."""
    info = None
    if source == "synthetic":
        rs = S.make_route_table(multi_lane=multi_lane)
    elif source == "planner":
        qs = route_queries(multi_lane)
        PL.plan_routes(qs[:1], device=device_index)      # untimed: module load, first use of the entry point
        t0 = time.perf_counter()
        res = PL.plan_routes(qs, device=device_index)
        t1 = time.perf_counter()
        if any(r.status != 0 for r in res):
            raise RuntimeError(f"route planner: status {[r.status for r in res]}")
        rs = [r.trajectory for r in res]
        info = {"routes": len(qs), "wall_ms": (t1 - t0) * 1e3, "queries": qs, "expanded": [int(r.n_expanded) for r in res],
                "points": int(sum(len(t) for t in rs))}
    else:
        raise ValueError(f"unknown route source {source!r}")
    for r in rs:
        S.smooth_yaw_inplace(r[:, 2])
    return rs, info


def ego_batch(routes, B: int, T: int, rank: int = 0) -> S.EgoBatch:
    """The egos rank `rank` owns: SURVEY 8d's distribution, whole routes (`path_len` = route length; config 3's cut-offs
    come from its obstacle vehicles, tick by tick)."""
    return S.make_ego_batch(routes, B, T, seed=1 + rank, truncate=False)


def make_engine(routes, batch: S.EgoBatch, T: int, device) -> Tuple[BatchedMPC, torch.Tensor]:
    eng = BatchedMPC(routes, batch.path_id, dl=S.DL, T=T, speed=batch.speed, device=device, smooth=False)
    eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    return eng, torch.from_numpy(batch.x0).to(eng.device)


def make_loop(cfg_id: int, eng: BatchedMPC, x0: torch.Tensor, hist_cap: int, routes=None, batch=None, respawn: str = "initial"):
    """(scenario loop or None, closed loop).  respawn="start": a finished ego re-enters at the first point of its route at
    standstill (State(x, y, yaw of the first point, v = 0): main/scenarios/mpc_intersection.py:78-79) instead of its own
    initial state."""
    from .closed_loop import ScenarioLoop
    sc = None
    if CONFIGS[cfg_id]["scenario"]:
        sc = ScenarioLoop(eng, x0, OBSTACLE_SPECS, hist_cap=hist_cap, max_age=MAX_AGE)
        loop = sc.loop
    else:
        loop = ClosedLoop(eng, x0, hist_cap=hist_cap, max_age=MAX_AGE)
    if respawn == "start":
        first = np.array([[routes[p][0, 0], routes[p][0, 1], 0.0, routes[p][0, 2]] for p in batch.path_id])
        loop.x0_spawn.copy_(torch.from_numpy(first).to(eng.device))
        loop.target_spawn.zero_()
    elif respawn != "initial":
        raise ValueError(f"unknown respawn rule {respawn!r}")
    return sc, loop
