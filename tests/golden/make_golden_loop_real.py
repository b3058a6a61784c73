#!/usr/bin/env python3
"""G4 on the REAL route: tests/golden/loop_real_T13.npz -- the `mpc_intersection` loop (config 1) as the
reference runs it, recorded tick by tick.  Build container only (needs /root/reference).

Everything on the path is the reference's own code, imported from /root/reference/main and executed unmodified:
    envs.intersection.intersection(start_pos=1, turn_indicator=1)        the scenario (mpc_intersection.py:37-41 defaults)
    lib.mp_search_ww_generic.MotionPrimitiveSearch(...).run()            the planner -> trajectory_full           (:63-64)
    lib.mpc.MPC(cx, cy, cyaw, dl, dt, car_dimensions, speed=30/3.6)      the controller, stock T = 13             (:75-76)
    lib.moving_obstacles.MovingObstacleTIntersection x 2                 the script's two vehicles                (:46-49)
    lib.trajectories / lib.moving_obstacles_prediction / lib.collision_avoidance / lib.simulation.HistorySimulation
The loop body is the script's own sequence of calls, lines :99-163 (its `main()` cannot be called: it loads pickles and
animates with matplotlib).  Two stand-ins, both stated: the motion primitives are regenerated from the reference's recipe
(main/create_motion_primitives_bicycle_model.py:12-27; the shipped pickles are never loaded), and `cvxpy` is the recording
stand-in tests/golden/cvxpy_recorder.py, whose `solve()` returns the exact optimum of the problem the reference emitted (ECOS's
stopping tolerance stays unpinned).  Unlike tests/golden/loop_closed_T13.npz (synthetic arc, the oracle in place of MPC.step)
nothing under oracle/ takes part.

Stored: the planned trajectory, and per tick the state, progress index in / out, previous and new path length, collision flag,
target_ind in / out, solver status, (delta, acceleration), xref deviation, the obstacles' get() tuples.
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "oracle"))


def main():
    import make_golden_refqp as RQ
    refmpc, rec = RQ.import_reference(13)
    from make_golden_planner import reference_primitives
    from envs.intersection import intersection
    from lib.car_dimensions import BicycleModelDimensions
    from lib.collision_avoidance import check_collision_moving_cars, get_cutoff_curve_by_position_idx
    from lib.moving_obstacles import MovingObstacleTIntersection
    from lib.moving_obstacles_prediction import MovingObstaclesPrediction
    from lib.mp_search_ww_generic import MotionPrimitiveSearch
    from lib.simulation import HistorySimulation, Simulation, State
    from lib.trajectories import calc_nearest_index_in_direction, resample_curve
    MPC, MAX_ACCEL = refmpc.MPC, refmpc.MAX_ACCEL

    DT = 0.2
    car = BicycleModelDimensions(skip_back_circle_collision_checking=False)
    mps = reference_primitives(car)
    scenario = intersection(start_pos=1, turn_indicator=1)
    moving_obstacles = [MovingObstacleTIntersection(car, direction=1, offset=2., turning=False, speed=25 / 3.6, dt=DT),
                        MovingObstacleTIntersection(car, direction=-1, offset=4., turning=True, speed=25 / 3.6, dt=DT)]
    search = MotionPrimitiveSearch(scenario, car, mps, margin=car.radius)
    _, _, trajectory_full = search.run(debug=False)
    planned = trajectory_full.copy()                       # before MPC.__init__ unwraps the yaw column in place
    dl = np.linalg.norm(trajectory_full[0, :2] - trajectory_full[1, :2])
    mpc = MPC(cx=trajectory_full[:, 0], cy=trajectory_full[:, 1], cyaw=trajectory_full[:, 2], dl=dl, dt=DT, car_dimensions=car,
              speed=30 / 3.6)
    state = State(x=trajectory_full[0, 0], y=trajectory_full[0, 1], yaw=trajectory_full[0, 2], v=0.0)
    simulation = HistorySimulation(car_dimensions=car, sample_time=DT, initial_state=state)
    TIME_HORIZON, FRAME_WINDOW = 7., 10
    EXTRA_CUTOFF_MARGIN = 4 * int(math.ceil(car.radius / dl))
    traj_agent_idx, tmp_trajectory = 0, None
    ticks = []
    for i in range(600):
        if mpc.is_goal(state):
            break
        prev_len = -1 if tmp_trajectory is None else len(tmp_trajectory)
        idx_in = traj_agent_idx
        if tmp_trajectory is None or np.any(tmp_trajectory[traj_agent_idx, :] != tmp_trajectory[-1, :]):
            traj_agent_idx = calc_nearest_index_in_direction(state, trajectory_full[:, 0], trajectory_full[:, 1],
                                                             start_index=traj_agent_idx, forward=True)
        trajectory_res = trajectory = trajectory_full[traj_agent_idx:]
        if state.v < Simulation.MAX_SPEED:
            resample_dl = np.zeros((trajectory_res.shape[0],)) + MAX_ACCEL
            resample_dl = np.cumsum(resample_dl) + state.v
            resample_dl = DT * np.minimum(resample_dl, Simulation.MAX_SPEED)
            trajectory_res = resample_curve(trajectory_res, dl=resample_dl)
        else:
            trajectory_res = resample_curve(trajectory_res, dl=DT * Simulation.MAX_SPEED)
        obst = [o.get() for o in moving_obstacles]
        trajs = [np.vstack(MovingObstaclesPrediction(*g, sample_time=DT, car_dimensions=car).state_prediction(TIME_HORIZON)).T
                 for g in obst]
        collision_xy = check_collision_moving_cars(car, trajectory_res, trajectory, trajs, frame_window=FRAME_WINDOW)
        if collision_xy is not None:
            cutoff_idx = get_cutoff_curve_by_position_idx(trajectory_full, collision_xy[0], collision_xy[1]) - EXTRA_CUTOFF_MARGIN
            cutoff_idx = max(traj_agent_idx + 1, cutoff_idx)
            tmp_trajectory = trajectory_full[:cutoff_idx]
        else:
            tmp_trajectory = trajectory_full
        mpc.set_trajectory_fromarray(tmp_trajectory)
        tind_in = mpc.target_ind
        del rec.RECORDS[:]
        delta, acceleration = mpc.step(state)
        status = 0 if mpc.odelta is not None else 1
        dev = mpc.get_current_xref_deviation() if status == 0 else np.nan
        ticks.append((state.x, state.y, state.yaw, state.v, idx_in, prev_len, traj_agent_idx, len(tmp_trajectory),
                      0.0 if collision_xy is None else 1.0, tind_in, mpc.target_ind, status, delta, acceleration, dev)
                     + tuple(np.array(obst, dtype=float).reshape(-1)))
        for o in moving_obstacles:
            o.step()
        state = simulation.step(a=acceleration, delta=delta, xref_deviation=dev if status == 0 else None)
    ticks = np.array(ticks, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "loop_real_T13.npz"), ticks=ticks, planned=planned, trajectory_smoothed=trajectory_full,
                        dl=np.float64(dl), margin=np.int64(EXTRA_CUTOFF_MARGIN), reached_goal=np.array(i < 599),
                        final=np.array([state.x, state.y, state.yaw, state.v]),
                        obstacle_specs=np.array([[1, 2., 0, 25 / 3.6], [-1, 4., 1, 25 / 3.6]]))   # direction, offset, turning, speed
    print(f"real-route closed loop: route of {len(planned)} points (dl = {dl:.6f}), {len(ticks)} ticks, goal reached: {i < 599}, "
          f"ticks with a cut-off: {int(ticks[:, 8].sum())}, failed solves: {int(ticks[:, 11].sum())}, final v {state.v:.4f}")


if __name__ == "__main__":
    main()
