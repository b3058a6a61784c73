"""Drop-in for the reference's `lib.mpc_with_speed` (main/lib/mpc_with_speed.py): the same controller as lib.mpc with a
per-point speed reference `cv` (xref[2] = cv[idx], :103-104), hard-coded weights (w_perp = 10, Q_v_yaw = diag(20, .5), :23,161),
MAX_DECEL = -5 (:35), speed limit Simulation.MAX_SPEED (:187) and `set_trajectory_fromarray(trajectory, cutoff_idx)` that
rebuilds cv = MAX_SPEED and zeroes it from cutoff_idx on (:276-282).  Same kernels, different data."""
from __future__ import annotations

import sys
from dataclasses import replace
from typing import Tuple

import numpy as np
import torch

from .batched import BatchedMPC
from .config import MPCConfig
from .mpc import MPC as _BaseMPC, smooth_yaw  # noqa: F401

# module constants with the reference's names and values (main/lib/mpc_with_speed.py:16-36)
NX, NU, T = 4, 2, 13
R = np.diag([0.01, 0.01])
Rd = np.diag([0.01, 1.0])
Q_v_yaw = np.diag([20, 0.5])
Qf = np.diag([1.0, 1.0, 0., 0.5]) * T
GOAL_DIS = 1.5
STOP_SPEED = 0.5 / 3.6
MAX_TIME = 13.0
MAX_ITER = 1
DU_TH = 0.1
MAX_DSTEER = np.deg2rad(30.0)
MAX_ACCEL = 2.0
MAX_DECEL = -5
MAX_SPEED = 25 / 3.6

config = replace(MPCConfig(), T=T, w_perp=10.0, w_para=1.0, R=[0.01, 0.01], Rd=[0.01, 1.0], Q_v_yaw=[20.0, 0.5],
                 Qf=[1.0, 1.0, 0.0, 0.5], GOAL_DIS=GOAL_DIS, STOP_SPEED=STOP_SPEED, MAX_DSTEER=30.0, MAX_ACCEL=MAX_ACCEL,
                 MAX_DECEL=float(MAX_DECEL))


class MPC(_BaseMPC):
    def __init__(self, cx: np.ndarray, cy: np.ndarray, cv: np.ndarray, cyaw: np.ndarray, dl: float, car_dimensions,
                 dt: float = 0.2, device: str = "cuda:0"):
        self.cv = cv
        super().__init__(cx, cy, cyaw, dl, car_dimensions, speed=30.0 / 3.6, dt=dt, device=device)  # limit = Simulation.MAX_SPEED

    def _make_engine(self, full: np.ndarray) -> BatchedMPC:
        cv = np.ascontiguousarray(self.cv, dtype=np.float64)
        if len(cv) != len(full):
            raise ValueError("cv must have one entry per path point")
        self._cv_dev = cv.copy()                                    # what the device's speed reference holds (before the cut-off)
        return BatchedMPC([full], [0], dl=self.dl, L=self.car_dimensions.distance_back_to_front_wheel,
                          speed=self.speed, dt=self.dt, T=T, config=config, device=self._device, smooth=False, cv=[cv])

    def set_trajectory_fromarray(self, trajectory: np.ndarray, cutoff_idx: int = 999):
        m = trajectory.shape[0]
        self.cx, self.cy, self.cyaw = trajectory[:, 0], trajectory[:, 1], trajectory[:, 2]
        self.cv = np.full(m, MAX_SPEED)                             # :280 -- EVERY call resets the reference to MAX_SPEED
        cut = -1
        if cutoff_idx != 999:
            self.cv[cutoff_idx:] = 0                                # :281-282 (a negative index counts from the end, as in Python)
            cut = min(cutoff_idx, m) if cutoff_idx >= 0 else max(m + cutoff_idx, 0)
        prefix = (m <= self._full.shape[0] and np.array_equal(trajectory[:, :3], self._full[:m])
                  and np.all(self._cv_dev[:m] == MAX_SPEED))
        if prefix:   # same points, and the device's reference is already MAX_SPEED there: only the visible length changes
            self._engine.set_path_len(np.array([m], dtype=np.int32))
        else:        # new points, or the constructor's cv is still on the device: upload path + a MAX_SPEED reference
            keep, self.cv = self.cv, np.full(m, MAX_SPEED)
            self._bind(np.asarray(trajectory[:, :3], dtype=np.float64))
            self.cv = keep
        self._engine.set_speed_cutoff(np.array([cut], dtype=np.int32))

    def _failure_decel(self) -> float:
        return MAX_DECEL      # this module's MAX_DECEL (-5), main/lib/mpc_with_speed.py:300

    def is_goal(self, state) -> bool:
        # main/lib/mpc_with_speed.py:313-330 with this module's GOAL_DIS / STOP_SPEED (0.5 / 3.6)
        import math
        d = math.hypot(state.x - self.goal[0], state.y - self.goal[1])
        isgoal = d <= GOAL_DIS
        if abs(self.target_ind - len(self.cx)) >= 5:
            isgoal = False
        return bool(isgoal and abs(state.v) <= STOP_SPEED)
