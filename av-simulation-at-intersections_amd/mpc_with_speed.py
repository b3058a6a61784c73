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
        self._cut = 999
        super().__init__(cx, cy, cyaw, dl, car_dimensions, speed=30.0 / 3.6, dt=dt, device=device)  # limit = Simulation.MAX_SPEED

    def _bind(self, full: np.ndarray):
        if self._engine is not None:
            self._engine.close()
        self._full = np.ascontiguousarray(full, dtype=np.float64).copy()
        cv = np.ascontiguousarray(self.cv, dtype=np.float64)
        if len(cv) != len(self._full):
            cv = np.full(len(self._full), MAX_SPEED)
        self._engine = BatchedMPC([self._full], [0], dl=self.dl, L=self.car_dimensions.distance_back_to_front_wheel,
                                  speed=self.speed, dt=self.dt, T=T, config=config, device=self._device, smooth=False,
                                  cv=[cv])

    def set_trajectory_fromarray(self, trajectory: np.ndarray, cutoff_idx: int = 999):
        self.cv = np.full(trajectory.shape[0], MAX_SPEED)          # :280
        if cutoff_idx != 999:
            self.cv[cutoff_idx:] = 0                               # :281-282
        same = (self._full is not None and trajectory.shape[0] <= self._full.shape[0]
                and np.array_equal(trajectory[:, :3], self._full[:trajectory.shape[0]])
                and self._engine.cv is not None and np.all(self._engine.cv[0] == MAX_SPEED))
        if same:
            self.cx, self.cy, self.cyaw = trajectory[:, 0], trajectory[:, 1], trajectory[:, 2]
            self._engine.set_path_len(np.array([trajectory.shape[0]], dtype=np.int32))
        else:
            super().set_trajectory_fromarray(trajectory)           # re-uploads path + a MAX_SPEED reference
        self._engine.set_speed_cutoff(np.array([cutoff_idx if cutoff_idx != 999 else -1], dtype=np.int32))

    def step(self, state) -> Tuple[float, float]:
        di, ai = super().step(state)
        if self.status == 1:
            self.ai = MAX_DECEL      # this module's MAX_DECEL (-5), main/lib/mpc_with_speed.py:300
        return self.di, self.ai
