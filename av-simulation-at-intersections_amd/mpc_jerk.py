"""Drop-in for the reference's `lib.mpc_jerk` (main/lib/mpc_jerk.py): the controller of lib.mpc with a fifth state, the
acceleration -- A[4][4] = 1, A[2][4] = dt, B[4][0] = dt (:67-78) -- whose initial value is a free decision variable
(`x[:4, 0] == x0`, :193), the extra cost (x[4,t+1] - x[4,t])^2 for t < T-1 (:190), hard-coded weights (w_perp = 10, :167;
Rd = diag(.3, 1), :23), MAX_DECEL = -5 (:39), the speed limit Simulation.MAX_SPEED (:194) and no `speed` argument (:250).
`oa` is u[0, :], the input of the acceleration state, and `ai = oa[0]` is what the caller applies (:309).

The reference itself never imports this module (`scenarios/mpc_intersection.py:21` and
`scenarios/mpc_sensitivity_analysis.py:20` hold it in commented-out imports).  On the device it is the LDS-resident
kernel with 2T + 1 decision variables (csrc/jsim_mpc.hip, JERK)."""
from __future__ import annotations

import sys
from dataclasses import replace
from typing import Tuple

import numpy as np

from .batched import BatchedMPC
from .config import MPCConfig
from .mpc import MPC as _BaseMPC, MPCSolutionNotFoundException, smooth_yaw  # noqa: F401

# module constants with the reference's names and values (main/lib/mpc_jerk.py:16-39)
NX, NU, T = 5, 2, 13
R = np.diag([0.01, 0.01])
Rd = np.diag([.3, 1.0])
Q_v_yaw = np.diag([0., 0.5])
Qf = np.diag([1.0, 1.0, 0., 0.5, 0]) * T
GOAL_DIS = 1.5
STOP_SPEED = 0.5 / 3.6
MAX_TIME = 13.0
jerk_penalty_weight = 1
MAX_ITER = 1
DU_TH = 0.1
MAX_DSTEER = np.deg2rad(30.0)
MAX_ACCEL = 2.0
MAX_DECEL = -5

config = replace(MPCConfig(), NX=NX, T=T, w_perp=10.0, w_para=1.0, R=[0.01, 0.01], Rd=[0.3, 1.0], Q_v_yaw=[0.0, 0.5],
                 Qf=[1.0, 1.0, 0.0, 0.5], GOAL_DIS=GOAL_DIS, STOP_SPEED=STOP_SPEED, MAX_DSTEER=30.0, MAX_ACCEL=MAX_ACCEL,
                 MAX_DECEL=float(MAX_DECEL), JERK_WEIGHT=float(jerk_penalty_weight))


class MPC(_BaseMPC):
    def __init__(self, cx: np.ndarray, cy: np.ndarray, cyaw: np.ndarray, dl: float, car_dimensions, dt: float = 0.2,
                 device: str = "cuda:0"):
        super().__init__(cx, cy, cyaw, dl, car_dimensions, speed=config.MAX_SPEED, dt=dt, device=device)  # x[2,:] <= Simulation.MAX_SPEED (:194)

    def _make_engine(self, full: np.ndarray) -> BatchedMPC:
        return BatchedMPC([full], [0], dl=self.dl, L=self.car_dimensions.distance_back_to_front_wheel,
                          speed=self.speed, dt=self.dt, T=T, config=config, device=self._device, smooth=False)

    def _failure_decel(self) -> float:
        return MAX_DECEL     # this module's MAX_DECEL (-5), main/lib/mpc_jerk.py:311

    def is_goal(self, state) -> bool:
        import math
        d = math.hypot(state.x - self.goal[0], state.y - self.goal[1])
        isgoal = d <= GOAL_DIS
        if abs(self.target_ind - len(self.cx)) >= 5:
            isgoal = False
        return bool(isgoal and abs(state.v) <= STOP_SPEED)
