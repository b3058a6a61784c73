"""Drop-in for the reference's `lib.mpc` module (main/lib/mpc.py): same names, same constructor /
`set_trajectory_fromarray` / `step` / `is_goal` / `get_current_xref_deviation` surface and the same
error behaviour, with the per-timestep solve running on the MI355X through libjsim_mpc.so.

    from lib.mpc import MPC, MAX_ACCEL            # what the scenario scripts import
    mpc = MPC(cx, cy, cyaw, dl, car_dimensions, speed=30/3.6, dt=0.2)
    mpc.set_trajectory_fromarray(trajectory_full[:cutoff_idx])
    delta, acceleration = mpc.step(state)          # steer first, accel second (mpc.py:303)

A single ego is a batch of one (one wavefront); there is no CPU fallback -- constructing an MPC without
a HIP device or without the built library raises.
"""
from __future__ import annotations

import math
import sys
from typing import List, Optional, Tuple

import numpy as np
import torch

from .batched import BatchedMPC
from .config import MPCConfig

config = MPCConfig.from_json()

# module constants with the reference's names (main/lib/mpc.py:20-39)
NX = config.NX
NU = config.NU
T = config.T
w_perp = config.w_perp
w_para = config.w_para
R = np.diag(config.R)
Rd = np.diag(config.Rd)
Q_v_yaw = np.diag(config.Q_v_yaw)
Qf = np.diag(config.Qf) * T
GOAL_DIS = config.GOAL_DIS
STOP_SPEED = config.STOP_SPEED
MAX_TIME = config.MAX_TIME
MAX_ITER = config.MAX_ITER
DU_TH = config.DU_TH
MAX_DSTEER = config.max_dsteer_rad
MAX_ACCEL = config.MAX_ACCEL
MAX_DECEL = config.MAX_DECEL


class MPCSolutionNotFoundException(Exception):
    pass


def smooth_yaw(yaw):
    """main/lib/mpc.py:46-58 (in place)."""
    from .synth import smooth_yaw_inplace
    return smooth_yaw_inplace(yaw)


class MPC:
    def __init__(self, cx: np.ndarray, cy: np.ndarray, cyaw: np.ndarray, dl: float, car_dimensions,
                 speed: float = 30 / 3.6, dt: float = 0.2, device: str = "cuda:0"):
        self.cx = cx
        self.cy = cy
        cyaw = smooth_yaw(cyaw)  # mutates the caller's array, like the reference (mpc.py:260)
        self.cyaw = cyaw
        self.dl = dl
        self.dt = dt
        self.car_dimensions = car_dimensions
        self.speed = speed
        self.goal: Tuple[float, float] = cx[-1], cy[-1]
        self.target_ind: int = 0
        self.odelta: Optional[np.ndarray] = None
        self.oa: Optional[np.ndarray] = None
        self.di: float = 0.0
        self.ai: float = 0.0
        self.ox = self.oy = self.oyaw = self.ov = self.xref = None
        self.status = 0
        self.n_iter = 0
        self.active_constraints: List[int] = []
        self._device = device
        self._x0 = torch.zeros(1, 4, dtype=torch.float64, device=device)
        self._engine: Optional[BatchedMPC] = None
        self._full: Optional[np.ndarray] = None
        self._rebound = False
        self._oa_dev = self._od_dev = None   # the host arrays whose values the device's warm start currently holds
        self._bind(np.stack([np.asarray(cx, dtype=np.float64), np.asarray(cy, dtype=np.float64),
                             np.asarray(cyaw, dtype=np.float64)], axis=1))

    # a (re)upload of the path table: the controller state lives in this object (target_ind, oa, odelta, di -- kept across
    # set_trajectory_fromarray exactly like the reference, main/lib/mpc.py:279-282) and is pushed to the new engine by step()
    def _make_engine(self, full: np.ndarray) -> BatchedMPC:
        return BatchedMPC([full], [0], dl=self.dl, L=self.car_dimensions.distance_back_to_front_wheel,
                          speed=self.speed, dt=self.dt, T=T, config=config, device=self._device, smooth=False)

    def _bind(self, full: np.ndarray):
        if self._engine is not None:
            self._engine.close()
        self._full = np.ascontiguousarray(full, dtype=np.float64).copy()
        self._engine = self._make_engine(self._full)
        self._rebound = True

    def set_trajectory_fromarray(self, trajectory: np.ndarray):
        self.cx = trajectory[:, 0]
        self.cy = trajectory[:, 1]
        self.cyaw = trajectory[:, 2]
        m = trajectory.shape[0]
        if m <= self._full.shape[0] and np.array_equal(trajectory[:, :3], self._full[:m]):
            self._engine.set_path_len(np.array([m], dtype=np.int32))  # the loop's trajectory_full[:cutoff]
        else:
            self._bind(np.asarray(trajectory[:, :3], dtype=np.float64))

    def _failure_decel(self) -> float:
        return MAX_DECEL  # mpc.py:301 (the variants apply their own module constant)

    def step(self, state) -> Tuple[float, float]:
        eng = self._engine
        # the controller state is resident on the device between ticks; what the caller (or a re-bound path) may have
        # changed on the host side is mirrored first: the remembered index, the warm start (None -> zeros, mpc.py:225-227)
        eng.target_ind.fill_(int(self.target_ind))
        if self.oa is None or self.odelta is None:
            eng.oa.zero_(); eng.od.zero_()
        elif self._rebound or self.oa is not self._oa_dev or self.odelta is not self._od_dev:
            eng.load_state(oa=np.asarray(self.oa, dtype=np.float64)[None], od=np.asarray(self.odelta, dtype=np.float64)[None])
        self._rebound = False
        self._x0.copy_(torch.tensor([[state.x, state.y, state.v, state.yaw]], dtype=torch.float64))  # mpc.py:291
        eng.solve(self._x0)
        out = eng.read_back()           # one synchronising device->host copy
        status = int(out["status"][0])
        self.status = status
        if status == 2:
            raise Exception("something wrong")  # main/lib/trajectories.py:120
        self.n_iter = int(out["n_iter"][0])
        self.target_ind = int(out["target_ind"][0])
        self.xref = out["xref"][0]
        if status == 0:
            self.oa, self.odelta = out["oa"][0], out["od"][0]
            self.ox, self.oy, self.ov, self.oyaw = out["ox"][0], out["oy"][0], out["ov"][0], out["oyaw"][0]
            words = out["active_mask"][0].view(np.uint32)
            self.active_constraints = [i for i in range(8 * eng.T) if (int(words[i >> 5]) >> (i & 31)) & 1]
            self.di, self.ai = float(self.odelta[0]), float(self.oa[0])
        else:
            print("Error: Cannot solve mpc...", file=sys.stderr)  # mpc.py:208
            self.oa = self.odelta = self.ox = self.oy = self.oyaw = self.ov = None
            self.active_constraints = []
            self.ai = self._failure_decel()
        self._oa_dev, self._od_dev = self.oa, self.odelta   # these very arrays are what the device holds
        return self.di, self.ai

    def get_current_xref_deviation(self):
        # mpc.py:305-312; after a failed solve self.ox is None and this raises TypeError, like the reference
        ref_point = np.array([self.cx[self.target_ind], self.cy[self.target_ind]])
        true_point = np.array([self.ox[0], self.oy[0]])
        ref_yaw_perp = self.cyaw[self.target_ind] + np.pi / 2
        diff_vect = ref_point - true_point
        ref_dir_normal = np.array([np.cos(ref_yaw_perp) * diff_vect[0], np.sin(ref_yaw_perp) * diff_vect[1]])
        return np.linalg.norm(ref_dir_normal)

    def is_goal(self, state) -> bool:
        # mpc.py:314-330
        d = math.hypot(state.x - self.goal[0], state.y - self.goal[1])
        isgoal = d <= GOAL_DIS
        if abs(self.target_ind - len(self.cx)) >= 5:
            isgoal = False
        isstop = abs(state.v) <= STOP_SPEED
        return bool(isgoal and isstop)
