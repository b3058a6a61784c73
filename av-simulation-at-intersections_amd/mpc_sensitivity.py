"""Drop-in for the reference's `lib.mpc_sensitivity` (main/lib/mpc_sensitivity.py), the controller behind
main/scenarios/mpc_sensitivity_analysis*.py: lib.mpc with two differences --
  * the weights (w_perp, w_para, R, Rd, Q_v_yaw, Qf) and the limits (MAX_DSTEER, MAX_ACCEL, MAX_DECEL) are re-read from
    config/mpc_config_sensitivity.json inside EVERY solve (:153-166), which is how the analysis scripts sweep them: they
    rewrite the file between runs;
  * no `speed` argument: the speed rows use Simulation.MAX_SPEED (:207).
Here the file is re-read before every step and, when it changed, pushed to the device with jsim_mpc_update_cfg; the
horizon T, GOAL_DIS and STOP_SPEED stay module constants read at import, as in the reference (:24-41).
`CONFIG_PATH` may be pointed at the reference tree's own file by the integrator."""
from __future__ import annotations

import json
import os
import sys
from dataclasses import replace
from typing import Tuple

import numpy as np

from .batched import BatchedMPC
from .config import MPCConfig
from .mpc import MPC as _BaseMPC, MPCSolutionNotFoundException, smooth_yaw  # noqa: F401

CONFIG_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mpc_config_sensitivity.json")
with open(CONFIG_PATH, "r") as _f:
    config = json.load(_f)

# module constants with the reference's names (main/lib/mpc_sensitivity.py:24-41)
NX = config["NX"]
NU = config["NU"]
T = config["T"]
GOAL_DIS = config["GOAL_DIS"]
STOP_SPEED = config["STOP_SPEED"]
MAX_TIME = config["MAX_TIME"]
MAX_ITER = config["MAX_ITER"]
DU_TH = config["DU_TH"]
MAX_DSTEER = np.deg2rad(config["MAX_DSTEER"])
MAX_ACCEL = config["MAX_ACCEL"]
MAX_DECEL = config["MAX_DECEL"]
MAX_SPEED = 30.0 / 3.6   # Simulation.MAX_SPEED, main/lib/simulation.py:24


def _load(path: str) -> MPCConfig:
    with open(path, "r") as f:
        raw = json.load(f)
    base = MPCConfig()
    known = {k: raw[k] for k in ("w_perp", "w_para", "R", "Rd", "Q_v_yaw", "Qf", "MAX_DSTEER", "MAX_ACCEL", "MAX_DECEL") if k in raw}
    known = {k: ([float(x) for x in v] if isinstance(v, list) else float(v)) for k, v in known.items()}
    return replace(base, T=T, GOAL_DIS=GOAL_DIS, STOP_SPEED=STOP_SPEED, **known)


class MPC(_BaseMPC):
    def __init__(self, cx: np.ndarray, cy: np.ndarray, cyaw: np.ndarray, dl: float, car_dimensions, dt: float = 0.2,
                 device: str = "cuda:0"):
        self._cfg_now = _load(CONFIG_PATH)
        super().__init__(cx, cy, cyaw, dl, car_dimensions, speed=MAX_SPEED, dt=dt, device=device)

    def _make_engine(self, full: np.ndarray) -> BatchedMPC:
        return BatchedMPC([full], [0], dl=self.dl, L=self.car_dimensions.distance_back_to_front_wheel,
                          speed=self.speed, dt=self.dt, T=T, config=self._cfg_now, device=self._device, smooth=False)

    def _failure_decel(self) -> float:
        return MAX_DECEL                              # the module constant read at import (:41), not the file's current value

    def step(self, state) -> Tuple[float, float]:
        cfg = _load(CONFIG_PATH)                      # the reference opens the file in every solve (:153-154)
        if cfg != self._cfg_now:
            self._engine.update_config(cfg)
            self._cfg_now = cfg
        return super().step(state)
