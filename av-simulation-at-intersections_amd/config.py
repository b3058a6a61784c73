"""MPC parameters: the reference's main/config/mpc_config.json keys (loaded at import by
main/lib/mpc.py:15-39) plus the Simulation class constants (main/lib/simulation.py:23-25)."""
from __future__ import annotations

import json
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_CONFIG_PATH = os.path.join(_HERE, "mpc_config.json")
# Horizons with a register-resident kernel in libjsim_mpc.so (csrc/jsim_mpc.hip: JSIM_ONE_WAVE_HORIZONS / JSIM_FOUR_WAVE_HORIZONS; a
# CPU test compares the lists).  Any other T <= 48 runs on the LDS kernel: same results, 4-6 x slower.
ONE_WAVE_HORIZONS = (13, 15, 16, 20, 25, 30)
FOUR_WAVE_HORIZONS = (32, 40)
HELP_HORIZONS = (13, 15, 16, 20, 25)     # one-wave horizons with the three-helper-wavefronts form, taken for batches of up to 256 egos
HELP_PRE_HORIZONS = (13, 20)             # ... those whose scenario-loop form (the glue inside the launch) has helpers, too


def deg2rad(x: float) -> float:
    """Bit-identical to np.deg2rad for float64: x * (pi / 180)."""
    return x * (math.pi / 180.0)


@dataclass
class MPCConfig:
    NX: int = 4
    NU: int = 2
    T: int = 13
    w_perp: float = 20.0
    w_para: float = 1.0
    R: List[float] = field(default_factory=lambda: [0.01, 0.01])
    Rd: List[float] = field(default_factory=lambda: [0.01, 1.0])
    Q_v_yaw: List[float] = field(default_factory=lambda: [0.0, 0.5])
    Qf: List[float] = field(default_factory=lambda: [1.0, 1.0, 0.0, 0.5])  # scaled by T inside, mpc.py:28
    GOAL_DIS: float = 1.5
    STOP_SPEED: float = 0.1389
    MAX_TIME: float = 13.0
    MAX_ITER: int = 1
    DU_TH: float = 0.1
    MAX_DSTEER: float = 30.0   # deg/s in the JSON
    MAX_ACCEL: float = 2.0
    MAX_DECEL: float = -10.0
    # Simulation class constants (main/lib/simulation.py:23-25)
    MAX_STEER_RAD: float = deg2rad(45.0)
    MAX_SPEED: float = 30.0 / 3.6
    MIN_SPEED: float = -5.0
    # literals in main/lib/mpc.py
    R_END: List[float] = field(default_factory=lambda: [10.0, 10.0])  # :181
    MIN_REF_SPEED: float = 10 / 3.6                                    # :99
    # NX = 5 only (main/lib/mpc_jerk.py:31,190): weight of (x[4,t+1] - x[4,t])^2
    JERK_WEIGHT: float = 1.0

    @classmethod
    def from_json(cls, path: Optional[str] = None) -> "MPCConfig":
        with open(path or DEFAULT_CONFIG_PATH, "r") as f:
            raw = json.load(f)
        known = {k: raw[k] for k in raw if k in cls.__dataclass_fields__}
        cfg = cls(**known)
        if cfg.NX not in (4, 5) or cfg.NU != 2:
            raise ValueError("NX must be 4 (main/lib/mpc.py) or 5 (the acceleration state of main/lib/mpc_jerk.py), NU = 2")
        return cfg

    @property
    def max_dsteer_rad(self) -> float:
        return deg2rad(float(self.MAX_DSTEER))
