"""ctypes wrapper around oracle/libmpc_oracle.so (the CPU restatement of the reference MPC step).

TEST INFRASTRUCTURE ONLY -- see oracle/mpc_oracle.h.  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by the product package.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmpc_oracle.so")

# Stock main/config/mpc_config.json values (+ Simulation constants, lib/simulation.py:23-25).
STOCK_CONFIG = {
    "NX": 4, "NU": 2, "T": 13, "w_perp": 20.0, "w_para": 1.0, "R": [0.01, 0.01], "Rd": [0.01, 1.0],
    "Q_v_yaw": [0.0, 0.5], "Qf": [1.0, 1.0, 0.0, 0.5], "GOAL_DIS": 1.5, "STOP_SPEED": 0.1389,
    "MAX_TIME": 13.0, "MAX_ITER": 1, "DU_TH": 0.1, "MAX_DSTEER": 30.0, "MAX_ACCEL": 2.0,
    "MAX_DECEL": -10,
}


def deg2rad(x: float) -> float:
    """np.deg2rad(x) == x * (pi / 180) in float64."""
    return x * (math.pi / 180.0)


class OrcParams(C.Structure):
    _fields_ = [
        ("T", C.c_int32), ("max_iter", C.c_int32),
        ("dt", C.c_double), ("dl", C.c_double), ("L", C.c_double),
        ("w_perp", C.c_double), ("w_para", C.c_double),
        ("R", C.c_double * 2), ("Rd", C.c_double * 2), ("Q_v_yaw", C.c_double * 2),
        ("Qf", C.c_double * 4), ("R_end", C.c_double * 2),
        ("max_dsteer", C.c_double), ("max_accel", C.c_double), ("max_decel", C.c_double),
        ("max_steer", C.c_double), ("max_speed", C.c_double), ("min_speed", C.c_double),
        ("min_ref_speed", C.c_double), ("goal_dis", C.c_double), ("stop_speed", C.c_double),
        ("nx", C.c_int32), ("reserved_", C.c_int32), ("jerk_weight", C.c_double),
    ]


class OrcStepOut(C.Structure):
    _fields_ = [
        ("oa", C.c_void_p), ("od", C.c_void_p), ("ox", C.c_void_p), ("oy", C.c_void_p),
        ("ov", C.c_void_p), ("oyaw", C.c_void_p), ("xref", C.c_void_p), ("xbar", C.c_void_p),
        ("idx", C.c_void_p), ("reaches_end", C.c_void_p), ("lam", C.c_void_p),
        ("active_mask", C.c_void_p), ("H", C.c_void_p), ("g", C.c_void_p),
        ("target_ind", C.c_int64), ("n_iter", C.c_int32), ("status", C.c_int32),
    ]


def make_params(T: int = 13, dt: float = 0.2, dl: float = 0.083, L: float = 2.86,
                config: Optional[dict] = None) -> OrcParams:
    cfg = dict(STOCK_CONFIG)
    if config:
        cfg.update(config)
    p = OrcParams()
    p.T = int(T)
    p.max_iter = int(cfg["MAX_ITER"])
    p.dt, p.dl, p.L = float(dt), float(dl), float(L)
    p.w_perp, p.w_para = float(cfg["w_perp"]), float(cfg["w_para"])
    p.R[:] = [float(v) for v in cfg["R"]]
    p.Rd[:] = [float(v) for v in cfg["Rd"]]
    p.Q_v_yaw[:] = [float(v) for v in cfg["Q_v_yaw"]]
    p.Qf[:] = [float(v) for v in cfg["Qf"]]
    p.R_end[:] = [10.0, 10.0]                       # mpc.py:181
    p.max_dsteer = deg2rad(float(cfg["MAX_DSTEER"]))  # mpc.py:37
    p.max_accel = float(cfg["MAX_ACCEL"])
    p.max_decel = float(cfg["MAX_DECEL"])
    p.max_steer = deg2rad(45.0)                     # simulation.py:23
    p.max_speed = 30.0 / 3.6                        # simulation.py:24
    p.min_speed = -5.0                              # simulation.py:25
    p.min_ref_speed = 10 / 3.6                      # mpc.py:99
    p.goal_dis = float(cfg["GOAL_DIS"])
    p.stop_speed = float(cfg["STOP_SPEED"])
    p.nx = int(cfg.get("NX", 4))                    # 5: main/lib/mpc_jerk.py
    p.jerk_weight = float(cfg.get("JERK_WEIGHT", 1.0))  # mpc_jerk.py:31
    return p


def nvar(p) -> int:
    """Decision variables of the condensed QP: 2T, plus the free acc_0 of the acceleration-state variant."""
    return 2 * p.T + (1 if p.nx == 5 else 0)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(_HERE, "mpc_oracle.c")
    hdr = os.path.join(_HERE, "mpc_oracle.h")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmpc_oracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
        bp = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
        L.orc_smooth_yaw.argtypes = [dp, C.c_int64]
        L.orc_smooth_yaw.restype = None
        L.orc_nearest_index_in_direction.argtypes = [C.c_double, C.c_double, dp, dp, C.c_int64,
                                                     C.c_int64, C.c_int, C.POINTER(C.c_int64)]
        L.orc_nearest_index_in_direction.restype = C.c_int
        L.orc_calc_ref_trajectory.argtypes = [C.POINTER(OrcParams), C.c_double, C.c_double, C.c_double,
                                              dp, dp, dp, C.c_int64, C.c_int64, dp, ip, bp,
                                              C.POINTER(C.c_int64)]
        L.orc_calc_ref_trajectory.restype = C.c_int
        L.orc_plant_step.argtypes = [C.POINTER(OrcParams), dp, C.c_double, C.c_double]
        L.orc_plant_step.restype = None
        L.orc_predict_motion.argtypes = [C.POINTER(OrcParams), dp, dp, dp, dp]
        L.orc_predict_motion.restype = None
        L.orc_linear_model_matrix.argtypes = [C.c_double] * 5 + [dp, dp, dp]
        L.orc_linear_model_matrix.restype = None
        L.orc_build_qp.argtypes = [C.POINTER(OrcParams), dp, dp, dp, bp, C.c_double, dp, dp, dp, dp, bp,
                                   dp, dp]
        L.orc_build_qp.restype = C.c_int
        L.orc_solve_qp.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, bp, dp, dp, C.POINTER(C.c_int32)]
        L.orc_solve_qp.restype = C.c_int
        L.orc_mpc_step.argtypes = [C.POINTER(OrcParams)] + [C.c_double] * 4 + [dp, dp, dp, C.c_int64,
                                   C.c_int64, C.c_double, C.c_void_p, C.c_void_p,
                                   C.POINTER(OrcStepOut)]
        L.orc_mpc_step.restype = C.c_int
        L.orc_mpc_step_batch.argtypes = [C.POINTER(OrcParams), C.c_int32] + [C.c_void_p] * 19 + [C.c_int32]
        L.orc_mpc_step_batch.restype = C.c_int
        L.orc_mpc_step_batch_cv.argtypes = [C.POINTER(OrcParams), C.c_int32] + [C.c_void_p] * 21 + [C.c_int32]
        L.orc_mpc_step_batch_cv.restype = C.c_int
        L.orc_mpc_step_cv.argtypes = [C.POINTER(OrcParams)] + [C.c_double] * 4 + [dp, dp, dp, C.c_void_p, C.c_int64,
                                      C.c_int64, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.POINTER(OrcStepOut)]
        L.orc_mpc_step_cv.restype = C.c_int
        L.orc_closed_loop.argtypes = [C.POINTER(OrcParams), C.c_int32, C.c_int32] + [C.c_void_p] * 15 + [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32]
        L.orc_closed_loop.restype = C.c_int
        L.orc_xref_deviation.argtypes = [dp, dp, dp, C.c_int64, C.c_double, C.c_double]
        L.orc_xref_deviation.restype = C.c_double
        L.orc_is_goal.argtypes = [C.POINTER(OrcParams)] + [C.c_double] * 5 + [C.c_int64, C.c_int64]
        L.orc_is_goal.restype = C.c_int
        _lib = L
    return _lib


def _c(a, dtype=np.float64):
    return np.ascontiguousarray(a, dtype=dtype)


def smooth_yaw(yaw):
    y = _c(yaw).copy()
    lib().orc_smooth_yaw(y, y.shape[0])
    return y


def nearest_index_in_direction(x, y, cx, cy, start_index=0, forward=True):
    out = C.c_int64(0)
    cx, cy = _c(cx), _c(cy)
    st = lib().orc_nearest_index_in_direction(float(x), float(y), cx, cy, cx.shape[0],
                                              int(start_index), int(bool(forward)), C.byref(out))
    return int(st), int(out.value)


def calc_ref_trajectory(p, sx, sy, sv, cx, cy, cyaw, start_idx):
    T = p.T
    xref = np.zeros((4, T + 1))
    idx = np.zeros(T + 1, dtype=np.int64)
    rend = np.zeros(T + 1, dtype=np.uint8)
    tind = C.c_int64(0)
    cx, cy, cyaw = _c(cx), _c(cy), _c(cyaw)
    st = lib().orc_calc_ref_trajectory(C.byref(p), float(sx), float(sy), float(sv), cx, cy, cyaw,
                                       cx.shape[0], int(start_idx), xref, idx, rend, C.byref(tind))
    return int(st), xref, idx, rend.astype(bool), int(tind.value)


def plant_step(p, state_xyvyaw, a, delta):
    s = _c(state_xyvyaw).copy()
    lib().orc_plant_step(C.byref(p), s, float(a), float(delta))
    return s


def predict_motion(p, x0, oa, od):
    xbar = np.zeros((4, p.T + 1))
    lib().orc_predict_motion(C.byref(p), _c(x0), _c(oa), _c(od), xbar)
    return xbar


def linear_model_matrix(v, phi, delta, dt, L):
    A = np.zeros((4, 4)); B = np.zeros((4, 2)); Cv = np.zeros(4)
    lib().orc_linear_model_matrix(float(v), float(phi), float(delta), float(dt), float(L), A, B, Cv)
    return A, B, Cv


def build_qp(p, xref, xbar, x0, reaches_end, speed):
    T = p.T; n = nvar(p); m = 8 * T; NX = 5 if p.nx == 5 else 4
    H = np.zeros((n, n)); g = np.zeros(n); G = np.zeros((m, n)); h = np.zeros(m)
    skip = np.zeros(m, dtype=np.uint8)
    fresp = np.zeros((NX, T + 1)); Sens = np.zeros((NX * (T + 1), n))
    st = lib().orc_build_qp(C.byref(p), _c(xref), _c(xbar), _c(x0), _c(reaches_end, np.uint8),
                            float(speed), H, g, G, h, skip, fresp, Sens)
    return int(st), H, g, G, h, skip, fresp, Sens


def solve_qp(H, g, G, h, skip=None):
    n = g.shape[0]; m = h.shape[0]
    if skip is None:
        skip = np.zeros(m, dtype=np.uint8)
    u = np.zeros(n); lam = np.zeros(m); it = C.c_int32(0)
    st = lib().orc_solve_qp(n, m, _c(H), _c(g), _c(G), _c(h), _c(skip, np.uint8), u, lam, C.byref(it))
    return int(st), u, lam, int(it.value)


def active_indices_from_mask(mask_row, m):
    return [i for i in range(m) if (int(mask_row[i >> 5]) >> (i & 31)) & 1]


def mpc_step(p, state_xyyawv, cx, cy, cyaw, target_ind, speed, oa=None, od=None, want_qp=False, cv=None, cv_cut=-1):
    """One reference MPC.step for one ego.  state = (x, y, yaw, v) like lib.simulation.State.
    cv / cv_cut: the mpc_with_speed variant's per-point speed reference, zeroed from index cv_cut on."""
    T = p.T; n = nvar(p); m = 8 * T
    res = {
        "oa": np.zeros(T), "od": np.zeros(T), "ox": np.zeros(T + 1), "oy": np.zeros(T + 1),
        "ov": np.zeros(T + 1), "oyaw": np.zeros(T + 1), "xref": np.zeros((4, T + 1)),
        "xbar": np.zeros((4, T + 1)), "idx": np.zeros(T + 1, dtype=np.int64),
        "reaches_end": np.zeros(T + 1, dtype=np.uint8), "lam": np.zeros(m),
        "active_mask": np.zeros((m + 31) // 32, dtype=np.uint32),
    }
    if want_qp:
        res["H"] = np.zeros((n, n)); res["g"] = np.zeros(n)
    out = OrcStepOut()
    for k, v in res.items():
        setattr(out, k, v.ctypes.data)
    cx, cy, cyaw = _c(cx), _c(cy), _c(cyaw)
    oa_c = _c(oa) if oa is not None else None
    od_c = _c(od) if od is not None else None
    sx, sy, syaw, sv = [float(v) for v in state_xyyawv]
    cv_c = _c(cv) if cv is not None else None
    st = lib().orc_mpc_step_cv(C.byref(p), sx, sy, syaw, sv, cx, cy, cyaw, cv_c.ctypes.data if cv_c is not None else None,
                               int(cv_cut), cx.shape[0], int(target_ind),
                               float(speed), oa_c.ctypes.data if oa_c is not None else None,
                               od_c.ctypes.data if od_c is not None else None, C.byref(out))
    res["status"] = int(st)
    res["n_iter"] = int(out.n_iter)
    res["target_ind"] = int(out.target_ind)
    res["reaches_end"] = res["reaches_end"].astype(bool)
    res["active"] = active_indices_from_mask(res["active_mask"], m)
    return res


def mpc_step_batch(p, x0, path_id, path_len, speed, cx, cy, cyaw, path_off, target_ind, oa, od,
                   n_threads=1, cv=None, cv_cut=None):
    """Batched oracle step with the product's [B][..] layouts.  Returns a dict of fresh arrays;
    oa/od/target_ind inputs are not modified."""
    B = x0.shape[0]; T = p.T; MW = (8 * T + 31) // 32
    x0 = _c(x0); path_id = _c(path_id, np.int32); path_len = _c(path_len, np.int32)
    speed = _c(speed); cx, cy, cyaw = _c(cx), _c(cy), _c(cyaw); path_off = _c(path_off, np.int64)
    out = {
        "target_ind": _c(target_ind, np.int64).copy(), "oa": _c(oa).copy(), "od": _c(od).copy(),
        "ox": np.zeros((B, T + 1)), "oy": np.zeros((B, T + 1)), "ov": np.zeros((B, T + 1)),
        "oyaw": np.zeros((B, T + 1)), "xref": np.zeros((B, 4, T + 1)),
        "active_mask": np.zeros((B, MW), dtype=np.uint32), "status": np.zeros(B, dtype=np.int32),
        "n_iter": np.zeros(B, dtype=np.int32),
    }
    ptr = lambda a: None if a is None else a.ctypes.data
    cv = _c(cv) if cv is not None else None
    cv_cut = _c(cv_cut, np.int32) if cv_cut is not None else None
    lib().orc_mpc_step_batch_cv(C.byref(p), B, ptr(x0), ptr(path_id), ptr(path_len), ptr(speed), ptr(cx),
                             ptr(cy), ptr(cyaw), ptr(cv), ptr(cv_cut), ptr(path_off), ptr(out["target_ind"]), ptr(out["oa"]),
                             ptr(out["od"]), ptr(out["ox"]), ptr(out["oy"]), ptr(out["ov"]),
                             ptr(out["oyaw"]), ptr(out["xref"]), ptr(out["active_mask"]),
                             ptr(out["status"]), ptr(out["n_iter"]), int(n_threads))
    return out


def closed_loop(p, state, cx, cy, cyaw, path_off, n_ticks, max_age=0, n_threads=1, record=True):
    """n_ticks closed-loop ticks (MPC.step -> plant -> goal / respawn) for every ego, IN PLACE on `state`, a dict of C-contiguous
    arrays: x0 [B,4], path_id, path_len (int32), speed, target_ind (int64), oa, od [B,T], di_ai [B,2], x0_spawn, target_spawn,
    age (int32).  Returns {"hist": [n_ticks,B,2] or None, "n_respawn", "n_iter_sum", "n_fail"}."""
    B = state["x0"].shape[0]
    hist = np.zeros((n_ticks, B, 2)) if record else None
    cnt = np.zeros(3, dtype=np.int64)
    cx, cy, cyaw = _c(cx), _c(cy), _c(cyaw); path_off = _c(path_off, np.int64)
    for k, dt in (("x0", np.float64), ("path_id", np.int32), ("path_len", np.int32), ("speed", np.float64),
                  ("target_ind", np.int64), ("oa", np.float64), ("od", np.float64), ("di_ai", np.float64),
                  ("x0_spawn", np.float64), ("target_spawn", np.int64), ("age", np.int32)):
        a = state[k]
        if not (isinstance(a, np.ndarray) and a.dtype == dt and a.flags.c_contiguous):
            raise ValueError(f"state[{k!r}] must be a C-contiguous {dt.__name__} array (it is updated in place)")
    ptr = lambda a: None if a is None else a.ctypes.data
    lib().orc_closed_loop(C.byref(p), B, int(n_ticks), ptr(state["x0"]), ptr(state["path_id"]), ptr(state["path_len"]),
                          ptr(state["speed"]), ptr(cx), ptr(cy), ptr(cyaw), ptr(path_off), ptr(state["target_ind"]),
                          ptr(state["oa"]), ptr(state["od"]), ptr(state["di_ai"]), ptr(state["x0_spawn"]),
                          ptr(state["target_spawn"]), ptr(state["age"]), int(max_age), ptr(hist),
                          cnt[0:].ctypes.data, cnt[1:].ctypes.data, cnt[2:].ctypes.data, int(n_threads))
    return {"hist": hist, "n_respawn": int(cnt[0]), "n_iter_sum": int(cnt[1]), "n_fail": int(cnt[2])}


def loop_state_from_batch(batch, T):
    """Fresh closed-loop state dict (see closed_loop) from a synth.EgoBatch."""
    B = batch.x0.shape[0]
    return {"x0": _c(batch.x0).copy(), "path_id": _c(batch.path_id, np.int32).copy(), "path_len": _c(batch.path_len, np.int32).copy(),
            "speed": _c(batch.speed).copy(), "target_ind": _c(batch.target_ind, np.int64).copy(), "oa": _c(batch.oa).copy(),
            "od": _c(batch.od).copy(), "di_ai": np.zeros((B, 2)), "x0_spawn": _c(batch.x0).copy(),
            "target_spawn": _c(batch.target_ind, np.int64).copy(), "age": np.zeros(B, dtype=np.int32)}


def xref_deviation(cx, cy, cyaw, target_ind, ox0, oy0):
    return float(lib().orc_xref_deviation(_c(cx), _c(cy), _c(cyaw), int(target_ind), float(ox0), float(oy0)))


def is_goal(p, sx, sy, sv, goal, target_ind, ncourse):
    return bool(lib().orc_is_goal(C.byref(p), float(sx), float(sy), float(sv), float(goal[0]),
                                  float(goal[1]), int(target_ind), int(ncourse)))
