"""ctypes binding of libjsim_mpc.so (include/jsim_mpc.h).  Thin: pointers + sizes only.

The product path has NO CPU fallback: if the HIP library is missing or cannot be loaded this module
raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

from .config import MPCConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JSIM_LIB_PATH") or os.path.join(_HERE, "libjsim_mpc.so")  # JSIM_LIB_PATH: A/B runs of diagnostic builds
ABI_VERSION = 2


class JsimCfg(C.Structure):
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


EXPORTS = (
    "jsim_abi_version", "jsim_last_error", "jsim_mpc_create", "jsim_mpc_destroy", "jsim_mpc_set_paths",
    "jsim_mpc_step", "jsim_mpc_step_debug", "jsim_plant_step", "jsim_loop_advance", "jsim_mpc_run_ticks", "jsim_mpc_set_launch_order", "jsim_mpc_get_launch_order", "jsim_mpc_iter_totals",
    "jsim_loop_set_geometry", "jsim_loop_set_obstacle_geometry", "jsim_loop_predict_obstacles", "jsim_loop_pre_tick",
    "jsim_mpc_set_path_speed", "jsim_mpc_set_speed_cutoff", "jsim_mpc_update_cfg", "jsim_mpc_set_ego_config", "jsim_loop_obstacles",
    "jsim_mpc_xref_deviation_goal", "jsim_loop_run_scenario",
    "jsim_comm_unique_id", "jsim_comm_init", "jsim_mpc_gather", "jsim_comm_destroy", "jsim_plan_routes",
)

_lib = None


class JsimError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library; raise if it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise JsimError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "The MPC path is HIP-only; there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    lib.jsim_abi_version.restype = C.c_int
    lib.jsim_abi_version.argtypes = []
    lib.jsim_last_error.restype = C.c_char_p
    lib.jsim_last_error.argtypes = [vp]
    lib.jsim_mpc_create.restype = C.c_int
    lib.jsim_mpc_create.argtypes = [C.POINTER(JsimCfg), C.c_int, C.POINTER(vp)]
    lib.jsim_mpc_destroy.restype = None
    lib.jsim_mpc_destroy.argtypes = [vp]
    lib.jsim_mpc_set_paths.restype = C.c_int
    lib.jsim_mpc_set_paths.argtypes = [vp, vp, vp, vp, vp, i32]
    lib.jsim_mpc_step.restype = C.c_int
    lib.jsim_mpc_step.argtypes = [vp, i32] + [vp] * 16
    lib.jsim_mpc_step_debug.restype = C.c_int
    lib.jsim_mpc_step_debug.argtypes = [vp, i32] + [vp] * 21
    lib.jsim_plant_step.restype = C.c_int
    lib.jsim_plant_step.argtypes = [vp, i32] + [vp] * 6
    lib.jsim_loop_advance.restype = C.c_int
    lib.jsim_loop_advance.argtypes = [vp, i32] + [vp] * 11 + [i32, vp, vp, i32, vp, vp]
    lib.jsim_loop_set_geometry.restype = C.c_int
    lib.jsim_loop_set_geometry.argtypes = [vp, dbl, dbl, dbl]
    lib.jsim_loop_set_obstacle_geometry.restype = C.c_int
    lib.jsim_loop_set_obstacle_geometry.argtypes = [vp, dbl, dbl, dbl, dbl]
    lib.jsim_loop_predict_obstacles.restype = C.c_int
    lib.jsim_loop_predict_obstacles.argtypes = [vp, i32, vp, i32, vp, vp]
    lib.jsim_loop_pre_tick.restype = C.c_int
    lib.jsim_loop_pre_tick.argtypes = [vp, i32] + [vp] * 9 + [i32, i32, vp, vp, vp]
    lib.jsim_mpc_set_path_speed.restype = C.c_int
    lib.jsim_mpc_set_path_speed.argtypes = [vp, vp]
    lib.jsim_mpc_set_speed_cutoff.restype = C.c_int
    lib.jsim_mpc_set_speed_cutoff.argtypes = [vp, vp]
    lib.jsim_mpc_update_cfg.restype = C.c_int
    lib.jsim_mpc_update_cfg.argtypes = [vp, C.POINTER(JsimCfg)]
    lib.jsim_mpc_set_ego_config.restype = C.c_int
    lib.jsim_mpc_set_ego_config.argtypes = [vp, vp]
    lib.jsim_loop_obstacles.restype = C.c_int
    lib.jsim_loop_obstacles.argtypes = [vp, i32, vp, vp, vp, i32, vp]
    lib.jsim_mpc_set_launch_order.restype = C.c_int
    lib.jsim_mpc_set_launch_order.argtypes = [vp, i32]
    lib.jsim_mpc_get_launch_order.restype = C.c_int
    lib.jsim_mpc_get_launch_order.argtypes = [vp, i32, vp, vp]
    lib.jsim_mpc_iter_totals.restype = C.c_int
    lib.jsim_mpc_iter_totals.argtypes = [vp, i32, vp, i32]
    lib.jsim_mpc_run_ticks.restype = C.c_int
    lib.jsim_mpc_run_ticks.argtypes = [vp, i32, i32] + [vp] * 19 + [i32, vp, vp, i32, vp, vp]
    lib.jsim_loop_run_scenario.restype = C.c_int
    #                                       ctx B    ticks  x0..n_iter,di_ai,x0_spawn,target_spawn,age  max_age hist tick cap  n_resp
    lib.jsim_loop_run_scenario.argtypes = ([vp, i32, i32] + [vp] * 19 + [i32, vp, vp, i32, vp] +
                                           # traj_idx prev_len col_flag pre_status  window margin n_obs state param get  n_steps speed_cutoff stream
                                           [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, i32, i32, vp])
    lib.jsim_mpc_xref_deviation_goal.restype = C.c_int
    lib.jsim_mpc_xref_deviation_goal.argtypes = [vp, i32] + [vp] * 9
    lib.jsim_comm_unique_id.restype = C.c_int
    lib.jsim_comm_unique_id.argtypes = [vp]
    lib.jsim_comm_init.restype = C.c_int
    lib.jsim_comm_init.argtypes = [vp, vp, i32, i32]
    lib.jsim_mpc_gather.restype = C.c_int
    lib.jsim_mpc_gather.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
    lib.jsim_comm_destroy.restype = C.c_int
    lib.jsim_comm_destroy.argtypes = [vp]
    lib.jsim_plan_routes.restype = C.c_int
    #                               dev  R    start goal box tol hp hp_off  n_obs  r_off mp_pts mp_len  n_prim n_pts cc  cc_off wh wc  max_path  outs
    lib.jsim_plan_routes.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, i32] + [vp] * 7
    if lib.jsim_abi_version() != ABI_VERSION:
        raise JsimError(f"libjsim_mpc.so ABI {lib.jsim_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def make_cfg(config: MPCConfig, T: int, dt: float, dl: float, L: float) -> JsimCfg:
    c = JsimCfg()
    c.T = int(T)
    c.max_iter = int(config.MAX_ITER)
    c.dt, c.dl, c.L = float(dt), float(dl), float(L)
    c.w_perp, c.w_para = float(config.w_perp), float(config.w_para)
    c.R[:] = [float(v) for v in config.R]
    c.Rd[:] = [float(v) for v in config.Rd]
    c.Q_v_yaw[:] = [float(v) for v in config.Q_v_yaw]
    c.Qf[:] = [float(v) for v in config.Qf]
    c.R_end[:] = [float(v) for v in config.R_END]
    c.max_dsteer = config.max_dsteer_rad
    c.max_accel = float(config.MAX_ACCEL)
    c.max_decel = float(config.MAX_DECEL)
    c.max_steer = float(config.MAX_STEER_RAD)
    c.max_speed = float(config.MAX_SPEED)
    c.min_speed = float(config.MIN_SPEED)
    c.min_ref_speed = float(config.MIN_REF_SPEED)
    c.goal_dis = float(config.GOAL_DIS)
    c.stop_speed = float(config.STOP_SPEED)
    c.nx = int(config.NX)
    c.jerk_weight = float(config.JERK_WEIGHT)
    return c


def check(rc: int, ctx=None, what: str = "jsim call") -> None:
    if rc != 0:
        msg = load().jsim_last_error(ctx)
        raise JsimError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
