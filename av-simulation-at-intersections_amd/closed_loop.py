"""Closed-loop driver for a batch of egos: the reference's per-vehicle loop
(main/scenarios/mpc_intersection.py:99-163) with the B-ego `for` replaced by one MPC launch + one
bookkeeping launch per tick, nothing leaving the device between ticks."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

import math

from . import _cabi
from .batched import BatchedMPC, _ptr


class ClosedLoop:
    """tick(): jsim_mpc_step then jsim_loop_advance (plant, history, respawn of finished egos).

    hist_cap: ticks of (di, ai) history kept on the device ([hist_cap, B, 2]); max_age: safety respawn
    after that many ticks (<= 0: only MPC.is_goal ends an ego's run)."""

    def __init__(self, engine: BatchedMPC, x0: torch.Tensor, hist_cap: int = 0, max_age: int = 0):
        self.eng = engine
        eng = engine
        eng._check_x0(x0)
        self.x0 = x0
        dev = eng.device
        self.x0_spawn = x0.clone()
        self.target_spawn = eng.target_ind.clone()
        self.age = torch.zeros(eng.B, dtype=torch.int32, device=dev)
        self.max_age = int(max_age)
        self.hist_cap = int(hist_cap)
        self.hist = torch.zeros(max(hist_cap, 1), eng.B, 2, dtype=torch.float64, device=dev) if hist_cap > 0 else None
        self.tick_counter = torch.zeros(1, dtype=torch.int32, device=dev)
        self.n_respawn = torch.zeros(1, dtype=torch.int64, device=dev)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._graph_ticks = 0

    def tick(self):
        eng = self.eng
        eng.solve(self.x0)
        _cabi.check(eng.lib.jsim_loop_advance(
            eng._ctx, eng.B, _ptr(self.x0), _ptr(eng.oa), _ptr(eng.od), _ptr(eng.status), _ptr(eng.di_ai),
            _ptr(eng.target_ind), _ptr(eng.path_id), _ptr(eng.path_len), _ptr(self.x0_spawn),
            _ptr(self.target_spawn), _ptr(self.age), self.max_age, _ptr(self.hist), _ptr(self.tick_counter),
            self.hist_cap, _ptr(self.n_respawn), eng._stream()), eng._ctx, "jsim_loop_advance")

    def run(self, n_ticks: int):
        """n_ticks ticks in one call (jsim_mpc_run_ticks): for T = 13 / 20 a single launch in which each wavefront
        advances its own ego n_ticks times -- same results as n_ticks x tick(), without per-tick synchronisation."""
        eng = self.eng
        eng._check_x0(self.x0)
        _cabi.check(eng.lib.jsim_mpc_run_ticks(
            eng._ctx, eng.B, int(n_ticks), _ptr(self.x0), _ptr(eng.path_id), _ptr(eng.path_len), _ptr(eng.speed),
            _ptr(eng.target_ind), _ptr(eng.oa), _ptr(eng.od), _ptr(eng.ox), _ptr(eng.oy), _ptr(eng.ov), _ptr(eng.oyaw),
            _ptr(eng.xref), _ptr(eng.active_mask), _ptr(eng.status), _ptr(eng.n_iter), _ptr(eng.di_ai),
            _ptr(self.x0_spawn), _ptr(self.target_spawn), _ptr(self.age), self.max_age, _ptr(self.hist),
            _ptr(self.tick_counter), self.hist_cap, _ptr(self.n_respawn), eng._stream()), eng._ctx,
            "jsim_mpc_run_ticks")

    # ---- hipGraph: a launch-bound inner loop (two short kernels per tick) replayed without host work
    def capture(self, ticks: int):
        """Capture `ticks` consecutive ticks into one hipGraph (all pointers are fixed device buffers and the
        history slot comes from the device tick counter, so replays continue the same simulation)."""
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=self.eng.device)
        s.wait_stream(torch.cuda.current_stream(self.eng.device))
        with torch.cuda.stream(s):
            self.tick()                      # warm-up outside capture (lazy module load)
        torch.cuda.current_stream(self.eng.device).wait_stream(s)
        torch.cuda.synchronize(self.eng.device)
        with torch.cuda.graph(g, stream=s):
            for _ in range(ticks):
                self.tick()
        self._graph, self._graph_ticks = g, ticks
        return g

    def replay(self):
        self._graph.replay()
        return self._graph_ticks


def car_circles(L: float = 2.86, width: float = 2.0, extra_length: float = 0.64):
    """Collision circles of the reference's BicycleModelDimensions (main/lib/car_dimensions.py:62-79,82-90): bounding
    box (width, L + 0.64), radius = width / sqrt(2), two centres on the body axis at L/2 +- (length/2 - width/2) from the
    rear axle.  Returns (radius, (front_offset, rear_offset))."""
    length = L + extra_length
    offset = length / 2 - width / 2
    return width / (2 ** .5), (L / 2 + offset, L / 2 - offset)


class PreTick:
    """The loop glue ahead of MPC.step (main/scenarios/mpc_intersection.py:104-143) for the whole batch: progress index,
    ego-path resampling, obstacle prediction, collision check, cut-off -> writes the engine's `path_len`, i.e. the batched
    `mpc.set_trajectory_fromarray(trajectory_full[:cutoff_idx])`.  Obstacles are shared by all egos of the batch.

    mode = "speed_cutoff" is the glue of main/scenarios/mpc_intersection_new_ref.py (:122-139, FRAME_WINDOW = 20 there): the
    path is never truncated, the cut-off index goes to the mpc_with_speed controller instead
    (`mpc.set_trajectory_fromarray(trajectory_full, cutoff_idx=...)`: the speed reference is zeroed from it on) -- the
    engine must carry a speed reference (`cv`)."""

    def __init__(self, engine: BatchedMPC, frame_window: int = 10, time_horizon: float = 7.0, car_width: float = 2.0,
                 extra_length: float = 0.64, mode: str = "truncate", obstacle_dims: Optional[dict] = None,
                 margin_factor: int = 4):
        """obstacle_dims = dict(L=1.0, width=0.45, extra_length=0.64) gives the obstacles their own shape (the cyclist's
        BicycleRealDimensions of main/scenarios/overtaking_cyclist_bidirectional_road.py: prediction with that wheelbase,
        `check_collision_moving_bicycle`); margin_factor: EXTRA_CUTOFF_MARGIN = margin_factor * ceil(radius / dl) (4 in
        mpc_intersection.py:88-89, 2 in the cyclist scenario :94-95)."""
        if mode not in ("truncate", "speed_cutoff"):
            raise ValueError("mode must be 'truncate' or 'speed_cutoff'")
        if mode == "speed_cutoff" and engine.cv is None:
            raise ValueError("mode='speed_cutoff' needs an engine with a speed reference (cv)")
        self.mode = mode
        self.eng = engine
        eng = engine
        self.radius, (c0, c1) = car_circles(eng.L, car_width, extra_length)
        self.frame_window = int(frame_window)
        self.n_steps = int(math.ceil(time_horizon / eng.dt - 1e-9))          # len(np.arange(0, horizon, dt))
        self.margin = int(margin_factor) * int(math.ceil(self.radius / eng.dl))   # EXTRA_CUTOFF_MARGIN, :88-89
        _cabi.check(eng.lib.jsim_loop_set_geometry(eng._ctx, c0, c1, self.radius), eng._ctx, "jsim_loop_set_geometry")
        if obstacle_dims is not None:
            oL = float(obstacle_dims["L"])
            orad, (o0, o1) = car_circles(oL, float(obstacle_dims.get("width", 2.0)), float(obstacle_dims.get("extra_length", 0.64)))
            _cabi.check(eng.lib.jsim_loop_set_obstacle_geometry(eng._ctx, o0, o1, orad, oL), eng._ctx,
                        "jsim_loop_set_obstacle_geometry")
        dev, B = eng.device, eng.B
        self.traj_idx = torch.zeros(B, dtype=torch.int64, device=dev)
        self.prev_len = torch.full((B,), -1, dtype=torch.int32, device=dev)   # tmp_trajectory is None
        self.col_flag = torch.zeros(B, dtype=torch.int32, device=dev)
        self.col_xy = torch.zeros(B, 2, dtype=torch.float64, device=dev)
        self.first_idx = torch.zeros(B, dtype=torch.int32, device=dev)
        self.status = torch.zeros(B, dtype=torch.int32, device=dev)
        self.pred = None
        self.n_obs = 0
        self.cut = None
        if mode == "speed_cutoff":
            self.full_len = eng.path_len.clone()
            self.cut = eng.path_len.clone()              # = "no cut-off" (the reference's 999)
            eng.set_speed_cutoff(self.cut)
            self.cut = eng.cv_cut                        # the buffer the controller reads; updated in place every tick
        self.predict(torch.zeros(0, 6, dtype=torch.float64, device=dev))

    def predict(self, obst: torch.Tensor):
        """obst: device float64 [n_obs, 6] = (x, y, v, yaw, a, steer) per obstacle, as MovingObstacle*.get() returns."""
        eng = self.eng
        if not (obst.is_cuda and obst.dtype == torch.float64 and obst.dim() == 2 and obst.shape[1] == 6 and obst.is_contiguous()):
            raise ValueError("obst must be a contiguous float64 device tensor [n_obs, 6]")
        self.n_obs = int(obst.shape[0])
        self.pred = torch.zeros(max(self.n_obs, 1), self.n_steps, 3, dtype=torch.float64, device=eng.device)
        _cabi.check(eng.lib.jsim_loop_predict_obstacles(eng._ctx, self.n_obs, _ptr(obst) if self.n_obs else None,
                                                        self.n_steps, _ptr(self.pred), eng._stream()), eng._ctx,
                    "jsim_loop_predict_obstacles")
        return self.pred[: self.n_obs]

    def run(self, x0: torch.Tensor, debug: Optional[dict] = None):
        """Updates traj_idx and the engine's path_len for this tick (then remembers it as the previous truncated path)."""
        eng = self.eng
        eng._check_x0(x0)
        dbg_idx = dbg_n = None
        if debug is not None:
            dbg_idx, dbg_n = debug["res_idx"], debug["n_res"]
        out = eng.path_len if self.mode == "truncate" else self.cut
        _cabi.check(eng.lib.jsim_loop_pre_tick(
            eng._ctx, eng.B, _ptr(x0), _ptr(eng.path_id), _ptr(self.traj_idx), _ptr(self.prev_len), _ptr(out),
            _ptr(self.col_flag), _ptr(self.col_xy), _ptr(self.first_idx), _ptr(self.status), self.frame_window,
            self.margin, _ptr(dbg_idx), _ptr(dbg_n), eng._stream()), eng._ctx, "jsim_loop_pre_tick")
        # the previous tmp_trajectory: the truncated path, or always the full one (mpc_intersection_new_ref.py:131)
        self.prev_len.copy_(eng.path_len if self.mode == "truncate" else self.full_len)


class ScriptedObstacles:
    """Device-resident scripted obstacle vehicles of main/lib/moving_obstacles.py.
    specs: list of dicts like the scenarios build them (main/scenarios/mpc_intersection.py:46-49):
      kind="t_intersection" (default) | "roundabout": direction=+-1, turning=bool, speed=float, offset=float|None
      kind="arterial": x_init, y_init, speed, initial_speed, offset=float|None   (drives straight up)"""

    KINDS = {"t_intersection": 0.0, "roundabout": 1.0, "arterial": 2.0}

    def __init__(self, engine: BatchedMPC, specs, dt: Optional[float] = None):
        self.eng = engine
        dt = engine.dt if dt is None else dt
        st, pr = [], []
        for s in specs:
            kind = s.get("kind", "t_intersection")
            if kind not in self.KINDS:
                raise ValueError(f"unknown obstacle kind {kind!r}")
            off = s.get("offset", None)
            off = float(off) if (off is not None and off > 0) else 0.0
            if kind == "arterial":
                st.append([float(s["x_init"]), float(s["y_init"]), math.pi / 2, 0.0])
                pr.append([1.0, 0.0, float(s["speed"]), off, 0.0, float(dt), self.KINDS[kind], float(s["initial_speed"])])
                continue
            d = 1.0 if s.get("direction", 1) >= 0 else -1.0
            if d > 0:
                st.append([-30.0, -3.0, 0.0, 0.0]); x_turn = -10.0
            else:
                st.append([30.0, 3.0, math.pi, 0.0]); x_turn = 12.0
            pr.append([d, 1.0 if s.get("turning", False) else 0.0, float(s["speed"]), off, x_turn, float(dt), self.KINDS[kind], 0.0])
        dev = engine.device
        self.n = len(specs)
        self.state = torch.tensor(st, dtype=torch.float64, device=dev).reshape(self.n, 4)
        self.param = torch.tensor(pr, dtype=torch.float64, device=dev).reshape(self.n, 8)
        self.get_buf = torch.zeros(max(self.n, 1), 6, dtype=torch.float64, device=dev)
        self._state0 = self.state.clone()

    def reset(self):
        """Send the vehicles in again from their start poses (asynchronous device copy on the current stream)."""
        self.state.copy_(self._state0)

    def get(self, step: bool = False) -> torch.Tensor:
        """The `o.get()` tuples of all obstacles ([n, 6] device tensor); step=True also applies `o.step()` afterwards."""
        eng = self.eng
        _cabi.check(eng.lib.jsim_loop_obstacles(eng._ctx, self.n, _ptr(self.state), _ptr(self.param), _ptr(self.get_buf),
                                                1 if step else 0, eng._stream()), eng._ctx, "jsim_loop_obstacles")
        return self.get_buf[: self.n]


class ScenarioLoop:
    """The reference scenario loop for a batch, entirely on the device (main/scenarios/mpc_intersection.py:99-163):
    obstacle get() -> prediction -> progress index / resample / collision / cut-off -> MPC.step -> plant, history, goal ->
    obstacle step()."""

    def __init__(self, engine: BatchedMPC, x0: torch.Tensor, obstacle_specs, hist_cap: int = 0, max_age: int = 0,
                 frame_window: int = 10, mode: str = "truncate"):
        self.loop = ClosedLoop(engine, x0, hist_cap=hist_cap, max_age=max_age)
        self.pre = PreTick(engine, frame_window=frame_window, mode=mode)
        self.obst = ScriptedObstacles(engine, obstacle_specs)

    def tick(self):
        g = self.obst.get(step=False)
        self.pre.predict(g)
        self.pre.run(self.loop.x0)
        self.loop.tick()
        resp = self.loop.age == 0                    # respawned this tick: the glue starts over like for a new run
        self.pre.traj_idx.masked_fill_(resp, 0)
        self.pre.prev_len.masked_fill_(resp, -1)
        self.obst.get(step=True)

    def run(self, n_ticks: int):
        """n_ticks ticks in one call (jsim_loop_run_scenario).  With a register kernel (config.ONE_WAVE_HORIZONS / FOUR_WAVE_HORIZONS) and MAX_ITER = 1
        three launches -- the scripted obstacles rolled forward n_ticks ticks, their predictions for every
        tick, and one fused launch in which each ego's wavefront does glue + solve + plant n_ticks times.  Same results as
        n_ticks x tick()."""
        loop, pre, ob, eng = self.loop, self.pre, self.obst, self.loop.eng
        eng._check_x0(loop.x0)
        _cabi.check(eng.lib.jsim_loop_run_scenario(
            eng._ctx, eng.B, int(n_ticks), _ptr(loop.x0), _ptr(eng.path_id), _ptr(eng.path_len), _ptr(eng.speed),
            _ptr(eng.target_ind), _ptr(eng.oa), _ptr(eng.od), _ptr(eng.ox), _ptr(eng.oy), _ptr(eng.ov), _ptr(eng.oyaw),
            _ptr(eng.xref), _ptr(eng.active_mask), _ptr(eng.status), _ptr(eng.n_iter), _ptr(eng.di_ai),
            _ptr(loop.x0_spawn), _ptr(loop.target_spawn), _ptr(loop.age), loop.max_age, _ptr(loop.hist),
            _ptr(loop.tick_counter), loop.hist_cap, _ptr(loop.n_respawn), _ptr(pre.traj_idx), _ptr(pre.prev_len),
            _ptr(pre.col_flag), _ptr(pre.status), pre.frame_window, pre.margin, ob.n, _ptr(ob.state) if ob.n else None,
            _ptr(ob.param) if ob.n else None, _ptr(ob.get_buf) if ob.n else None, pre.n_steps,
            1 if pre.mode == "speed_cutoff" else 0, eng._stream()), eng._ctx,
            "jsim_loop_run_scenario")
        pre.n_obs = ob.n
