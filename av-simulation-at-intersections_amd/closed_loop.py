"""Closed-loop driver for a batch of egos: the reference's per-vehicle loop
(main/scenarios/mpc_intersection.py:99-163) with the B-ego `for` replaced by one MPC launch + one
bookkeeping launch per tick, nothing leaving the device between ticks."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

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
