"""BatchedMPC: B independent egos stepped by one HIP launch (one wavefront per ego).

Host side of the reference's `MPC` controller (main/lib/mpc.py:245-330) for a batch.  PyTorch-ROCm
tensors are only the device-memory containers whose `data_ptr()` is handed to the C-ABI
(include/jsim_mpc.h); all arithmetic of the step happens in csrc/jsim_mpc.hip.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import _cabi
from .config import MPCConfig
from .synth import pack_paths, smooth_yaw_inplace


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class BatchedMPC:
    """paths: list of (M, 3) float64 arrays [x, y, yaw] (a shared route table);
    path_id[b] selects ego b's route.  Like `MPC.__init__` (main/lib/mpc.py:260) the yaw column of
    each path is unwrapped IN PLACE unless smooth=False.

    Controller state kept resident on the device across ticks: target_ind [B], warm start oa/od [B, T],
    per-ego truncated path length path_len [B] (the batched form of `set_trajectory_fromarray`).
    """

    def __init__(self, paths: Sequence[np.ndarray], path_id: Union[np.ndarray, Sequence[int]], dl: float,
                 L: float = 2.86, speed: Union[float, np.ndarray] = 30 / 3.6, dt: float = 0.2,
                 T: Optional[int] = None, config: Optional[MPCConfig] = None,
                 device: Union[str, torch.device] = "cuda:0", smooth: bool = True,
                 cv: Optional[Sequence[np.ndarray]] = None):
        self.lib = _cabi.load()  # raises if the HIP library is missing: no fallback
        if not torch.cuda.is_available():
            raise _cabi.JsimError("BatchedMPC needs a HIP device (torch.cuda.is_available() is False); "
                                  "the MPC path has no CPU fallback")
        self.config = config or MPCConfig.from_json()
        self.T = int(T if T is not None else self.config.T)
        self.dl, self.dt, self.L = float(dl), float(dt), float(L)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _cabi.JsimError(f"device must be a HIP device, got {self.device}")
        self.dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.paths = [np.asarray(p, dtype=np.float64) for p in paths]
        for p in self.paths:
            if p.ndim != 2 or p.shape[1] != 3 or p.shape[0] < 1:
                raise ValueError("each path must be an (M, 3) array [x, y, yaw] with M >= 1")
            if smooth:
                smooth_yaw_inplace(p[:, 2])
        cfg = _cabi.make_cfg(self.config, self.T, self.dt, self.dl, self.L)
        self._cfg = cfg
        self._ctx = C.c_void_p()
        _cabi.check(self.lib.jsim_mpc_create(C.byref(cfg), self.dev_index, C.byref(self._ctx)), None, "jsim_mpc_create")
        self._upload_paths()
        # mpc_with_speed variant: per-point speed reference (one array per path), xref[2] = cv[idx]
        self.cv = None if cv is None else [np.ascontiguousarray(c, dtype=np.float64) for c in cv]
        self.cv_cut = None
        if self.cv is not None:
            if len(self.cv) != len(self.paths) or any(len(c) != len(p) for c, p in zip(self.cv, self.paths)):
                raise ValueError("cv must hold one speed array per path, of the path's length")
            flat = np.concatenate(self.cv)
            _cabi.check(self.lib.jsim_mpc_set_path_speed(self._ctx, flat.ctypes.data), self._ctx, "jsim_mpc_set_path_speed")

        pid = np.ascontiguousarray(path_id, dtype=np.int32)
        if pid.ndim != 1 or (pid.size and (pid.min() < 0 or pid.max() >= len(self.paths))):
            raise ValueError("path_id must be a 1-D array of indices into paths")
        self.B = int(pid.shape[0])
        B, T = self.B, self.T
        dev = self.device
        self.path_id = torch.from_numpy(pid).to(dev)
        full = np.array([self.paths[i].shape[0] for i in pid], dtype=np.int32)
        self.path_len = torch.from_numpy(full.copy()).to(dev)
        self.full_len = torch.from_numpy(full).to(dev)
        sp = np.broadcast_to(np.asarray(speed, dtype=np.float64), (B,)).copy()
        self.speed = torch.from_numpy(sp).to(dev)
        # controller state + step outputs: typed views of ONE device allocation, so that a caller who wants everything on the
        # host (the single-ego drop-in) pays one device->host copy per step instead of one per array
        MW = (8 * T + 31) // 32
        spec = [("target_ind", torch.int64, (B,)), ("oa", torch.float64, (B, T)), ("od", torch.float64, (B, T)),
                ("ox", torch.float64, (B, T + 1)), ("oy", torch.float64, (B, T + 1)), ("ov", torch.float64, (B, T + 1)),
                ("oyaw", torch.float64, (B, T + 1)), ("xref", torch.float64, (B, 4, T + 1)), ("di_ai", torch.float64, (B, 2)),
                ("active_mask", torch.int32, (B, MW)), ("status", torch.int32, (B,)), ("n_iter", torch.int32, (B,))]
        self._arena_layout = {}
        nbytes = 0
        for name, dt, shape in spec:
            n = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            self._arena_layout[name] = (nbytes, n, dt, shape)
            nbytes += (n + 7) & ~7
        self._arena = torch.zeros(max(nbytes, 8), dtype=torch.uint8, device=dev)   # oa/od zeros = the reference's None (mpc.py:225-227); di = ai = 0 (:274-275)
        for name, (o, n, dt, shape) in self._arena_layout.items():
            setattr(self, name, self._arena[o:o + n].view(dt).view(shape))

    def read_back(self) -> dict:
        """Everything the step wrote, on the host, with ONE synchronising device->host copy: a dict of numpy views
        (target_ind, oa, od, ox, oy, ov, oyaw, xref, di_ai, active_mask, status, n_iter)."""
        host = self._arena.cpu().numpy()
        return {name: host[o:o + n].view(torch.empty((), dtype=dt).numpy().dtype).reshape(shape)
                for name, (o, n, dt, shape) in self._arena_layout.items()}

    # ------------------------------------------------------------------ paths
    def _upload_paths(self):
        cx, cy, cyaw, off = pack_paths(self.paths)
        self._path_off = off
        _cabi.check(self.lib.jsim_mpc_set_paths(self._ctx, cx.ctypes.data, cy.ctypes.data, cyaw.ctypes.data,
                                                off.ctypes.data, len(self.paths)), self._ctx, "jsim_mpc_set_paths")

    def set_path_len(self, path_len: Union[np.ndarray, torch.Tensor]):
        """Batched `set_trajectory_fromarray(trajectory_full[:cutoff])` (main/lib/mpc.py:279-282,
        main/scenarios/mpc_intersection.py:129-143): ego b now sees the first path_len[b] points."""
        if isinstance(path_len, torch.Tensor):
            pl = path_len.to(device=self.device, dtype=torch.int32)
        else:
            pl = torch.from_numpy(np.ascontiguousarray(path_len, dtype=np.int32)).to(self.device)
        if pl.shape != (self.B,):
            raise ValueError("path_len must have shape [B]")
        if bool((pl < 1).any()) or bool((pl > self.full_len).any()):
            raise ValueError("path_len must satisfy 1 <= path_len[b] <= len(paths[path_id[b]])")
        self.path_len = pl.contiguous()

    def set_speed_cutoff(self, cutoff: Optional[Union[np.ndarray, torch.Tensor]]):
        """Batched `set_trajectory_fromarray(trajectory, cutoff_idx)` of the mpc_with_speed variant
        (main/lib/mpc_with_speed.py:276-282): ego b's speed reference is 0 from path index cutoff[b] on (< 0: nowhere)."""
        if cutoff is None:
            self.cv_cut = None
            _cabi.check(self.lib.jsim_mpc_set_speed_cutoff(self._ctx, None), self._ctx, "jsim_mpc_set_speed_cutoff")
            return
        t = cutoff if isinstance(cutoff, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(cutoff, dtype=np.int32))
        self.cv_cut = t.to(device=self.device, dtype=torch.int32).contiguous()
        if self.cv_cut.shape != (self.B,):
            raise ValueError("cutoff must have shape [B]")
        _cabi.check(self.lib.jsim_mpc_set_speed_cutoff(self._ctx, _ptr(self.cv_cut)), self._ctx, "jsim_mpc_set_speed_cutoff")

    def update_config(self, config: MPCConfig):
        """New weights / limits for the following solves (what main/lib/mpc_sensitivity.py does by re-reading its JSON in
        every solve, :153-166).  The horizon cannot change."""
        cfg = _cabi.make_cfg(config, self.T, self.dt, self.dl, self.L)
        _cabi.check(self.lib.jsim_mpc_update_cfg(self._ctx, C.byref(cfg)), self._ctx, "jsim_mpc_update_cfg")
        self.config, self._cfg = config, cfg

    EGO_CFG_DOUBLES = 16

    def set_ego_configs(self, configs):
        """Per-ego weights and limits (jsim_mpc_set_ego_config): `configs` is a sequence of B MPCConfig objects -- e.g. the
        parameter sets of a sensitivity sweep (main/scenarios/mpc_sensitivity_analysis*.py), solved as ONE batch -- or an
        array [B, 16] in the layout of include/jsim_mpc.h, or None to go back to the context's configuration.
        T, dt, dl, L, R_end, the speed limits and the goal test stay those of the engine."""
        if configs is None:
            self._pe = None
            _cabi.check(self.lib.jsim_mpc_set_ego_config(self._ctx, None), self._ctx, "jsim_mpc_set_ego_config")
            return
        if isinstance(configs, (np.ndarray, torch.Tensor)):
            arr = configs if isinstance(configs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(configs, dtype=np.float64))
        else:
            rows = []
            for c in configs:
                if c.T != self.T:
                    raise ValueError("per-ego configurations share the engine's horizon")
                rows.append([c.w_perp, c.w_para, c.R[0], c.R[1], c.Rd[0], c.Rd[1], c.Q_v_yaw[0], c.Q_v_yaw[1],
                             c.Qf[0], c.Qf[1], c.Qf[2], c.Qf[3], c.max_dsteer_rad, c.MAX_ACCEL, c.MAX_DECEL, 0.0])
            arr = torch.tensor(rows, dtype=torch.float64)
        if tuple(arr.shape) != (self.B, self.EGO_CFG_DOUBLES):
            raise ValueError(f"need [{self.B}, {self.EGO_CFG_DOUBLES}] per-ego parameters, got {tuple(arr.shape)}")
        self._pe = arr.to(device=self.device, dtype=torch.float64).contiguous()
        _cabi.check(self.lib.jsim_mpc_set_ego_config(self._ctx, C.c_void_p(self._pe.data_ptr())), self._ctx,
                    "jsim_mpc_set_ego_config")

    def load_state(self, target_ind=None, oa=None, od=None, path_len=None):
        """Overwrite the resident controller state (host arrays or tensors): remembered path index, warm
        start, truncated path lengths."""
        def put(dst, src):
            src = src if isinstance(src, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(src))
            dst.copy_(src.to(device=self.device, dtype=dst.dtype))
        if target_ind is not None:
            put(self.target_ind, target_ind)
        if oa is not None:
            put(self.oa, oa)
        if od is not None:
            put(self.od, od)
        if path_len is not None:
            self.set_path_len(path_len)

    # ------------------------------------------------------------------ one tick
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_x0(self, x0: torch.Tensor):
        if not (isinstance(x0, torch.Tensor) and x0.is_cuda and x0.dtype == torch.float64
                and x0.shape == (self.B, 4) and x0.is_contiguous()):
            raise ValueError("x0 must be a contiguous float64 device tensor [B, 4] = (x, y, v, yaw)")

    def solve(self, x0: torch.Tensor, debug: Optional[dict] = None):
        """Launch the MPC step for all egos (asynchronous on the current stream).  Updates oa, od, ox, oy,
        ov, oyaw, xref, target_ind, active_mask, status, n_iter in place.  `debug` may hold device tensors
        'xbar' [B,4,T+1], 'ref_idx' [B,T+1] int64, 'H' [B,2T,2T], 'g' [B,2T], 'lam' [B,8T]."""
        self._check_x0(x0)
        args = [self._ctx, self.B, _ptr(x0), _ptr(self.path_id), _ptr(self.path_len), _ptr(self.speed),
                _ptr(self.target_ind), _ptr(self.oa), _ptr(self.od), _ptr(self.ox), _ptr(self.oy), _ptr(self.ov),
                _ptr(self.oyaw), _ptr(self.xref), _ptr(self.active_mask), _ptr(self.status), _ptr(self.n_iter)]
        if debug is None:
            rc = self.lib.jsim_mpc_step(*args, self._stream())
        else:
            rc = self.lib.jsim_mpc_step_debug(*args, _ptr(debug.get("xbar")), _ptr(debug.get("ref_idx")),
                                              _ptr(debug.get("H")), _ptr(debug.get("g")), _ptr(debug.get("lam")),
                                              self._stream())
        _cabi.check(rc, self._ctx, "jsim_mpc_step")

    def step(self, x0: torch.Tensor):
        """`MPC.step` for the batch: returns (di[B], ai[B]) = (steer, accel), steer first like the reference
        (main/lib/mpc.py:303).  Failed egos get ai = MAX_DECEL and keep their previous di (:298-301)."""
        self.solve(x0)
        ok = self.status == 0
        di = torch.where(ok, self.od[:, 0], self.di_ai[:, 0])
        ai = torch.where(ok, self.oa[:, 0], torch.full_like(di, float(self.config.MAX_DECEL)))
        self.di_ai[:, 0] = di
        self.di_ai[:, 1] = ai
        return di, ai

    def step_and_advance(self, x0: torch.Tensor):
        """One tick of the per-vehicle loop without leaving the device: MPC step, (di, ai) selection with the
        failure path, then `Simulation.step` on x0 IN PLACE (main/scenarios/mpc_intersection.py:146,163)."""
        self.solve(x0)
        _cabi.check(self.lib.jsim_plant_step(self._ctx, self.B, _ptr(x0), _ptr(self.oa), _ptr(self.od),
                                             _ptr(self.status), _ptr(self.di_ai), self._stream()),
                    self._ctx, "jsim_plant_step")
        return self.di_ai

    def xref_deviation_and_goal(self, x0: torch.Tensor):
        """Batched `get_current_xref_deviation` / `is_goal` (main/lib/mpc.py:305-330)."""
        self._check_x0(x0)
        dev = torch.empty(self.B, dtype=torch.float64, device=self.device)
        goal = torch.empty(self.B, dtype=torch.int32, device=self.device)
        _cabi.check(self.lib.jsim_mpc_xref_deviation_goal(
            self._ctx, self.B, _ptr(x0), _ptr(self.path_id), _ptr(self.path_len), _ptr(self.target_ind),
            _ptr(self.ox), _ptr(self.oy), _ptr(dev), _ptr(goal), self._stream()), self._ctx,
            "jsim_mpc_xref_deviation_goal")
        return dev, goal.bool()

    def active_indices(self, b: int) -> List[int]:
        """Active-constraint indices of ego b in the canonical row order (include/jsim_mpc.h)."""
        words = self.active_mask[b].cpu().numpy().view(np.uint32)
        return [i for i in range(8 * self.T) if (int(words[i >> 5]) >> (i & 31)) & 1]

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self.lib.jsim_mpc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
