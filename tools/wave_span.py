#!/usr/bin/env python3
"""Diagnostic: per-ego duration of one MPC step in the SHIPPED kernel code (build with -DJSIM_SPAN: two s_memtime
stamps per solve and nothing else), regressed on the active-set iteration count.  usage: wave_span.py [T] [B] [ticks]"""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
pkg = importlib.import_module("av-simulation-at-intersections_amd")

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lib = os.environ.get("JSIM_SPAN_LIB") or os.path.join(REPO, "av-simulation-at-intersections_amd", "libjsim_mpc_span.so")
if not os.environ.get("JSIM_SPAN_LIB"):   # (JSIM_SPAN_LIB: a -DJSIM_SPAN library built beforehand, e.g. where there is no GPU)
    subprocess.check_call([pkg.build._hipcc()] + pkg.build.HIPCC_FLAGS + ["-DJSIM_SPAN", "-I", pkg.build.INC, pkg.build.SRC, "-o", lib])
pkg._cabi.LIB_PATH = lib
pkg._cabi._lib = None
S = pkg.synth
routes = S.make_route_table(multi_lane=os.environ.get("JSIM_MULTI_LANE") == "1")
for r in routes:
    S.smooth_yaw_inplace(r[:, 2])
batch = S.make_ego_batch(routes, B, T, seed=1)
eng = pkg.BatchedMPC(routes, batch.path_id, dl=S.DL, T=T, speed=batch.speed, smooth=False)
eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
x0 = torch.from_numpy(batch.x0).cuda()
loop = pkg.ClosedLoop(eng, x0, max_age=400)
clk = torch.zeros(B * 24, dtype=torch.int64, device="cuda")
eng.lib.jsim_debug_set_clock_buffer.argtypes = [C.c_void_p, C.c_void_p]
eng.lib.jsim_debug_set_clock_buffer(eng._ctx, C.c_void_p(clk.data_ptr()))
span = np.zeros((K, B)); its = np.zeros((K, B)); okm = np.zeros((K, B), dtype=bool)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
wall = np.zeros(K)
for k in range(K):
    ev0.record(); loop.tick(); ev1.record()
    torch.cuda.synchronize()
    wall[k] = ev0.elapsed_time(ev1) * 1e3
    c = clk.cpu().numpy()[:B * 16].reshape(B, 16)
    span[k] = c[:, 11] - c[:, 0]
    its[k] = eng.n_iter.cpu().numpy()
    okm[k] = eng.status.cpu().numpy() == 0
sp, it = span[okm], its[okm]
A = np.vstack([np.ones_like(it), it]).T
coef, *_ = np.linalg.lstsq(A, sp, rcond=None)
print(f"T={T} B={B} ticks={K}: step span = {coef[0]:.0f} + {coef[1]:.0f} * n_iter  (s_memtime ticks); mean n_iter {it.mean():.2f}, "
      f"mean span {sp.mean():.0f}, p99 {np.percentile(sp, 99):.0f}, max {sp.max():.0f}")
per_tick_max = np.where(okm, span, 0).max(axis=1)
print(f"  per tick: mean over egos {np.where(okm, span, 0).sum() / okm.sum():.0f}, mean of per-tick max {per_tick_max.mean():.0f}; "
      f"single-tick launch wall {np.median(wall):.1f} us -> ticks per us = {np.median(per_tick_max / wall):.1f}")
ego_sum = np.where(okm, span, 0).sum(axis=0)
print(f"  per ego over {K} ticks: mean of sums {ego_sum.mean():.0f}, max of sums {ego_sum.max():.0f} (ratio {ego_sum.max() / ego_sum.mean():.3f}); "
      f"n_iter per ego: mean {its.mean():.2f}, max of per-ego means {its.mean(axis=0).max():.2f}")
