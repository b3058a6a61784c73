#!/usr/bin/env python3
"""Test infrastructure (run by hand on an MI355X; not collected by pytest): the scenario loop with scripted obstacles on the
states it actually visits.  Every tick the numpy restatement of the loop glue (oracle/loop_oracle.py: progress index, resample,
prediction, first collision, cut-off) and the oracle's MPC step are given the device's inputs and compared with what the device
did: progress index, path length and collision flag bit-exact, statuses / target indices / active sets identical, controls
within 1e-7.  Then the device advances (plant, goal / respawn, obstacle step).

    python tests/soak_scenario.py [B=96] [ticks=100] [T=20]
"""
import importlib
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
pkg = importlib.import_module("av-simulation-at-intersections_amd")
import oracle_py as O  # noqa: E402
import loop_oracle as LO  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 96
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
T = int(sys.argv[3]) if len(sys.argv) > 3 else 20
S = pkg.synth
routes = S.make_route_table()
for r in routes:
    S.smooth_yaw_inplace(r[:, 2])
cx, cy, cyaw, off = S.pack_paths(routes)
batch = S.make_ego_batch(routes, B, T, seed=23)
eng = pkg.BatchedMPC(routes, batch.path_id, dl=S.DL, T=T, speed=batch.speed, smooth=False)
eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
specs = [dict(direction=1, turning=False, speed=25 / 3.6, offset=None), dict(direction=-1, turning=True, speed=20 / 3.6, offset=2.0),
         dict(direction=1, turning=True, speed=15 / 3.6, offset=5.0)]
sc = pkg.ScenarioLoop(eng, torch.from_numpy(batch.x0).cuda(), specs, max_age=60)
p = O.make_params(T=T)
n_glue = n_st = n_ti = n_mk = n_col = n_cut = 0
worst = 0.0
for k in range(K):
    g = sc.obst.get(step=False)
    obst = g.cpu().numpy().copy()
    x0 = sc.loop.x0.cpu().numpy().copy()
    tidx = sc.pre.traj_idx.cpu().numpy().copy(); prev = sc.pre.prev_len.cpu().numpy().copy()
    plen_before = eng.path_len.cpu().numpy().copy()
    sc.pre.predict(g)
    sc.pre.run(sc.loop.x0)
    torch.cuda.synchronize()
    d_idx = sc.pre.traj_idx.cpu().numpy(); d_len = eng.path_len.cpu().numpy().copy(); d_col = sc.pre.col_flag.cpu().numpy(); d_st = sc.pre.status.cpu().numpy()
    for b in range(B):
        full = routes[batch.path_id[b]]
        st, idx, plen, col = LO.loop_pre_tick((x0[b, 0], x0[b, 1], x0[b, 3], x0[b, 2]), int(tidx[b]), None if prev[b] < 0 else int(prev[b]),
                                              full, obst, S.DL)
        if st != 0:
            ok = d_st[b] == st and d_len[b] == plen_before[b] and d_idx[b] == tidx[b]
        else:
            ok = d_st[b] == 0 and d_idx[b] == idx and d_len[b] == plen and bool(d_col[b]) == (col is not None)
        n_glue += not ok
        n_col += col is not None
    n_cut += int((d_len < batch.path_len).sum())
    tind = eng.target_ind.cpu().numpy().copy(); oa = eng.oa.cpu().numpy().copy(); od = eng.od.cpu().numpy().copy()
    eng.solve(sc.loop.x0)
    torch.cuda.synchronize()
    ref = O.mpc_step_batch(p, x0, batch.path_id, d_len, batch.speed, cx, cy, cyaw, off, tind, oa, od, n_threads=16)
    stg = eng.status.cpu().numpy()
    okm = (stg == 0) & (ref["status"] == 0)
    n_st += int((stg != ref["status"]).sum())
    use = ref["status"] != 2
    n_ti += int((eng.target_ind.cpu().numpy() != ref["target_ind"])[use].sum())
    n_mk += int((eng.active_mask.cpu().numpy().view(np.uint32) != ref["active_mask"]).any(axis=1).sum())
    if okm.any():
        worst = max(worst, float(np.abs(eng.oa.cpu().numpy() - ref["oa"])[okm].max()), float(np.abs(eng.od.cpu().numpy() - ref["od"])[okm].max()))
    lp = sc.loop
    pkg._cabi.check(eng.lib.jsim_loop_advance(
        eng._ctx, eng.B, lp.x0.data_ptr(), eng.oa.data_ptr(), eng.od.data_ptr(), eng.status.data_ptr(), eng.di_ai.data_ptr(),
        eng.target_ind.data_ptr(), eng.path_id.data_ptr(), eng.path_len.data_ptr(), lp.x0_spawn.data_ptr(), lp.target_spawn.data_ptr(),
        lp.age.data_ptr(), lp.max_age, None, lp.tick_counter.data_ptr(), 0, lp.n_respawn.data_ptr(), eng._stream()), eng._ctx, "jsim_loop_advance")
    resp = lp.age == 0
    sc.pre.traj_idx.masked_fill_(resp, 0); sc.pre.prev_len.masked_fill_(resp, -1)
    sc.obst.get(step=True)
print(f"T={T}: {B} egos x {K} ticks with {len(specs)} obstacle vehicles: {n_col} collision findings, {n_cut} truncated-path steps, "
      f"{int(sc.loop.n_respawn.item())} respawns; glue diffs {n_glue}, status diffs {n_st}, target_ind diffs {n_ti}, active-set diffs {n_mk}, max|du| {worst:.2e}")
bad = n_glue + n_st + n_ti + n_mk + (worst > 1e-7)
print("SCENARIO SOAK", "CLEAN" if bad == 0 else f"FOUND {bad} DIFFERENCES")
sys.exit(0 if bad == 0 else 1)
