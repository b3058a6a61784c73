#!/usr/bin/env python3
"""Diagnostic: where one mpc_step_kernel launch spends its cycles.  Builds csrc/jsim_mpc.hip with
-DJSIM_STAMPS into a SEPARATE library (libjsim_mpc_stamps.so; the shipped library carries no stamps),
runs the bench workload for a few ticks and prints per-phase cycle shares (median over egos).
Never quote this build's run time -- read its shares."""
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
lib = os.environ.get("JSIM_STAMPS_LIB") or os.path.join(REPO, "av-simulation-at-intersections_amd", "libjsim_mpc_stamps.so")
if not os.environ.get("JSIM_STAMPS_LIB"):   # (JSIM_STAMPS_LIB: a -DJSIM_STAMPS library built beforehand)
    subprocess.check_call([pkg.build._hipcc()] + pkg.build.HIPCC_FLAGS + ["-DJSIM_STAMPS", "-I", pkg.build.INC, pkg.build.SRC, "-o", lib])
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
names = ["S1 nearest", "S1 idx/xref", "S2 rollout", "S3 coef+scans", "S4a H mfma", "g + R/Rd", "cholesky", "J=L^-T",
         "u0=-JJ'g", "active-set loop", "S5 outputs"]
acc = np.zeros((0, 11))
gacc = np.zeros((0, 8)); itacc = np.zeros(0)
for tick in range(30):
    loop.tick()
    torch.cuda.synchronize()
    call = clk.cpu().numpy()
    c = call[:B * 16].reshape(B, 16)
    ok = (eng.status.cpu().numpy() == 0)
    acc = np.concatenate([acc, np.diff(c[ok][:, :12], axis=1)])
    gacc = np.concatenate([gacc, call[B * 16:].reshape(B, 8)[ok]])
    itacc = np.concatenate([itacc, eng.n_iter.cpu().numpy()[ok]])
it = eng.n_iter.cpu().numpy()
tot = acc.sum(axis=1)
print(f"T={T} B={B}: total cycles/ego median {np.median(tot):.0f} (p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f}); "
      f"last-tick mean n_iter {it.mean():.1f}")
for k, nme in enumerate(names):
    print(f"  {nme:18s} median {np.median(acc[:, k]):10.0f} cyc  {100 * acc[:, k].sum() / tot.sum():5.1f} %")

gn = ["step1 argmax", "publish row", "reduce d", "dots z,y", "backsubst", "ratio+step", "add (Householder)", "drop (Givens)"]
tot_it = itacc.sum()
print(f"active-set loop: {tot_it:.0f} iterations over {len(itacc)} solves (mean {itacc.mean():.1f}); cycles per iteration by section:")
for k, nme in enumerate(gn):
    print(f"  {nme:20s} {gacc[:, k].sum() / max(tot_it, 1):8.0f} cyc/iter   {100 * gacc[:, k].sum() / gacc.sum():5.1f} %")

# per-solve regression: cycles of the active-set phase and of the whole step against the iteration count
A = np.vstack([np.ones_like(itacc), itacc]).T
for nme, y in (("active-set loop", acc[:, 9]), ("whole step", tot), ("u0 phase", acc[:, 8])):
    coef, *_ = np.linalg.lstsq(A, y, rcond=None)
    print(f"  fit {nme:16s} = {coef[0]:9.0f} + {coef[1]:7.0f} * n_iter cycles")
# the same sections for the solves with the most iterations (the egos that set the launch time)
big = itacc >= np.percentile(itacc, 99.5)
if big.sum():
    print(f"solves with n_iter >= {np.percentile(itacc, 99.5):.0f} ({int(big.sum())} solves, mean n_iter {itacc[big].mean():.1f}, mean step {tot[big].mean():.0f} cycles):")
    for k, nme in enumerate(gn):
        print(f"  {nme:20s} {gacc[big, k].sum() / itacc[big].sum():8.0f} cyc/iter   {100 * gacc[big, k].sum() / gacc[big].sum():5.1f} %")
