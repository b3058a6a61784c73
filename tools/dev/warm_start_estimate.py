#!/usr/bin/env python3
"""Dev: would a working-set warm start pay for the egos that set the launch time?  CPU oracle closed loop (synthetic routes); per
solve: the final active set A_k, and the prediction P_k = A_{k-1} shifted by one time step (row (family, t) -> (family, t - 1)).
For the heavy solves (>= 30 iterations) reports |A|, right = |P & A|, wrong = |P - A|, missed = |A - P|.
usage: warm_start_estimate.py [T=30] [B=1024] [ticks=60]"""
import importlib, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
pkg = importlib.import_module("av-simulation-at-intersections_amd")
import oracle_py as O
S = pkg.synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = int(sys.argv[3]) if len(sys.argv) > 3 else 60
routes = S.make_route_table(multi_lane=(T == 40))
for r in routes: S.smooth_yaw_inplace(r[:, 2])
batch = S.make_ego_batch(routes, B, T, seed=1, truncate=False)
p = O.make_params(T=T)
cx, cy, cyaw, off = S.pack_paths(routes)
st = O.loop_state_from_batch(batch, T)
m = 8 * T
def shift(bits):
    out = np.zeros_like(bits)
    # D: pairs t = 0..T-2 -> ids 2t, 2t+1 ; VU t=0..T -> 2T-2+t ; VL -> 3T-1+t ; AU -> 4T+t ; AL -> 5T+t ; S pairs -> 6T+2t(+1)
    out[:, 0:2 * T - 4] = bits[:, 2:2 * T - 2]
    out[:, 2 * T - 2:3 * T - 2] = bits[:, 2 * T - 1:3 * T - 1]
    out[:, 3 * T - 1:4 * T - 1] = bits[:, 3 * T:4 * T]
    out[:, 4 * T:5 * T - 1] = bits[:, 4 * T + 1:5 * T]
    out[:, 5 * T:6 * T - 1] = bits[:, 5 * T + 1:6 * T]
    out[:, 6 * T:8 * T - 2] = bits[:, 6 * T + 2:8 * T]
    out[:, 2 * T - 2] = False; out[:, 3 * T - 1] = False     # the t = 0 speed rows are constants
    return out
prev = None
rows = []
for k in range(K):
    ref = O.mpc_step_batch(p, st["x0"], st["path_id"], st["path_len"], st["speed"], cx, cy, cyaw, off, st["target_ind"], st["oa"], st["od"], n_threads=8)
    w = ref["active_mask"]
    bits = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(B, -1)[:, :m].astype(bool)
    age0 = st["age"].copy()
    O.closed_loop(p, st, cx, cy, cyaw, off, 1, max_age=400, n_threads=8, record=False)
    if prev is not None:
        P = shift(prev)
        fresh = age0 == 0          # respawned last tick: no prediction
        for b in range(B):
            if ref["status"][b] != 0 or fresh[b]: continue
            rows.append((ref["n_iter"][b], bits[b].sum(), (P[b] & bits[b]).sum(), (P[b] & ~bits[b]).sum(), (bits[b] & ~P[b]).sum()))
    prev = bits
    prev[st["age"] == 0] = False
a = np.array(rows, dtype=float)
print(f"T={T}, {B} egos x {K} ticks: {len(a)} solves with a prediction; iterations mean {a[:,0].mean():.1f}")
for lo, hi in ((0, 10), (10, 20), (20, 30), (30, 50), (50, 1000)):
    s = a[(a[:, 0] >= lo) & (a[:, 0] < hi)]
    if len(s):
        print(f"  solves with {lo:3d} <= n_iter < {hi:4d}: {len(s):6d}; |A| {s[:,1].mean():6.1f}  predicted right {s[:,2].mean():6.1f}  wrong {s[:,3].mean():5.1f}  missed {s[:,4].mean():5.1f}")
