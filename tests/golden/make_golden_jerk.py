#!/usr/bin/env python3
"""Golden vectors for the acceleration-state variant `lib.mpc_jerk`, made by the REFERENCE's own
functions (build container only): its module constants, `_get_linear_model_matrix` (the 5x5 / 5x2 / 5 model),
`_calc_ref_trajectory` (five-row xref) and `_predict_motion` (five-row xbar).  `import cvxpy` at the top of that module is
satisfied by an empty placeholder module (cvxpy is not installed; nothing of it is called); `_linear_mpc_control`, the only
function that touches cvxpy, is NOT run: the variant's QP has no golden vectors (parity unpinned against ECOS, like
lib.mpc) and is pinned by tests/qp_sparse_numpy.py instead.  No pickle is loaded; routes are this repo's synthetic ones."""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_MAIN = "/root/reference/main"


def main():
    if not os.path.isdir(REF_MAIN):
        raise SystemExit("reference not present")
    sys.modules.setdefault("cvxpy", types.ModuleType("cvxpy"))
    sys.path.insert(0, REF_MAIN)
    import matplotlib
    matplotlib.use("Agg")
    import lib.mpc_jerk as ref
    from lib.car_dimensions import BicycleModelDimensions
    from lib.simulation import State
    spec = importlib.util.spec_from_file_location("jsim_synth", os.path.join(REPO, "av-simulation-at-intersections_amd", "synth.py"))
    S = importlib.util.module_from_spec(spec); sys.modules["jsim_synth"] = S; spec.loader.exec_module(S)
    routes = S.make_route_table()
    for r in routes:
        S.smooth_yaw_inplace(r[:, 2])
    car = BicycleModelDimensions()
    consts = dict(NX=ref.NX, NU=ref.NU, T=ref.T, R=np.diag(ref.R), Rd=np.diag(ref.Rd), Q_v_yaw=np.diag(ref.Q_v_yaw),
                  Qf_scaled=np.diag(ref.Qf), GOAL_DIS=ref.GOAL_DIS, STOP_SPEED=ref.STOP_SPEED, MAX_ITER=ref.MAX_ITER,
                  MAX_DSTEER=ref.MAX_DSTEER, MAX_ACCEL=ref.MAX_ACCEL, MAX_DECEL=ref.MAX_DECEL,
                  jerk_penalty_weight=ref.jerk_penalty_weight)
    rng = np.random.default_rng(11)
    # linear model
    K = 64
    v = rng.uniform(-5, 8.4, K); phi = rng.uniform(-7, 7, K); delta = np.where(rng.random(K) < 0.5, 0.0, rng.uniform(-0.7, 0.7, K))
    A = np.zeros((K, 5, 5)); B = np.zeros((K, 5, 2)); C = np.zeros((K, 5))
    for k in range(K):
        A[k], B[k], C[k] = ref._get_linear_model_matrix(v[k], phi[k], delta[k], S.DT, car.distance_back_to_front_wheel)
    # reference window + rollout
    N = 32
    batch = S.make_ego_batch(routes, N, ref.T, seed=78, truncate=True, near_end_frac=0.3)
    xref_all = np.zeros((N, 5, ref.T + 1)); xbar_all = np.zeros((N, 5, ref.T + 1))
    tind = np.zeros(N, dtype=np.int64); rend = np.zeros((N, ref.T + 1), dtype=bool)
    for b in range(N):
        r = routes[batch.path_id[b]][: batch.path_len[b]]
        x, y, vv, yaw = batch.x0[b]
        xref, ti, dref, re = ref._calc_ref_trajectory(State(x=x, y=y, yaw=yaw, v=vv), r[:, 0], r[:, 1], r[:, 2], S.DL, S.DT,
                                                      int(batch.target_ind[b]), None)
        xbar = ref._predict_motion([x, y, vv, yaw], batch.oa[b], batch.od[b], xref, car_dimensions=car, dt=S.DT)
        xref_all[b], tind[b], rend[b], xbar_all[b] = xref, ti, re, xbar
    np.savez(os.path.join(HERE, "variant_jerk.npz"), lm_v=v, lm_phi=phi, lm_delta=delta, lm_A=A, lm_B=B, lm_C=C,
             x0=batch.x0, path_id=batch.path_id, path_len=batch.path_len, target_ind_in=batch.target_ind, oa=batch.oa, od=batch.od,
             xref=xref_all, xbar=xbar_all, target_ind_out=tind, reaches_end=rend,
             **{"c_" + k: np.asarray(v_) for k, v_ in consts.items()})
    print("variant_jerk.npz:", K, "linear models,", N, "windows/rollouts; fifth rows all zero:",
          bool((xref_all[:, 4] == 0).all() and (xbar_all[:, 4] == 0).all()))


if __name__ == "__main__":
    main()
