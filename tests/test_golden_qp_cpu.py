"""CPU: the oracle reproduces its own committed G3 fixtures (tests/golden/qp_T*.npz) -- guards the fixtures against drift of
oracle/mpc_oracle.c -- and the fixtures hold what their header promises (>= 50 cases per horizon, every constraint family
active, the infeasible and the coincident-rows cases, the 50-digit KKT certificate, the scipy trust-constr cross-check)."""
import numpy as np
import pytest

from conftest import load_golden


@pytest.mark.parametrize("T", (13, 20, 30, 40))
def test_oracle_reproduces_golden_qp(oracle, pkg, routes, T):
    g = load_golden(f"qp_T{T}.npz")
    n = len(g["x0"])
    assert n >= 50 and (g["status"] == 1).sum() >= 2
    solved = g["status"] == 0
    assert float(g["du_mp"][solved].max()) <= 1e-7                    # 50-digit KKT certificate of every solved case
    assert float(g["du_scipy"][g["scipy_method"] == 0].max()) <= 1e-4  # trust-constr where it converged ...
    assert float(g["du_scipy"][solved].max()) <= 5e-3                  # ... and never far where it stalled (recorded, higher objective)
    p = oracle.make_params(T=T)
    cx, cy, cyaw, off = pkg.synth.pack_paths(routes)
    ref = oracle.mpc_step_batch(p, np.ascontiguousarray(g["x0"]), g["path_id"], g["path_len"], g["speed"], cx, cy, cyaw, off,
                                g["target_ind_in"], g["oa_in"], g["od_in"], n_threads=4)
    assert np.array_equal(ref["status"], g["status"]) and np.array_equal(ref["target_ind"], g["target_ind_out"])
    ok = g["status"] == 0
    assert np.abs(ref["oa"] - g["oa"])[ok].max() <= 1e-12 and np.abs(ref["od"] - g["od"])[ok].max() <= 1e-12
    assert np.array_equal(ref["active_mask"][ok], g["active_mask"][ok]) and np.array_equal(ref["n_iter"][ok], g["n_iter"][ok])
    bits = ((g["active_mask"][:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, -1)[:, :8 * T].astype(bool)
    assert bits[:, :2 * T - 2].any() and bits[:, 2 * T - 2:3 * T - 1].any() and bits[:, 3 * T - 1:4 * T].any()
    assert bits[:, 4 * T:5 * T].any() and bits[:, 5 * T:6 * T].any() and bits[:, 6 * T:].any()
    # the coincident-rows case (v0 = speed - MAX_ACCEL*dt): of `v1 <= speed` (row 2T-1) and `a0 <= MAX_ACCEL` (row 4T), which are
    # the same half-space, exactly the lower row id is reported active
    c = int(g["crafted_first"]) + 11
    assert abs(g["x0"][c][2] - (30 / 3.6 - 0.4)) < 1e-12
    assert bits[c, 2 * T - 1] and not bits[c, 4 * T]
