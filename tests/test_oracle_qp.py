"""S4 (QP build + exact solve) of the oracle.  The reference's solver (cvxpy->ECOS) is absent and the
reference has no fixtures for it => against ECOS this stage is PARITY UNPINNED; what pins the oracle
instead (CPU only):
  * its condensed (H, g, G, h) equal an independent numpy restatement of the reference's SPARSE
    cvxpy problem (tests/qp_sparse_numpy.py) after generic elimination of the states;
  * KKT residuals of the strictly convex QP (unique optimum) <= 1e-8, and an independent
    equality-constrained re-solve on the reported active set reproduces u*;
  * scipy (SLSQP / trust-constr) agrees on a sample;
  * the survey's probe: standing start => u0 = (2.0, 0.0), active set = the first 12 accel-upper rows.
"""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import qp_sparse_numpy as QS

TS = (13, 20, 30, 40)


def _ego_cases(pkg, routes, T, n=24, seed=7):
    return pkg.synth.make_ego_batch(routes, n, T, seed=seed + T, truncate=True, near_end_frac=0.3)


def _stages(oracle, p, routes, batch, b):
    r = routes[int(batch.path_id[b])][: int(batch.path_len[b])]
    x, y, v, yaw = batch.x0[b]
    st_, xref, idx, rend, tind = oracle.calc_ref_trajectory(p, x, y, v, r[:, 0], r[:, 1], r[:, 2],
                                                            int(batch.target_ind[b]))
    assert st_ == 0
    xbar = oracle.predict_motion(p, batch.x0[b], batch.oa[b], batch.od[b])
    return r, xref, xbar, rend


@pytest.mark.parametrize("T", TS)
def test_condensed_qp_equals_sparse_reference_form(oracle, pkg, routes, T):
    p = oracle.make_params(T=T)
    batch = _ego_cases(pkg, routes, T, n=6)
    for b in range(6):
        r, xref, xbar, rend = _stages(oracle, p, routes, batch, b)
        st_, H, g, G, h, skip, fresp, Sens = oracle.build_qp(p, xref, xbar, batch.x0[b], rend, batch.speed[b])
        P, q, c0, Aeq, beq, Gin, hin = QS.build_sparse(oracle.STOCK_CONFIG, T, p.dt, p.L, xref, xbar,
                                                       batch.x0[b], rend, batch.speed[b])
        H2, g2, G2, h2, Phi, phi = QS.condense(P, q, Aeq, beq, Gin, hin, T)
        sc = np.abs(H2).max()
        assert np.abs(H - H2).max() <= 1e-10 * sc
        assert np.abs(g - g2).max() <= 1e-9 * max(1.0, np.abs(g2).max())
        assert np.abs(G - G2).max() <= 1e-12
        assert np.abs(h - h2).max() <= 1e-11
        # free response / sensitivities are the eliminated states
        np.testing.assert_allclose(fresp.T.reshape(-1), phi, rtol=0, atol=1e-10)
        np.testing.assert_allclose(Sens, Phi, rtol=0, atol=1e-11)
        assert H.shape == (2 * T, 2 * T) and G.shape == (8 * T, 2 * T)
        assert np.linalg.eigvalsh(H).min() >= 2 * 0.01 - 1e-9   # strictly convex: lambda_min >= 2*min(R)


def _verify_solution(H, g, G, h, skip, u, lam, active):
    stat, prim, dual, comp = QS.kkt_check(H, g, G, h, skip, u, lam)
    sc = max(1.0, np.abs(g).max())
    assert stat <= 1e-8 * sc and prim <= 1e-9 and dual == 0.0 and comp <= 1e-8 * sc
    # independent equality-constrained solve on the reported active set
    A = np.array(sorted(active), dtype=int)
    n = len(g)
    K = np.block([[H, G[A].T], [G[A], np.zeros((len(A), len(A)))]])
    sol = np.linalg.lstsq(K, np.concatenate([-g, h[A]]), rcond=None)[0]
    # (the indefinite KKT matrix is the ill-conditioned side here: cond(H) reaches 2.6e8 at T=40 when
    # truncated paths put Qf*T on many steps; against a 50-digit solve the oracle is within 2e-9)
    np.testing.assert_allclose(sol[:n], u, rtol=0, atol=2e-6)


@pytest.mark.parametrize("T", TS)
def test_solve_kkt_random_egos(oracle, pkg, routes, T):
    p = oracle.make_params(T=T)
    n_cases = 24 if T <= 20 else 10
    batch = _ego_cases(pkg, routes, T, n=n_cases)
    kinds = set()
    for b in range(n_cases):
        r, xref, xbar, rend = _stages(oracle, p, routes, batch, b)
        st_, H, g, G, h, skip, fresp, Sens = oracle.build_qp(p, xref, xbar, batch.x0[b], rend, batch.speed[b])
        assert st_ == 0
        st2, u, lam, it = oracle.solve_qp(H, g, G, h, skip)
        assert st2 == 0 and it < 20 * T
        thr = 1e-9 * max(1.0, np.abs(g).max())
        active = [i for i in range(8 * T) if lam[i] > thr]
        _verify_solution(H, g, G, h, skip, u, lam, active)
        for i in active:
            kinds.add("D" if i < 2 * T - 2 else "VU" if i < 3 * T - 1 else "VL" if i < 4 * T else
                      "AU" if i < 5 * T else "AL" if i < 6 * T else "S")
        # full step API agrees with the staged calls
        res = oracle.mpc_step(p, (batch.x0[b][0], batch.x0[b][1], batch.x0[b][3], batch.x0[b][2]),
                              r[:, 0], r[:, 1], r[:, 2], int(batch.target_ind[b]), batch.speed[b],
                              oa=batch.oa[b], od=batch.od[b], want_qp=True)
        assert res["status"] == 0 and res["active"] == active
        np.testing.assert_array_equal(res["oa"], u[0::2])
        np.testing.assert_array_equal(res["od"], u[1::2])
        # predicted states obey the linearised dynamics (what cvxpy's x variable would hold)
        z = fresp + (Sens @ u).reshape(T + 1, 4).T
        np.testing.assert_allclose(np.stack([res["ox"], res["oy"], res["ov"], res["oyaw"]]), z, atol=1e-12)
    assert {"AU"} <= kinds  # accelerate-to-cruise is always present in this batch


def _crafted(oracle, routes, T, state, target=0, rid=1, speed=30 / 3.6, path_len=None, oa=None, od=None):
    p = oracle.make_params(T=T)
    r = routes[rid][: path_len or len(routes[rid])]
    return p, r, oracle.mpc_step(p, state, r[:, 0], r[:, 1], r[:, 2], target, speed, oa=oa, od=od, want_qp=True)


def test_standing_start_matches_survey_probe(oracle, routes):
    """scipy trust-constr on the condensed QP gave u0 = (2.0, 0.0) and the active set =
    the first 12 accel-upper rows for a standing-start ego (T = 13)."""
    r = routes[1]
    p, r, res = _crafted(oracle, routes, 13, (r[0, 0], r[0, 1], r[0, 2], 0.0))
    assert res["status"] == 0
    assert abs(res["oa"][0] - 2.0) < 1e-12 and abs(res["od"][0]) < 1e-10
    assert res["active"] == list(range(4 * 13, 4 * 13 + 12))


def test_each_constraint_family_can_be_active(oracle, routes):
    T = 20
    fam = lambda i: ("D" if i < 2 * T - 2 else "VU" if i < 3 * T - 1 else "VL" if i < 4 * T else
                     "AU" if i < 5 * T else "AL" if i < 6 * T else "S")
    seen = set()
    r = routes[0]  # left turn
    # big lateral offset + heading error inside the turn -> steer and steer-rate rows
    k = 330
    _, _, res = _crafted(oracle, routes, T, (r[k, 0] + 2.5, r[k, 1] - 2.5, r[k, 2] - 1.2, 6.0), target=k - 5, rid=0)
    assert res["status"] == 0
    seen |= {fam(i) for i in res["active"]}
    # fast ego just before a truncated path end -> hard braking (accel lower) rows
    _, _, res = _crafted(oracle, routes, T, (r[100, 0], r[100, 1], r[100, 2], 8.3), target=95, rid=0, path_len=130)
    assert res["status"] == 0
    seen |= {fam(i) for i in res["active"]}
    # at the speed cap with a warm start that keeps accelerating -> speed-upper rows
    _, _, res = _crafted(oracle, routes, T, (r[50, 0], r[50, 1], r[50, 2], 30 / 3.6), target=48, rid=0,
                         speed=30 / 3.6, oa=np.full(T, 2.0), od=np.zeros(T))
    assert res["status"] == 0
    seen |= {fam(i) for i in res["active"]}
    # reversing ego at the lower speed bound
    _, _, res = _crafted(oracle, routes, T, (r[300, 0], r[300, 1], r[300, 2] + np.pi, -4.9), target=280, rid=0)
    assert res["status"] == 0
    seen |= {fam(i) for i in res["active"]}
    assert {"D", "S", "AL", "AU"} <= seen, seen


def test_infeasible_when_v0_above_speed(oracle, routes):
    r = routes[1]
    _, _, res = _crafted(oracle, routes, 13, (r[10, 0], r[10, 1], r[10, 2], 8.0), target=5, speed=5.0)
    assert res["status"] == 1          # reference: x[2,0] == v0 and x[2,:] <= speed are inconsistent
    assert res["target_ind"] >= 5      # target_ind is still advanced (mpc.py:293)
    _, _, res = _crafted(oracle, routes, 13, (r[10, 0], r[10, 1], r[10, 2], -5.5), target=5)
    assert res["status"] == 1


@pytest.mark.parametrize("T,which", [(13, 0), (13, 3), (13, 11), (20, 5)])
def test_scipy_cross_check(oracle, pkg, routes, T, which):
    from scipy.optimize import minimize
    p = oracle.make_params(T=T)
    batch = _ego_cases(pkg, routes, T, n=12)
    r, xref, xbar, rend = _stages(oracle, p, routes, batch, which)
    st_, H, g, G, h, skip, _, _ = oracle.build_qp(p, xref, xbar, batch.x0[which], rend, batch.speed[which])
    st2, u, lam, it = oracle.solve_qp(H, g, G, h, skip)
    keep = ~skip.astype(bool)
    f = lambda x: 0.5 * x @ H @ x + g @ x
    res = minimize(f, np.zeros(2 * T), jac=lambda x: H @ x + g, method="SLSQP",
                   constraints=[{"type": "ineq", "fun": lambda x: h[keep] - G[keep] @ x,
                                 "jac": lambda x: -G[keep]}],
                   options={"maxiter": 500, "ftol": 1e-14})
    assert res.success
    assert f(u) <= f(res.x) + 1e-8 * max(1.0, abs(f(res.x)))   # the oracle's optimum is at least as good
    np.testing.assert_allclose(res.x, u, rtol=0, atol=2e-5)


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(2, 14), st.integers(1, 40))
def test_generic_dense_qp_solver_kkt(seed, n, m):
    """Property test of the active-set solver alone on random strictly convex QPs (incl. duplicated and
    linearly dependent rows)."""
    import oracle_py as oracle
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n))
    H = A @ A.T + 0.05 * np.eye(n)
    g = rng.normal(size=n) * 3
    G = rng.normal(size=(m, n))
    if m >= 4:
        G[1] = G[0]                # duplicate row
        G[3] = G[0] + G[2]         # dependent row
    u_feas = rng.normal(size=n)
    h = G @ u_feas + rng.uniform(0.0, 1.0, size=m)
    st_, u, lam, it = oracle.solve_qp(H, g, G, h)
    assert st_ == 0
    stat, prim, dual, comp = QS.kkt_check(H, g, G, h, np.zeros(m, dtype=np.uint8), u, lam)
    sc = max(1.0, np.abs(g).max(), np.abs(H).max())
    assert stat <= 1e-8 * sc and prim <= 1e-8 and dual == 0.0 and comp <= 1e-7 * sc


def test_generic_solver_detects_infeasible():
    import oracle_py as oracle
    H = np.eye(2); g = np.zeros(2)
    G = np.array([[1.0, 0.0], [-1.0, 0.0]]); h = np.array([-1.0, -1.0])   # x <= -1 and x >= 1
    st_, u, lam, it = oracle.solve_qp(H, g, G, h)
    assert st_ == 1


# ------------------------------------------------------------------------------------------------
# the acceleration-state variant (main/lib/mpc_jerk.py): 2T + 1 decision variables
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T", (13, 20))
def test_jerk_variant_condensed_qp_equals_its_sparse_form_and_solves(oracle, pkg, routes, T):
    """lib/mpc_jerk.py is dead code in the reference (only commented-out imports), so nothing but its source pins it:
    the oracle's condensed (H, g, G, h) with the free acc_0 as variable 2T against an independent numpy restatement of
    the variant's sparse cvxpy problem, then the exact solve by KKT residuals and an active-set re-solve."""
    cfg = dict(QS.JERK_CONFIG)
    p = oracle.make_params(T=T, config=cfg)
    assert p.nx == 5 and oracle.nvar(p) == 2 * T + 1
    batch = _ego_cases(pkg, routes, T, n=10, seed=31)
    vmax = 30 / 3.6                                     # Simulation.MAX_SPEED (mpc_jerk.py:194), not the ego's speed
    n_act = 0
    for b in range(10):
        r, xref, xbar, rend = _stages(oracle, p, routes, batch, b)
        st_, H, g, G, h, skip, fresp, Sens = oracle.build_qp(p, xref, xbar, batch.x0[b], rend, vmax)
        assert st_ == 0
        P, q, c0, Aeq, beq, Gin, hin = QS.build_sparse_jerk(cfg, T, p.dt, p.L, xref, xbar, batch.x0[b], rend, vmax)
        H2, g2, G2, h2, Phi, phi = QS.condense_jerk(P, q, Aeq, beq, Gin, hin, T)
        sc = np.abs(H2).max()
        assert H.shape == (2 * T + 1, 2 * T + 1)
        assert np.abs(H - H2).max() <= 1e-10 * sc
        assert np.abs(g - g2).max() <= 1e-9 * max(1.0, np.abs(g2).max())
        assert np.abs(G - G2).max() <= 1e-12
        assert np.abs(h - h2).max() <= 1e-11
        np.testing.assert_allclose(Sens, Phi, rtol=0, atol=1e-11)
        np.testing.assert_allclose(fresp.T.reshape(-1), phi, rtol=0, atol=1e-10)
        assert np.linalg.eigvalsh(H).min() > 0.0        # acc_0 has no cost of its own: curvature comes through v -> x, y
        st2, u, lam, it = oracle.solve_qp(H, g, G, h, skip)
        assert st2 == 0
        thr = 1e-9 * max(1.0, np.abs(g).max())
        active = [i for i in range(8 * T) if lam[i] > thr]
        n_act += len(active)
        _verify_solution(H, g, G, h, skip, u, lam, active)
        res = oracle.mpc_step(p, (batch.x0[b][0], batch.x0[b][1], batch.x0[b][3], batch.x0[b][2]),
                              r[:, 0], r[:, 1], r[:, 2], int(batch.target_ind[b]), vmax,
                              oa=batch.oa[b], od=batch.od[b], want_qp=True)
        assert res["status"] == 0 and res["active"] == active
        np.testing.assert_array_equal(res["oa"], u[0:2 * T:2])     # oa = u[0, :], the input of the acceleration state
        np.testing.assert_array_equal(res["od"], u[1:2 * T:2])
        z = fresp + (Sens @ u).reshape(T + 1, 5).T
        np.testing.assert_allclose(np.stack([res["ox"], res["oy"], res["ov"], res["oyaw"]]), z[:4], atol=1e-12)
        # v_{t+1} = v_t + dt * (acc_t + u0_t),  acc_{t+1} = acc_t + dt * u0_t  (mpc_jerk.py:67-78)
        acc = np.concatenate([[u[2 * T]], u[2 * T] + p.dt * np.cumsum(u[0:2 * T:2])])
        np.testing.assert_allclose(z[4], acc, atol=1e-12)
        np.testing.assert_allclose(np.diff(z[2]), p.dt * (acc[:-1] + u[0:2 * T:2]), atol=1e-12)
    assert n_act > 0
