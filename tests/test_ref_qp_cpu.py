"""Stage S4 against the REFERENCE's own assembly.

tests/golden/ref_qp_T*.npz hold the problem `_linear_mpc_control` (main/lib/mpc.py:141-211) emitted, unmodified, under the
recording cvxpy stand-in, and what the reference's S5 lines (:199-211, :298-303) made of its optimum.  Checked here, on CPU:
  * the oracle's condensed (H, g, G, h) == the emitted sparse problem with x eliminated GENERICALLY through A z = b
    (tests/qp_sparse_numpy.condense), <= 1e-12 relative (g: 1e-11), row for row: canonical row order == emitted order;
  * the emitted constraint list has the shape include/jsim_mpc.h documents (which family sits where);
  * the oracle's whole step == the reference's whole step: status / target_ind / xref bit for bit, u* and predicted states
    <= 1e-8, active sets identical on every case whose tight rows are independent, `active subset of tight` on the others;
  * the stand-in itself on problems with known answers.
What stays unpinned: ECOS's stopping tolerance around this (unique) optimum -- see the stand-in's header."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, load_golden
import qp_sparse_numpy as QS
import refqp_tools as RT

TS = (13, 20, 30, 40)


@pytest.mark.parametrize("T", TS)
def test_oracle_assembly_equals_reference_emitted_problem(oracle, pkg, routes, T):
    g = load_golden(f"ref_qp_T{T}.npz")
    n = len(g["x0"])
    assert n >= 150 and (g["status"] == 1).sum() == 2
    p = oracle.make_params(T=T)
    worst = dict(dH=0.0, dg=0.0, dG=0.0, dh=0.0)
    for i in range(n):
        P, q, c0, A, b, G, h = RT.emitted_problem(g, i)
        assert A.shape == (4 * (T + 1), 6 * T + 4) and G.shape == (8 * T, 6 * T + 4)
        Hr, gr, Gr, hr, Phi, phi = QS.condense(P, q, A, b, G, h, T)
        x0 = g["x0"][i]
        rt = routes[int(g["path_id"][i])][:int(g["path_len"][i])]
        st, xref, idx, rend, tind = oracle.calc_ref_trajectory(p, x0[0], x0[1], x0[2], rt[:, 0], rt[:, 1], rt[:, 2],
                                                              int(g["target_ind_in"][i]))
        assert st == 0 and tind == g["target_ind_out"][i] and np.array_equal(xref, g["xref"][i])
        xbar = oracle.predict_motion(p, x0, g["oa_in"][i], g["od_in"][i])
        st3, H, gv, Go, ho, skip, fresp, Sens = oracle.build_qp(p, xref, xbar, x0, rend, float(g["speed"][i]))
        keep = ~skip.astype(bool)
        # the two t = 0 speed rows are constants of the emitted problem too (x[:, 0] == x0): zero rows after elimination
        assert list(np.flatnonzero(~keep)) == [2 * T - 2, 3 * T - 1] and np.abs(Gr[~keep]).max() <= 1e-13
        worst["dH"] = max(worst["dH"], np.abs(H - Hr).max() / np.abs(Hr).max())
        worst["dg"] = max(worst["dg"], np.abs(gv - gr).max() / max(1.0, np.abs(gr).max()))
        worst["dG"] = max(worst["dG"], np.abs(Go[keep] - Gr[keep]).max())
        worst["dh"] = max(worst["dh"], np.abs(ho[keep] - hr[keep]).max())
        # infeasible <=> a constant row is violated beyond the feasibility tolerance (the reference's "Cannot solve mpc" path)
        assert (g["status"][i] == 1) == bool((hr[~keep] < -1e-8).any())
    assert worst["dH"] <= 1e-12 and worst["dG"] <= 1e-12 and worst["dh"] <= 1e-12 and worst["dg"] <= 1e-11, worst


@pytest.mark.parametrize("T", TS)
def test_emitted_constraint_list_is_the_canonical_order(T):
    """Which constraint of the reference's list each row came from, and what the row looks like: D rows interleaved with the
    dynamics inside the t loop (mpc.py:187), then x0, VU, VL, AU, AL, S (mpc.py:189-194)."""
    g = load_golden(f"ref_qp_T{T}.npz")
    src = g["in_src"]
    assert len(src) == 8 * T and np.all(np.diff(src) >= 0)
    fam = RT.canonical_row_families(T)
    counts = [int((src == s).sum()) for s in np.unique(src)]
    assert counts == [2] * (T - 1) + [T + 1, T + 1, T, T, 2 * T]           # T-1 abs() pairs, then the five vector constraints
    P, q, c0, A, b, G, h = RT.emitted_problem(g, 0)
    nx = 4 * (T + 1)
    xi = lambda t, r: 4 * t + r
    ui = lambda t, c: nx + 2 * t + c
    for name, first, cnt in fam:
        for k in range(cnt):
            row = G[first + k]
            nz = np.flatnonzero(row)
            if name == "D":
                t, sg = k // 2, 1.0 if k % 2 == 0 else -1.0
                assert list(nz) == [ui(t, 1), ui(t + 1, 1)] and row[ui(t + 1, 1)] == sg and row[ui(t, 1)] == -sg
                assert h[first + k] == np.deg2rad(30.0) * 0.2
            elif name in ("VU", "VL"):
                assert list(nz) == [xi(k, 2)] and row[xi(k, 2)] == (1.0 if name == "VU" else -1.0)
                assert h[first + k] == (g["speed"][0] if name == "VU" else 5.0)
            elif name in ("AU", "AL"):
                assert list(nz) == [ui(k, 0)] and row[ui(k, 0)] == (1.0 if name == "AU" else -1.0)
                assert h[first + k] == (2.0 if name == "AU" else 10.0)
            else:
                t, sg = k // 2, 1.0 if k % 2 == 0 else -1.0
                assert list(nz) == [ui(t, 1)] and row[ui(t, 1)] == sg and h[first + k] == np.deg2rad(45.0)
    # equalities: T dynamics blocks (4 rows each) inside the loop, then x[:, 0] == x0
    assert list(np.bincount(g["eq_src"])[np.unique(g["eq_src"])]) == [4] * (T + 1)
    assert np.array_equal(A[-4:, :4], np.eye(4)) and np.array_equal(b[-4:], g["x0"][0])


@pytest.mark.parametrize("T", TS)
def test_oracle_step_equals_reference_step(oracle, pkg, routes, T):
    g = load_golden(f"ref_qp_T{T}.npz")
    n = len(g["x0"])
    p = oracle.make_params(T=T)
    cx, cy, cyaw, off = pkg.synth.pack_paths(routes)
    ref = oracle.mpc_step_batch(p, np.ascontiguousarray(g["x0"]), g["path_id"], g["path_len"], g["speed"], cx, cy, cyaw, off,
                                g["target_ind_in"], g["oa_in"], g["od_in"], n_threads=4)
    assert np.array_equal(ref["status"], g["status"]) and np.array_equal(ref["target_ind"], g["target_ind_out"])
    assert np.array_equal(ref["xref"], g["xref"])
    ok = g["status"] == 0
    for a, b_ in (("oa", "oa"), ("od", "od"), ("ox", "ox"), ("oy", "oy"), ("ov", "ov"), ("oyaw", "oyaw")):
        assert np.abs(ref[a] - g[b_])[ok].max() <= 1e-8, a
    # S5 of the reference: (di, ai) = (odelta[0], oa[0]); a failed solve keeps di and commands MAX_DECEL (mpc.py:298-301)
    assert np.array_equal(g["di"][ok], g["od"][ok][:, 0]) and np.array_equal(g["ai"][ok], g["oa"][ok][:, 0])
    assert np.all(g["ai"][~ok] == -10.0) and np.all(g["di_kept"][~ok])
    bits = RT.active_bits(ref["active_mask"], 8 * T)
    nd = ok & ~g["degenerate"]
    assert nd.sum() >= 140 and np.array_equal(bits[nd], g["active"][nd])
    assert np.all(g["tight"][ok] | ~bits[ok])                               # degenerate cases: active subset of tight
    for name, first, cnt in RT.canonical_row_families(T):
        assert g["active"][:, first:first + cnt].any(), name                 # every family active somewhere
    # the stand-in's own certificate of each optimum
    assert g["kkt_stationarity"].max() <= 1e-9 and g["kkt_primal_eq"].max() <= 1e-10 and g["kkt_primal_in"].max() <= 1e-9
    assert g["kkt_dual_min"][ok].min() >= -1e-11 and g["du_ipm_vs_polish"].max() <= 1e-3   # (the interior-point iterate before the polish: informational)


def test_sweep_summary():
    s = json.load(open(os.path.join(GOLDEN, "ref_qp_sweep.json")))
    for T in TS:
        r = s[str(T)]
        assert r["egos"] == 1000 == r["status_equal"] == r["target_equal"] == r["xref_equal"] == r["active_in_tight"]
        assert r["active_equal_nondegenerate"] == 1000 - r["degenerate"] and r["degenerate"] <= 5
        assert max(r["max_dH"], r["max_dG"], r["max_dh"]) <= 1e-12 and r["max_dg"] <= 1e-11 and r["max_du"] <= 1e-8


# ---- the stand-in on problems with known answers -------------------------------------------------------------------------
@pytest.fixture()
def rec():
    sys.path.insert(0, GOLDEN)
    import cvxpy_recorder
    del cvxpy_recorder.RECORDS[:]
    return cvxpy_recorder


def test_stand_in_known_answers(rec):
    x = rec.Variable((2, 2)); u = rec.Variable((1, 2))
    target = np.array([3.0, -1.0])
    cost = 0.0
    cost += rec.quad_form(target - x[:, 1], np.diag([1.0, 2.0]))
    cost += rec.quad_form(u[:, 0], np.array([[0.5]]))
    cost += rec.quad_form(u[:, 1], np.array([[0.5]]))
    M = np.array([[1.0, 1.0], [0.0, 1.0]])
    cons = [x[:, 1] == M @ x[:, 0] + np.array([1.0, 0.0]) * 0.0 + np.array([0.0, 1.0]),
            x[1, 0] == u[0, 0], x[0, 0] == u[0, 1], rec.abs(u[0, :]) <= 0.75, x[0, :] >= -10.0]
    prob = rec.Problem(rec.Minimize(cost), cons)
    prob.solve(solver=rec.ECOS, verbose=False)
    assert prob.status == rec.OPTIMAL
    r = rec.RECORDS[-1]
    # abs() over a vector: (+e0, -e0, +e1, -e1); >= becomes <= of the negation
    G, h = r["G"], r["h"]
    assert G.shape == (6, 6) and np.array_equal(h, [0.75, 0.75, 0.75, 0.75, 10.0, 10.0])
    assert np.array_equal(G[0], -G[1]) and np.array_equal(G[2], -G[3]) and G[0, 4] == 1.0 and G[2, 5] == 1.0
    assert G[4, 0] == -1.0 and G[5, 2] == -1.0
    # brute force over the box in (u0, u1): x[:,0] = (u1, u0), x[:,1] = (u1 + u0, u0 + 1)
    f = lambda a, b_: (3 - a - b_) ** 2 + 2 * (-1 - a - 1) ** 2 + 0.5 * a * a + 0.5 * b_ * b_
    gr = np.linspace(-0.75, 0.75, 601)
    F = f(gr[:, None], gr[None, :])
    ia, ib = np.unravel_index(np.argmin(F), F.shape)
    assert abs(u.value[0, 0] - gr[ia]) <= 3e-3 and abs(u.value[0, 1] - gr[ib]) <= 3e-3
    assert x.value.shape == (2, 2) and abs(x.value[1, 1] - (u.value[0, 0] + 1.0)) <= 1e-12
    # infeasible: a constant row violated
    y = rec.Variable((1, 1))
    prob = rec.Problem(rec.Minimize(0.0 + rec.quad_form(y[:, 0], np.eye(1))), [y[:, 0] == np.array([2.0]), y[0, :] <= 1.0])
    prob.solve(solver=rec.ECOS)
    assert prob.status == rec.INFEASIBLE and y.value is None
    with pytest.raises(NotImplementedError):
        rec.quad_form(y[:, 0], np.array([[1.0, 2.0], [0.0, 1.0]]))
