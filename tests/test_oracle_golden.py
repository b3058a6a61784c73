"""Pin the CPU oracle (oracle/mpc_oracle.c) against golden vectors produced by the REFERENCE's own
Python functions (tests/golden/make_golden.py).  Stages S1-S3 + S5; CPU only.

Bars: integers (path indices, target_ind, reaches_end, status) bit-exact; reals <= 1e-12 abs
(libm vs numpy transcendental last-ulp differences only).
"""
import numpy as np
import pytest

from conftest import load_golden

TS = (13, 20, 30, 40)


def test_constants_match_reference(oracle):
    c = load_golden("constants.npz")
    p = oracle.make_params(T=int(c["T"]))
    assert p.T == 13 and int(c["NX"]) == 4 and int(c["NU"]) == 2
    assert p.w_perp == float(c["w_perp"]) and p.w_para == float(c["w_para"])
    assert list(p.R) == list(c["R"]) and list(p.Rd) == list(c["Rd"])
    assert list(p.Q_v_yaw) == list(c["Q_v_yaw"])
    # reference scales Qf by T at import (main/lib/mpc.py:28); the oracle does it inside build_qp
    assert [q * p.T for q in p.Qf] == list(c["Qf_scaled"])
    assert p.max_dsteer == float(c["MAX_DSTEER"])      # bit-exact deg2rad
    assert p.max_steer == float(c["MAX_STEER"])
    assert p.max_speed == float(c["MAX_SPEED"]) and p.min_speed == float(c["MIN_SPEED"])
    assert p.max_accel == float(c["MAX_ACCEL"]) and p.max_decel == float(c["MAX_DECEL"])
    assert p.goal_dis == float(c["GOAL_DIS"]) and p.stop_speed == float(c["STOP_SPEED"])
    assert p.L == float(c["L"]) and p.max_iter == int(c["MAX_ITER"])


def test_smooth_yaw(oracle, pkg):
    g = load_golden("smooth_yaw.npz")
    for k, n in enumerate(g["route_len"]):
        raw = g["route_yaw_raw"][k, :n]
        ref = g["route_yaw_smooth"][k, :n]
        assert np.array_equal(oracle.smooth_yaw(raw), ref)
        assert np.array_equal(pkg.synth.smooth_yaw_inplace(raw.copy()), ref)
    for a, b in zip(g["extra_in"], g["extra_out"]):
        assert np.array_equal(oracle.smooth_yaw(a), b)
        assert np.array_equal(pkg.synth.smooth_yaw_inplace(a.copy()), b)


def test_synthetic_routes_are_the_golden_routes(pkg):
    """The fixtures index routes by id; guard against synth.py drifting from what they were made with."""
    g = load_golden("smooth_yaw.npz")
    rs = pkg.synth.make_route_table()
    assert [len(r) for r in rs] == list(g["route_len"])
    for k, r in enumerate(rs):
        assert np.array_equal(r[:, 2], g["route_yaw_raw"][k, :len(r)])
        assert abs(np.linalg.norm(r[1, :2] - r[0, :2]) - 0.083) < 1e-12


def test_nearest_index_random_and_edges(oracle, routes):
    g = load_golden("nearest_index.npz")
    n_short = 0
    for rid, M, start, x, y, fwd, out, status in g["cases"]:
        rid, M, start, fwd, out, status = int(rid), int(M), int(start), int(fwd), int(out), int(status)
        r = routes[rid][:M]
        st, idx = oracle.nearest_index_in_direction(x, y, r[:, 0], r[:, 1], start, bool(fwd))
        assert st == status
        if status == 0:
            assert idx == out
        n_short += (M - start) <= 3
    assert n_short >= 20  # empty / 1 / 2 / 3-point tails are exercised


def test_nearest_index_hairpin_anomaly(oracle):
    g = load_golden("nearest_index.npz")
    hair = g["hairpin"]
    n_anom = 0
    for x, y, out, status in g["hairpin_cases"]:
        st, idx = oracle.nearest_index_in_direction(x, y, hair[:, 0], hair[:, 1], 0, True)
        assert st == int(status)
        if st == 0:
            assert idx == int(out)
        n_anom += st == 2
    assert n_anom >= 3  # the reference's `raise Exception("something wrong")` path is covered


@pytest.mark.parametrize("T", TS)
def test_stages_S1_S3(oracle, routes, T):
    g = load_golden(f"stages_T{T}.npz")
    p = oracle.make_params(T=T)
    n_end = 0
    for b in range(g["x0"].shape[0]):
        r = routes[int(g["path_id"][b])][: int(g["path_len"][b])]
        x, y, v, yaw = g["x0"][b]
        st, xref, idx, rend, tind = oracle.calc_ref_trajectory(p, x, y, v, r[:, 0], r[:, 1], r[:, 2],
                                                               int(g["target_ind_in"][b]))
        assert st == int(g["status"][b])
        if st != 0:
            continue
        assert tind == int(g["target_ind_out"][b])
        assert np.array_equal(rend, g["reaches_end"][b])
        assert np.array_equal(xref, g["xref"][b])          # gathered path points: bit-exact
        assert np.array_equal(idx == len(r) - 1, rend)
        n_end += int(rend.any())
        xbar = oracle.predict_motion(p, g["x0"][b], g["oa"][b], g["od"][b])
        np.testing.assert_allclose(xbar, g["xbar"][b], rtol=0, atol=1e-12)
        if b < g["A"].shape[0]:
            for t in range(T):
                A, B, C = oracle.linear_model_matrix(g["xbar"][b][2, t], g["xbar"][b][3, t], 0.0, p.dt, p.L)
                np.testing.assert_allclose(A, g["A"][b, t], rtol=0, atol=1e-13)
                np.testing.assert_allclose(B, g["B"][b, t], rtol=0, atol=1e-13)
                np.testing.assert_allclose(C, g["C"][b, t], rtol=0, atol=1e-12)
    assert n_end >= 5  # truncated paths where reaches_end fires are present


def test_plant_projector_deviation_goal(oracle, routes):
    g = load_golden("misc.npz")
    p = oracle.make_params(T=13)
    for row, out in zip(g["plant_in"], g["plant_out"]):
        x, y, v, yaw, a, d = row
        np.testing.assert_allclose(oracle.plant_step(p, [x, y, v, yaw], a, d), out, rtol=0, atol=1e-12)
    r = routes[int(g["route_id"])]
    for ti, ox0, oy0, dev in g["deviation"]:
        assert abs(oracle.xref_deviation(r[:, 0], r[:, 1], r[:, 2], int(ti), ox0, oy0) - dev) < 1e-12
    goal = (r[-1, 0], r[-1, 1])
    seen = set()
    for ti, sx, sy, sv, isg in g["goal"]:
        got = oracle.is_goal(p, sx, sy, sv, goal, int(ti), len(r))
        assert got == bool(isg)
        seen.add(got)
    assert seen == {True, False}
