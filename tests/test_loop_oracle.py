"""Row f1 of SURVEY.md 8 -- the loop glue that produces the truncated path: the numpy oracle (oracle/loop_oracle.py)
against golden vectors made by the reference's own resample_curve / state_prediction / check_collision_moving_cars /
get_cutoff_curve_by_position_idx, and against the recorded closed loop of config 1.  CPU only.
Bars: every index (kept points, hit index, cut-off, progress index, path_len) bit-exact; predictions <= 1e-12."""
import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture(scope="module")
def LO(oracle):
    import loop_oracle
    return loop_oracle


def test_car_circles_match_reference(LO):
    g = load_golden("loop_f1.npz")
    radius, offs = LO.car_circles()
    assert radius == float(g["radius"])
    assert np.array_equal(np.array([[offs[0], 0.0], [offs[1], 0.0]]), g["circle_centers"])
    assert LO.extra_cutoff_margin(float(g["dl"])) == int(g["margin"])


def test_resample_prediction_collision_cutoff(LO, routes):
    g = load_golden("loop_f1.npz")
    n_col = 0
    for k in range(len(g["route"])):
        full = routes[int(g["route"][k])]
        idx, v = int(g["idx"][k]), float(g["v"][k])
        detailed = full[idx:]
        mask = LO.resample_mask(detailed[:, :2], LO.ego_resample_dl(len(detailed), v))
        keep = np.flatnonzero(mask)
        assert np.array_equal(keep, g["res_idx"][k][: int(g["n_res"][k])])
        preds = [LO.predict_obstacle(*o) for o in g["obst"][k]]
        np.testing.assert_allclose(np.stack(preds), g["pred"][k], rtol=0, atol=1e-12)
        preds = list(g["pred"][k])      # continue from the reference's predictions: indices must then be identical
        res = detailed[mask]
        col = LO.first_collision_fast(res, detailed, preds)
        flag, cx, cy, first = g["col"][k]
        if flag == 0:
            assert col is None
            continue
        n_col += 1
        assert col is not None and col[2] == int(first) and col[0] == cx and col[1] == cy
        if k % 8 == 0:   # the explicit-loop form agrees with the vectorised one
            assert LO.first_collision(res, detailed, preds) == col
        c = LO.cutoff_index(full, col[0], col[1])
        assert max(idx + 1, c - int(g["margin"])) == int(g["cutoff"][k])
    assert n_col >= 100


def test_closed_loop_pre_tick_matches_reference_loop(LO, routes):
    """Every tick of the recorded mpc_intersection loop: progress index, cut-off / path_len from the oracle's loop glue;
    controls from the oracle MPC on the same inputs."""
    g = load_golden("loop_closed_T13.npz")
    import oracle_py as O
    full = routes[int(g["route_id"])]
    p = O.make_params(T=13, dl=float(np.linalg.norm(full[0, :2] - full[1, :2])))
    oa = od = None
    n_cut = 0
    for row in g["ticks"]:
        x, y, yaw, v, idx_in, prev_len, idx_out, plen, hit, tind_in, tind_out, status, delta, accel, dev = row[:15]
        obst = row[15:].reshape(-1, 6)
        st, idx, path_len, col = LO.loop_pre_tick((x, y, yaw, v), int(idx_in), None if prev_len < 0 else int(prev_len),
                                                   full, obst, p.dl)
        assert st == 0 and idx == int(idx_out) and path_len == int(plen) and (col is not None) == bool(hit)
        n_cut += bool(hit)
        tr = full[:path_len]
        r = O.mpc_step(p, (x, y, yaw, v), tr[:, 0], tr[:, 1], tr[:, 2], int(tind_in), 30 / 3.6, oa=oa, od=od)
        assert r["status"] == int(status) and r["target_ind"] == int(tind_out)
        assert abs(r["od"][0] - delta) <= 1e-12 and abs(r["oa"][0] - accel) <= 1e-12
        oa, od = r["oa"], r["od"]
    assert n_cut >= 20 and bool(g["reached_goal"])


def test_real_route_loop_of_the_reference_replayed_by_the_oracles(LO):
    """tests/golden/loop_real_T13.npz: the mpc_intersection loop run by the REFERENCE's own code end to end -- its planner's route
    for intersection(1, 1), its MPC (under the recording cvxpy stand-in), its obstacle vehicles, collision check and plant
    (tests/golden/make_golden_loop_real.py; nothing under oracle/ took part).  Every tick: the oracle's loop glue reproduces
    progress index, path length and collision flag bit for bit; the oracle's MPC step on the tick's inputs reproduces target_ind
    and the applied controls to 1e-8; the oracle's plant step reproduces the next recorded state."""
    g = load_golden("loop_real_T13.npz")
    import oracle_py as O
    full = g["trajectory_smoothed"]
    assert full.shape == (720, 3) and abs(float(g["dl"]) - 0.083) < 1e-9
    yaw = g["planned"][:, 2].copy()
    assert np.array_equal(O.smooth_yaw(yaw), full[:, 2])                  # MPC.__init__ unwrapped the caller's yaw column in place
    p = O.make_params(T=13, dl=float(g["dl"]))
    assert LO.extra_cutoff_margin(float(g["dl"])) == int(g["margin"])
    oa = od = None
    n_cut, worst = 0, 0.0
    T = g["ticks"]
    for k, row in enumerate(T):
        x, y, yaw_, v, idx_in, prev_len, idx_out, plen, hit, tind_in, tind_out, status, delta, accel, dev = row[:15]
        obst = row[15:].reshape(-1, 6)
        st, idx, path_len, col = LO.loop_pre_tick((x, y, yaw_, v), int(idx_in), None if prev_len < 0 else int(prev_len), full, obst, p.dl)
        assert st == 0 and idx == int(idx_out) and path_len == int(plen) and (col is not None) == bool(hit), k
        n_cut += bool(hit)
        tr = full[:path_len]
        r = O.mpc_step(p, (x, y, yaw_, v), tr[:, 0], tr[:, 1], tr[:, 2], int(tind_in), 30 / 3.6, oa=oa, od=od)
        assert r["status"] == int(status) == 0 and r["target_ind"] == int(tind_out), k
        worst = max(worst, abs(r["od"][0] - delta), abs(r["oa"][0] - accel))
        assert abs(O.xref_deviation(tr[:, 0], tr[:, 1], tr[:, 2], int(tind_out), r["ox"][0], r["oy"][0]) - dev) <= 1e-9
        oa, od = r["oa"], r["od"]
        nxt = O.plant_step(p, np.array([x, y, v, yaw_]), accel, delta)      # [x, y, v, yaw] -> the next recorded State
        want = T[k + 1][:4] if k + 1 < len(T) else g["final"]
        np.testing.assert_allclose(nxt[[0, 1, 3, 2]], want, rtol=0, atol=1e-12)
    assert worst <= 1e-8, worst
    assert n_cut == 35 and len(T) == 91 and bool(g["reached_goal"])
    assert O.is_goal(p, *g["final"][[0, 1, 3]], (full[-1, 0], full[-1, 1]), int(T[-1][10]), int(T[-1][7]))


BIKE = (1.0, 0.45, 0.64)     # BicycleRealDimensions: wheelbase, bounding-box width, extra length (lib/car_dimensions.py:92-100)


def test_bicycle_obstacle_glue_vs_reference(LO, routes):
    """The same glue with an obstacle of another shape (the cyclist of scenarios/overtaking_cyclist_bidirectional_road.py):
    prediction with the bicycle's wheelbase, check_collision_moving_bicycle (min_distance = car radius + bicycle radius,
    the bicycle's circle centres), cut-off with the scenario's margin of 2 * ceil(radius / dl)."""
    g = load_golden("loop_bicycle.npz")
    orad, ooffs = LO.car_circles(*BIKE)
    assert orad == float(g["bike_radius"]) and float(g["bike_L"]) == BIKE[0]
    np.testing.assert_allclose(np.array([[ooffs[0], 0.0], [ooffs[1], 0.0]]), g["bike_circle_centers"], rtol=0, atol=1e-15)
    assert (LO.extra_cutoff_margin(float(g["dl"])) // 4) * 2 == int(g["margin"])
    n_col = 0
    for k in range(len(g["route"])):
        full = routes[int(g["route"][k])]
        idx, v = int(g["idx"][k]), float(g["v"][k])
        detailed = full[idx:]
        res = detailed[LO.resample_mask(detailed[:, :2], LO.ego_resample_dl(len(detailed), v))]
        preds = [LO.predict_obstacle(*o, L=BIKE[0]) for o in g["obst"][k]]
        np.testing.assert_allclose(np.stack(preds), g["pred"][k], rtol=0, atol=1e-12)
        col = LO.first_collision_fast(res, detailed, list(g["pred"][k]), obst_dims=BIKE)
        flag, cx, cy, first = g["col"][k]
        if flag == 0:
            assert col is None
        else:
            n_col += 1
            assert col is not None and col[2] == int(first) and col[0] == cx and col[1] == cy
            if k % 8 == 0:
                assert LO.first_collision(res, detailed, list(g["pred"][k]), obst_dims=BIKE) == col
        x, y, yaw = full[idx]
        st, i2, plen, _ = LO.loop_pre_tick((x, y, yaw, v), idx, idx + 1, full, g["obst"][k], float(g["dl"]), obst_dims=BIKE,
                                           margin_factor=2)
        assert st == 0 and i2 == idx and plen == int(g["cutoff"][k])
    assert 40 <= n_col <= 110


def test_squared_distance_threshold_is_exactly_the_sqrt_comparison():
    """The collision rows on the device compare dx*dx + dy*dy with the largest double whose correctly rounded square root is
    <= min_distance (csrc/jsim_mpc.hip: jsim_sqrt_threshold) instead of comparing the square root with min_distance as
    numpy does (collision_avoidance.py:99).  Same decision for every double: checked here on the doubles around thr^2."""
    for thr in (2 * 2.0 / np.sqrt(2.0), 2.0 / np.sqrt(2.0) + 0.45 / np.sqrt(2.0), 0.001, 1.0, 3.0):
        x = thr * thr
        while np.sqrt(x) > thr:
            x = np.nextafter(x, 0.0)
        while np.sqrt(np.nextafter(x, np.inf)) <= thr:
            x = np.nextafter(x, np.inf)
        d2 = x
        for _ in range(2000):                       # 2000 doubles below, then 2000 above the threshold
            d2 = np.nextafter(d2, 0.0)
        for _ in range(4000):
            assert (np.sqrt(d2) <= thr) == (d2 <= x)
            d2 = np.nextafter(d2, np.inf)
        rng = np.random.default_rng(1)
        far = rng.uniform(0.0, 4.0 * x, 20000)
        assert np.array_equal(np.sqrt(far) <= thr, far <= x)
