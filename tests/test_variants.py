"""SURVEY.md 8 row f3 -- the data-only MPC variants.  CPU: module constants of the drop-in equal the reference's,
the oracle's reference window with a speed reference equals `lib.mpc_with_speed._calc_ref_trajectory` (golden).
GPU: variant configuration (weights, MAX_DECEL, cv + cut-off) against the oracle through the C-ABI; per-solve
re-configuration (mpc_sensitivity)."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def _variant_params(oracle, T=13):
    return oracle.make_params(T=T, config={"w_perp": 10.0, "w_para": 1.0, "Q_v_yaw": [20.0, 0.5], "MAX_DECEL": -5,
                                           "STOP_SPEED": 0.5 / 3.6})


def test_variant_constants(pkg):
    g = load_golden("variant_with_speed.npz")
    m = pkg.mpc_with_speed
    assert m.T == int(g["c_T"]) and m.MAX_DECEL == float(g["c_MAX_DECEL"]) and m.MAX_SPEED == float(g["c_MAX_SPEED"])
    assert np.array_equal(np.diag(m.Q_v_yaw), g["c_Q_v_yaw"]) and np.array_equal(np.diag(m.Qf), g["c_Qf_scaled"])
    assert np.array_equal(np.diag(m.R), g["c_R"]) and np.array_equal(np.diag(m.Rd), g["c_Rd"])
    assert m.STOP_SPEED == float(g["c_STOP_SPEED"]) and m.MAX_DSTEER == float(g["c_MAX_DSTEER"])
    assert m.config.w_perp == 10.0 and m.config.Q_v_yaw == [20.0, 0.5] and m.config.MAX_DECEL == -5.0


def test_oracle_reference_window_with_speed_reference(oracle, routes):
    g = load_golden("variant_with_speed.npz")
    p = _variant_params(oracle)
    n_zero = 0
    for b in range(len(g["x0"])):
        r = routes[int(g["path_id"][b])][: int(g["path_len"][b])]
        cv = np.full(len(r), float(g["c_MAX_SPEED"]))
        cut = int(g["cutoff"][b])
        x, y, v, yaw = g["x0"][b]
        res = oracle.mpc_step(p, (x, y, yaw, v), r[:, 0], r[:, 1], r[:, 2], int(g["target_ind_in"][b]), 30 / 3.6,
                              cv=cv, cv_cut=cut if cut != 999 else -1)
        assert res["status"] in (0, 1)
        assert res["target_ind"] == int(g["target_ind_out"][b])
        assert np.array_equal(res["xref"], g["xref"][b])           # incl. row 2 = cv[idx] with the cut-off
        n_zero += bool((g["xref"][b][2] == 0).any())
    assert n_zero >= 5


@pytest.mark.gpu
def test_variant_gpu_vs_oracle(pkg, oracle, routes):
    T, B = 13, 96
    m = pkg.mpc_with_speed
    batch = pkg.synth.make_ego_batch(routes, B, T, seed=9, truncate=True, near_end_frac=0.3)
    rng = np.random.default_rng(3)
    cut = np.where(rng.random(B) < 0.6, rng.integers(0, 720, size=B), -1).astype(np.int32)
    cvs = [np.full(len(r), m.MAX_SPEED) for r in routes]
    eng = pkg.BatchedMPC(routes, batch.path_id, dl=pkg.synth.DL, T=T, speed=batch.speed, smooth=False, config=m.config, cv=cvs)
    eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    eng.set_speed_cutoff(cut)
    eng.solve(torch.from_numpy(batch.x0).to(eng.device))
    torch.cuda.synchronize()
    p = _variant_params(oracle, T)
    cx, cy, cyaw, off = pkg.synth.pack_paths(routes)
    ref = oracle.mpc_step_batch(p, batch.x0, batch.path_id, batch.path_len, batch.speed, cx, cy, cyaw, off, batch.target_ind,
                                batch.oa, batch.od, cv=np.concatenate(cvs), cv_cut=cut)
    assert np.array_equal(eng.status.cpu().numpy(), ref["status"])
    np.testing.assert_array_equal(eng.xref.cpu().numpy(), ref["xref"])
    assert float(np.abs(ref["xref"][:, 2]).max()) > 0 and (ref["xref"][:, 2] == 0).any()
    ok = ref["status"] == 0
    assert np.abs(eng.oa.cpu().numpy() - ref["oa"])[ok].max() <= 1e-8
    assert np.abs(eng.od.cpu().numpy() - ref["od"])[ok].max() <= 1e-8
    assert np.array_equal(eng.active_mask.cpu().numpy().view(np.uint32), ref["active_mask"])
    # with a speed weight of 20 the reference speed matters: the plain controller on the same inputs differs
    plain = pkg.BatchedMPC(routes, batch.path_id, dl=pkg.synth.DL, T=T, speed=batch.speed, smooth=False)
    plain.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    plain.solve(torch.from_numpy(batch.x0).to(eng.device))
    assert float((plain.oa - eng.oa).abs().max()) > 1e-2


@pytest.mark.gpu
def test_variant_dropin_and_reconfiguration(pkg, oracle, routes):
    m = pkg.mpc_with_speed
    r = routes[0].copy()
    cv = np.full(len(r), m.MAX_SPEED)
    mpc = m.MPC(cx=r[:, 0], cy=r[:, 1], cv=cv, cyaw=r[:, 2].copy(), dl=pkg.synth.DL, car_dimensions=pkg.BicycleModelDimensions())
    mpc.set_trajectory_fromarray(r, cutoff_idx=150)
    st = pkg.State(x=r[60, 0], y=r[60, 1], yaw=r[60, 2], v=5.0)
    mpc.target_ind = 55
    di, ai = mpc.step(st)
    p = _variant_params(oracle)
    cvz = cv.copy(); cvz[150:] = 0
    ref = oracle.mpc_step(p, (st.x, st.y, st.yaw, st.v), r[:, 0], r[:, 1], r[:, 2], 55, 30 / 3.6, cv=cvz)
    assert mpc.status == ref["status"] == 0
    assert abs(di - ref["od"][0]) <= 1e-8 and abs(ai - ref["oa"][0]) <= 1e-8
    assert np.array_equal(mpc.xref, ref["xref"])
    u1 = np.concatenate([mpc.oa, mpc.odelta])
    # mpc_sensitivity: weights re-read before every solve -> update_config between solves
    eng = mpc._engine
    from dataclasses import replace
    cfg2 = replace(m.config, w_perp=40.0, Q_v_yaw=[5.0, 0.5])
    eng.update_config(cfg2)
    mpc.oa = mpc.odelta = None
    mpc.target_ind = 55
    di2, ai2 = mpc.step(st)
    p2 = oracle.make_params(T=13, config={"w_perp": 40.0, "Q_v_yaw": [5.0, 0.5], "MAX_DECEL": -5})
    ref2 = oracle.mpc_step(p2, (st.x, st.y, st.yaw, st.v), r[:, 0], r[:, 1], r[:, 2], 55, 30 / 3.6, cv=cvz)
    assert abs(di2 - ref2["od"][0]) <= 1e-8 and abs(ai2 - ref2["oa"][0]) <= 1e-8
    np.testing.assert_allclose(np.concatenate([mpc.oa, mpc.odelta]), np.concatenate([ref2["oa"], ref2["od"]]), rtol=0, atol=1e-8)
    assert np.abs(np.concatenate([mpc.oa, mpc.odelta]) - u1).max() > 1e-4   # the new weights changed the solution


def test_sensitivity_module_constants(pkg):
    """lib.mpc_sensitivity's import-time constants and its shipped JSON equal the reference's config/mpc_config_sensitivity.json
    (values recorded here from the reference file: R = (0.1, 0.01), Rd = (10, 10), the rest as mpc_config.json)."""
    m = pkg.mpc_sensitivity
    assert (m.NX, m.NU, m.T) == (4, 2, 13) and m.GOAL_DIS == 1.5 and m.STOP_SPEED == 0.1389
    assert m.MAX_ACCEL == 2.0 and m.MAX_DECEL == -10 and abs(m.MAX_DSTEER - np.deg2rad(30.0)) == 0
    c = m._load(m.CONFIG_PATH)
    assert c.R == [0.1, 0.01] and c.Rd == [10.0, 10.0] and c.w_perp == 20.0 and c.w_para == 1.0
    assert c.Q_v_yaw == [0.0, 0.5] and c.Qf == [1.0, 1.0, 0.0, 0.5] and c.MAX_DECEL == -10.0 and c.T == 13


@pytest.mark.gpu
def test_sensitivity_dropin_rereads_its_json(pkg, oracle, routes, tmp_path, monkeypatch):
    """main/lib/mpc_sensitivity.py re-reads its JSON inside every solve; the analysis scripts rewrite the file between
    runs.  Same here: rewrite the file, the next step() solves with the new weights / limits (checked against the
    oracle), the speed rows use Simulation.MAX_SPEED."""
    import json
    m = pkg.mpc_sensitivity
    raw = json.load(open(m.CONFIG_PATH))
    path = tmp_path / "mpc_config_sensitivity.json"
    path.write_text(json.dumps(raw))
    monkeypatch.setattr(m, "CONFIG_PATH", str(path))
    r = routes[2].copy()
    mpc = m.MPC(cx=r[:, 0], cy=r[:, 1], cyaw=r[:, 2].copy(), dl=pkg.synth.DL, car_dimensions=pkg.BicycleModelDimensions())
    st = pkg.State(x=r[40, 0] + 0.2, y=r[40, 1] - 0.1, yaw=r[40, 2] + 0.05, v=6.0)
    mpc.target_ind = 36
    di, ai = mpc.step(st)
    base = {k: raw[k] for k in ("w_perp", "w_para", "R", "Rd", "Q_v_yaw", "Qf", "MAX_DSTEER", "MAX_ACCEL", "MAX_DECEL")}
    ref = oracle.mpc_step(oracle.make_params(T=13, config=base), (st.x, st.y, st.yaw, st.v), r[:, 0], r[:, 1], r[:, 2], 36, 30 / 3.6)
    assert mpc.status == ref["status"] == 0
    np.testing.assert_allclose(np.concatenate([mpc.oa, mpc.odelta]), np.concatenate([ref["oa"], ref["od"]]), rtol=0, atol=1e-8)
    u1 = np.concatenate([mpc.oa, mpc.odelta])
    raw2 = dict(raw, w_perp=5.0, Rd=[0.01, 1.0], MAX_ACCEL=0.5, MAX_DSTEER=10.0)
    path.write_text(json.dumps(raw2))
    mpc.oa = mpc.odelta = None
    mpc.target_ind = 36
    mpc.step(st)
    cfg2 = {k: raw2[k] for k in base}
    ref2 = oracle.mpc_step(oracle.make_params(T=13, config=cfg2), (st.x, st.y, st.yaw, st.v), r[:, 0], r[:, 1], r[:, 2], 36, 30 / 3.6)
    assert mpc.status == ref2["status"] == 0
    np.testing.assert_allclose(np.concatenate([mpc.oa, mpc.odelta]), np.concatenate([ref2["oa"], ref2["od"]]), rtol=0, atol=1e-8)
    assert mpc.oa.max() <= 0.5 + 1e-9 and np.abs(np.concatenate([mpc.oa, mpc.odelta]) - u1).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("T", (20, 30, 40, 24))
def test_per_ego_weights_one_batch(pkg, oracle, routes, T):
    """A sensitivity sweep as one launch: every ego of the batch carries its own weights and limits
    (jsim_mpc_set_ego_config).  T = 20 / 30 / 40 / 24 = the one-wave kernel with resident and with virtual speed
    rows, the four-wave kernel and the LDS kernel.  Each ego against the oracle run
    with that ego's parameters; switching the table off restores the engine's configuration."""
    from dataclasses import replace
    B = 48
    rng = np.random.default_rng(17)
    batch = pkg.synth.make_ego_batch(routes, B, T, seed=4, truncate=True, near_end_frac=0.2)
    base = pkg.MPCConfig.from_json()
    cfgs = [replace(base, T=T, w_perp=float(rng.uniform(5, 40)), w_para=float(rng.uniform(0.5, 3)),
                    R=[float(rng.uniform(0.005, 0.2)), float(rng.uniform(0.005, 0.05))],
                    Rd=[float(rng.uniform(0.005, 10)), float(rng.uniform(0.5, 10))],
                    Q_v_yaw=[float(rng.choice([0.0, 2.0])), float(rng.uniform(0.1, 1.0))],
                    Qf=[float(rng.uniform(0.5, 2)), float(rng.uniform(0.5, 2)), 0.0, float(rng.uniform(0.2, 1.0))],
                    MAX_DSTEER=float(rng.uniform(10, 60)), MAX_ACCEL=float(rng.uniform(0.5, 3)),
                    MAX_DECEL=float(-rng.uniform(3, 10))) for _ in range(B)]
    eng = pkg.BatchedMPC(routes, batch.path_id, dl=pkg.synth.DL, T=T, speed=batch.speed, smooth=False)
    eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    x0 = torch.from_numpy(batch.x0).to(eng.device)
    eng.solve(x0)
    plain = (eng.oa.clone(), eng.od.clone())
    eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    eng.set_ego_configs(cfgs)
    eng.solve(x0)
    torch.cuda.synchronize()
    oa, od, st = eng.oa.cpu().numpy(), eng.od.cpu().numpy(), eng.status.cpu().numpy()
    am = eng.active_mask.cpu().numpy().view(np.uint32)
    cx, cy, cyaw, off = pkg.synth.pack_paths(routes)
    n_ok = 0
    for b in range(B):
        c = cfgs[b]
        p = oracle.make_params(T=T, config={"w_perp": c.w_perp, "w_para": c.w_para, "R": c.R, "Rd": c.Rd, "Q_v_yaw": c.Q_v_yaw,
                                            "Qf": c.Qf, "MAX_DSTEER": c.MAX_DSTEER, "MAX_ACCEL": c.MAX_ACCEL, "MAX_DECEL": c.MAX_DECEL})
        o = off[batch.path_id[b]]
        n = batch.path_len[b]
        r = oracle.mpc_step(p, (batch.x0[b, 0], batch.x0[b, 1], batch.x0[b, 3], batch.x0[b, 2]), cx[o:o + n], cy[o:o + n],
                            cyaw[o:o + n], int(batch.target_ind[b]), batch.speed[b], oa=batch.oa[b], od=batch.od[b])
        assert st[b] == r["status"], b
        if st[b] == 0:
            n_ok += 1
            assert np.abs(oa[b] - r["oa"]).max() <= 1e-7 and np.abs(od[b] - r["od"]).max() <= 1e-7, b
            assert np.array_equal(am[b], np.asarray(r["active_mask"], dtype=np.uint32)), b
            assert oa[b].max() <= c.MAX_ACCEL + 1e-9 and oa[b].min() >= c.MAX_DECEL - 1e-9
    assert n_ok >= B - 6
    assert float((eng.oa - plain[0]).abs().max()) > 1e-2       # the table changed the solutions
    eng.set_ego_configs(None)
    eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    eng.solve(x0)
    assert torch.equal(eng.oa, plain[0]) and torch.equal(eng.od, plain[1])


@pytest.mark.gpu
def test_new_ref_loop_glue_speed_cutoff_mode(pkg, routes):
    """main/scenarios/mpc_intersection_new_ref.py keeps the full path and hands the collision cut-off to mpc_with_speed
    (cutoff_idx): PreTick(mode="speed_cutoff") writes the same index the truncating glue writes to path_len into the
    controller's speed cut-off instead, and the solve then sees a zero speed reference from that index on."""
    T, B = 13, 32
    m = pkg.mpc_with_speed
    batch = pkg.synth.make_ego_batch(routes, B, T, seed=12, truncate=False, near_end_frac=0.0)
    cvs = [np.full(len(r), m.MAX_SPEED) for r in routes]
    x0 = torch.from_numpy(batch.x0).cuda()
    # obstacles parked on the first ego's path ahead of it, so that several egos see a collision
    r0 = routes[int(batch.path_id[0])]
    j = min(int(batch.target_ind[0]) + 150, len(r0) - 1)
    obst = torch.tensor([[r0[j, 0], r0[j, 1], 0.0, r0[j, 2] + 1.2, 0.0, 0.0],
                         [r0[j, 0] + 1.0, r0[j, 1] - 2.0, 2.0, r0[j, 2] - 1.0, 0.0, 0.0]], dtype=torch.float64, device="cuda")
    def engine():
        e = pkg.BatchedMPC(routes, batch.path_id, dl=pkg.synth.DL, T=T, speed=batch.speed, smooth=False, config=m.config, cv=cvs)
        e.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
        return e
    ea, eb = engine(), engine()
    pa = pkg.PreTick(ea, frame_window=20)
    pb = pkg.PreTick(eb, frame_window=20, mode="speed_cutoff")
    for pre in (pa, pb):
        pre.predict(obst)
        pre.run(x0)
    torch.cuda.synchronize()
    assert int(pa.col_flag.sum().item()) >= 1                                    # somebody is cut off
    assert torch.equal(pb.cut, ea.path_len) and torch.equal(pa.traj_idx, pb.traj_idx)
    assert torch.equal(eb.path_len, torch.from_numpy(batch.path_len).cuda())      # the path itself stays whole
    eb.solve(x0)
    torch.cuda.synchronize()
    ridx_zero = (eb.xref[:, 2] == 0).cpu().numpy()
    cut = pb.cut.cpu().numpy()
    assert ridx_zero[cut < batch.path_len].any() and not ridx_zero[(cut >= batch.path_len)].any()
    with pytest.raises(ValueError):
        pkg.PreTick(pkg.BatchedMPC(routes, batch.path_id, dl=pkg.synth.DL, T=T, smooth=False), mode="speed_cutoff")


@pytest.mark.gpu
def test_fused_scenario_loop_speed_cutoff_glue(pkg, routes):
    """The new_ref glue (cut-off -> the controller's speed reference, path never truncated) through ScenarioLoop.run: one fused
    call against the same ticks driven from the host, every buffer bit-identical."""
    T, B, K = 13, 40, 12
    m = pkg.mpc_with_speed
    cvs = [np.full(len(r), m.MAX_SPEED) for r in routes]
    specs = [dict(direction=1, turning=False, speed=25 / 3.6, offset=None), dict(direction=-1, turning=True, speed=20 / 3.6, offset=1.0)]
    outs = []
    for fused in (False, True):
        batch = pkg.synth.make_ego_batch(routes, B, T, seed=19, truncate=False, near_end_frac=0.0)
        eng = pkg.BatchedMPC(routes, batch.path_id, dl=pkg.synth.DL, T=T, speed=batch.speed, smooth=False, config=m.config, cv=cvs)
        eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
        sc = pkg.ScenarioLoop(eng, torch.from_numpy(batch.x0).cuda(), specs, hist_cap=K, frame_window=20, mode="speed_cutoff")
        if fused:
            sc.run(K)
        else:
            for _ in range(K):
                sc.tick()
        torch.cuda.synchronize()
        outs.append(dict(x0=sc.loop.x0.clone(), cut=sc.pre.cut.clone(), path_len=eng.path_len.clone(), traj_idx=sc.pre.traj_idx.clone(),
                         prev=sc.pre.prev_len.clone(), col=sc.pre.col_flag.clone(), oa=eng.oa.clone(), od=eng.od.clone(),
                         hist=sc.loop.hist.clone(), xref=eng.xref.clone(), obs=sc.obst.state.clone()))
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert torch.equal(outs[1]["path_len"], torch.from_numpy(batch.path_len).cuda())          # the path stays whole
    assert int((outs[1]["cut"] < outs[1]["path_len"]).sum()) > 0                                # somebody got a speed cut-off


# ------------------------------------------------------------------------------------------------
# lib.mpc_jerk: the acceleration-state variant (NX = 5, free acc_0: 2T + 1 decision variables)
# ------------------------------------------------------------------------------------------------
JERK_ORACLE_CFG = {"NX": 5, "w_perp": 10.0, "w_para": 1.0, "R": [0.01, 0.01], "Rd": [0.3, 1.0], "Q_v_yaw": [0.0, 0.5],
                   "Qf": [1.0, 1.0, 0.0, 0.5], "STOP_SPEED": 0.5 / 3.6, "MAX_DECEL": -5, "JERK_WEIGHT": 1.0}


def test_jerk_module_constants(pkg):
    """Names and values of main/lib/mpc_jerk.py:16-39."""
    m = pkg.mpc_jerk
    assert (m.NX, m.NU, m.T) == (5, 2, 13) and m.MAX_DECEL == -5 and m.MAX_ACCEL == 2.0 and m.jerk_penalty_weight == 1
    assert np.array_equal(m.Rd, np.diag([.3, 1.0])) and np.array_equal(m.Qf, np.diag([1.0, 1.0, 0., 0.5, 0]) * 13)
    assert m.config.NX == 5 and m.config.w_perp == 10.0 and m.config.JERK_WEIGHT == 1.0
    with pytest.raises(ValueError):
        from dataclasses import asdict
        import json, tempfile, os
        bad = dict(asdict(pkg.MPCConfig()), NX=6)
        with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
            json.dump(bad, f)
        try:
            pkg.MPCConfig.from_json(f.name)
        finally:
            os.unlink(f.name)


def test_jerk_golden_from_reference_functions(pkg, oracle, routes):
    """What the reference's own lib.mpc_jerk functions returned (tests/golden/make_golden_jerk.py): module constants; the
    5x5 / 5x2 / 5 linear model against the numpy restatement that pins the oracle's condensed QP (so the chain reference ->
    restatement -> oracle -> HIP is closed for the model); the five-row reference window and rollout against the oracle's
    stages (rows 0..3 identical to lib.mpc's, the fifth rows zero)."""
    import qp_sparse_numpy as QS
    g = load_golden("variant_jerk.npz")
    m = pkg.mpc_jerk
    assert (int(g["c_NX"]), int(g["c_NU"]), int(g["c_T"])) == (m.NX, m.NU, m.T)
    assert np.array_equal(g["c_R"], np.diag(m.R)) and np.array_equal(g["c_Rd"], np.diag(m.Rd))
    assert np.array_equal(g["c_Q_v_yaw"], np.diag(m.Q_v_yaw)) and np.array_equal(g["c_Qf_scaled"], np.diag(m.Qf))
    assert float(g["c_MAX_DECEL"]) == m.MAX_DECEL and float(g["c_MAX_ACCEL"]) == m.MAX_ACCEL
    assert float(g["c_MAX_DSTEER"]) == m.MAX_DSTEER and float(g["c_jerk_penalty_weight"]) == m.jerk_penalty_weight
    assert float(g["c_STOP_SPEED"]) == m.STOP_SPEED and float(g["c_GOAL_DIS"]) == m.GOAL_DIS
    for k in range(len(g["lm_v"])):
        A, B, C = QS.linear_model_jerk(g["lm_v"][k], g["lm_phi"][k], g["lm_delta"][k], pkg.synth.DT, 2.86)
        np.testing.assert_allclose(A, g["lm_A"][k], rtol=0, atol=1e-15)
        np.testing.assert_allclose(B, g["lm_B"][k], rtol=0, atol=1e-15)
        np.testing.assert_allclose(C, g["lm_C"][k], rtol=0, atol=1e-15)
    assert np.all(g["lm_A"][:, 4, 4] == 1.0) and np.all(g["lm_A"][:, 2, 4] == pkg.synth.DT) and np.all(g["lm_B"][:, 4, 0] == pkg.synth.DT)
    p = oracle.make_params(T=13, config=JERK_ORACLE_CFG)
    for b in range(len(g["x0"])):
        r = routes[int(g["path_id"][b])][: int(g["path_len"][b])]
        x, y, v, yaw = g["x0"][b]
        st_, xref, idx, rend, tind = oracle.calc_ref_trajectory(p, x, y, v, r[:, 0], r[:, 1], r[:, 2], int(g["target_ind_in"][b]))
        assert st_ == 0 and tind == g["target_ind_out"][b]
        np.testing.assert_array_equal(xref, g["xref"][b][:4])
        assert np.array_equal(rend.astype(bool), g["reaches_end"][b])
        xbar = oracle.predict_motion(p, g["x0"][b], g["oa"][b], g["od"][b])
        np.testing.assert_allclose(xbar, g["xbar"][b][:4], rtol=0, atol=1e-12)
    assert not g["xref"][:, 4].any() and not g["xbar"][:, 4].any()


@pytest.mark.gpu
@pytest.mark.parametrize("T", (13, 20, 40))
def test_jerk_variant_gpu_vs_oracle(pkg, oracle, routes, T):
    """The HIP path with 2T + 1 variables against the oracle's (itself pinned by the variant's sparse form,
    tests/test_oracle_qp.py): statuses, reference window, u*, predicted states, active sets; the condensed (H, g) of a few
    egos.  T = 40 takes the two-rows-per-lane instantiation (81 variables)."""
    from dataclasses import replace
    B = 64 if T <= 20 else 32
    cfg = replace(pkg.mpc_jerk.config, T=T)
    batch = pkg.synth.make_ego_batch(routes, B, T, seed=13, truncate=True, near_end_frac=0.25)
    vmax = np.full(B, cfg.MAX_SPEED)                    # x[2,:] <= Simulation.MAX_SPEED (mpc_jerk.py:194)
    batch.x0[3, 2] = cfg.MAX_SPEED + 0.5                # infeasible constant row
    eng = pkg.BatchedMPC(routes, batch.path_id, dl=pkg.synth.DL, T=T, speed=vmax, smooth=False, config=cfg)
    eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    n = 2 * T + 1
    f = dict(dtype=torch.float64, device=eng.device)
    dbg = {"H": torch.zeros(B, n, n, **f), "g": torch.zeros(B, n, **f), "lam": torch.zeros(B, 8 * T, **f)}
    eng.solve(torch.from_numpy(batch.x0).to(eng.device), debug=dbg)
    torch.cuda.synchronize()
    p = oracle.make_params(T=T, config=JERK_ORACLE_CFG)
    cx, cy, cyaw, off = pkg.synth.pack_paths(routes)
    ref = oracle.mpc_step_batch(p, batch.x0, batch.path_id, batch.path_len, vmax, cx, cy, cyaw, off, batch.target_ind,
                                batch.oa, batch.od)
    st = eng.status.cpu().numpy()
    assert np.array_equal(st, ref["status"]) and st[3] == 1
    assert np.array_equal(eng.target_ind.cpu().numpy(), ref["target_ind"])
    np.testing.assert_array_equal(eng.xref.cpu().numpy(), ref["xref"])
    ok = st == 0
    assert ok.sum() >= B - 6
    err = max(np.abs(eng.oa.cpu().numpy() - ref["oa"])[ok].max(), np.abs(eng.od.cpu().numpy() - ref["od"])[ok].max())
    print(f"T={T}: max|du|={err:.2e}, mean n_iter {ref['n_iter'][ok].mean():.1f}, identical n_iter "
          f"{(eng.n_iter.cpu().numpy() == ref['n_iter']).mean() * 100:.0f} %")
    assert err <= 1e-4 and err <= 1e-6, err
    for name in ("ox", "oy", "ov", "oyaw"):
        np.testing.assert_allclose(getattr(eng, name).cpu().numpy()[ok], ref[name][ok], rtol=0, atol=1e-6)
    assert np.array_equal(eng.active_mask.cpu().numpy().view(np.uint32), ref["active_mask"])
    assert np.unpackbits(ref["active_mask"].view(np.uint8), axis=1).sum() > B     # the constraints do bind
    H = dbg["H"].cpu().numpy(); g = dbg["g"].cpu().numpy()
    for b in np.flatnonzero(ok)[:8]:
        o = off[batch.path_id[b]]
        r = oracle.mpc_step(p, (batch.x0[b, 0], batch.x0[b, 1], batch.x0[b, 3], batch.x0[b, 2]),
                            cx[o:o + batch.path_len[b]], cy[o:o + batch.path_len[b]], cyaw[o:o + batch.path_len[b]],
                            int(batch.target_ind[b]), vmax[b], oa=batch.oa[b], od=batch.od[b], want_qp=True)
        Hl = np.tril(H[b]); Hl = Hl + np.tril(Hl, -1).T
        assert np.abs(Hl - r["H"]).max() <= 1e-9 * np.abs(r["H"]).max()
        assert np.abs(g[b] - r["g"]).max() <= 1e-9 * max(1.0, np.abs(r["g"]).max())
    # the acceleration state matters: the stock controller with the same weights gives another answer
    plain = pkg.BatchedMPC(routes, batch.path_id, dl=pkg.synth.DL, T=T, speed=vmax, smooth=False, config=replace(cfg, NX=4))
    plain.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    plain.solve(torch.from_numpy(batch.x0).to(eng.device))
    assert float((plain.oa - eng.oa)[torch.from_numpy(ok).to(eng.device)].abs().max()) > 1e-2


@pytest.mark.gpu
def test_jerk_dropin_closed_loop(pkg, oracle, routes):
    """lib.mpc_jerk.MPC(cx, cy, cyaw, dl, car_dimensions, dt) through 25 ticks of the per-vehicle loop against the oracle
    driven the same way (warm start and target index carried; plant = Simulation.step)."""
    m = pkg.mpc_jerk
    r = routes[2].copy()
    mpc = m.MPC(cx=r[:, 0], cy=r[:, 1], cyaw=r[:, 2].copy(), dl=pkg.synth.DL, car_dimensions=pkg.BicycleModelDimensions())
    p = oracle.make_params(T=13, config=JERK_ORACLE_CFG)
    state = np.array([r[5, 0], r[5, 1], 0.0, r[5, 2]])             # x, y, v, yaw
    tind, oa, od = 0, None, None
    for k in range(25):
        st = pkg.State(x=state[0], y=state[1], yaw=state[3], v=state[2])
        di, ai = mpc.step(st)
        ref = oracle.mpc_step(p, (state[0], state[1], state[3], state[2]), r[:, 0], r[:, 1], r[:, 2], tind, 30 / 3.6, oa=oa, od=od)
        assert mpc.status == ref["status"] == 0
        assert mpc.target_ind == ref["target_ind"]
        assert abs(di - ref["od"][0]) <= 1e-7 and abs(ai - ref["oa"][0]) <= 1e-7, (k, di, ai)
        assert mpc.active_constraints == ref["active"]
        tind, oa, od = ref["target_ind"], ref["oa"], ref["od"]
        state = oracle.plant_step(p, state, ref["oa"][0], ref["od"][0])
    assert state[2] > 1.0                                          # it drove off


@pytest.mark.gpu
def test_with_speed_dropin_resets_the_constructor_reference(pkg, oracle, routes):
    """main/lib/mpc_with_speed.py:276-282: EVERY set_trajectory_fromarray rebuilds cv = MAX_SPEED (and zeroes it from
    cutoff_idx on, a negative index counting from the end like any Python slice) -- also when the constructor was given
    another speed reference and the trajectory is a prefix of the bound path; is_goal uses this module's STOP_SPEED."""
    m = pkg.mpc_with_speed
    r = routes[3].copy()
    cv0 = np.linspace(2.0, 6.0, len(r))                         # a constructor reference that is NOT MAX_SPEED
    mpc = m.MPC(cx=r[:, 0], cy=r[:, 1], cv=cv0, cyaw=r[:, 2].copy(), dl=pkg.synth.DL, car_dimensions=pkg.BicycleModelDimensions())
    p = _variant_params(oracle)
    st = pkg.State(x=r[80, 0], y=r[80, 1], yaw=r[80, 2], v=4.0)
    # 1. before any set_trajectory_fromarray the constructor's cv is the reference
    mpc.target_ind = 76
    mpc.step(st)
    ref0 = oracle.mpc_step(p, (st.x, st.y, st.yaw, st.v), r[:, 0], r[:, 1], r[:, 2], 76, 30 / 3.6, cv=cv0)
    assert np.array_equal(mpc.xref, ref0["xref"]) and float(mpc.xref[2].max()) < 6.0
    # 2. a prefix of the bound path: the reference becomes MAX_SPEED, zero over the last 40 points (cutoff_idx = -40)
    n = 400
    mpc.oa = mpc.odelta = None
    mpc.target_ind = 76
    mpc.set_trajectory_fromarray(r[:n], cutoff_idx=-40)
    assert np.all(mpc.cv[:n - 40] == m.MAX_SPEED) and np.all(mpc.cv[n - 40:] == 0) and len(mpc.cv) == n
    mpc.step(st)
    cv1 = np.full(n, m.MAX_SPEED); cv1[-40:] = 0
    ref1 = oracle.mpc_step(p, (st.x, st.y, st.yaw, st.v), r[:n, 0], r[:n, 1], r[:n, 2], 76, 30 / 3.6, cv=cv1)
    assert mpc.status == ref1["status"] == 0
    assert np.array_equal(mpc.xref, ref1["xref"]) and float(mpc.xref[2].max()) == m.MAX_SPEED
    np.testing.assert_allclose(np.concatenate([mpc.oa, mpc.odelta]), np.concatenate([ref1["oa"], ref1["od"]]), rtol=0, atol=1e-8)
    # 3. near the cut: part of the window has a zero speed reference
    st2 = pkg.State(x=r[345, 0], y=r[345, 1], yaw=r[345, 2], v=5.0)
    mpc.oa = mpc.odelta = None
    mpc.target_ind = 340
    mpc.set_trajectory_fromarray(r[:n], cutoff_idx=-40)
    mpc.step(st2)
    ref2 = oracle.mpc_step(p, (st2.x, st2.y, st2.yaw, st2.v), r[:n, 0], r[:n, 1], r[:n, 2], 340, 30 / 3.6, cv=cv1)
    assert np.array_equal(mpc.xref, ref2["xref"]) and (mpc.xref[2] == 0).any() and (mpc.xref[2] > 0).any()
    # 4. the goal test stops at this module's 0.5 / 3.6, not lib.mpc's 0.1389
    mpc.target_ind = len(mpc.cx) - 2
    g = pkg.State(x=mpc.goal[0], y=mpc.goal[1], yaw=0.0, v=0.1389)
    assert m.STOP_SPEED < 0.1389 and not mpc.is_goal(g)
    assert mpc.is_goal(pkg.State(x=mpc.goal[0], y=mpc.goal[1], yaw=0.0, v=0.13))
