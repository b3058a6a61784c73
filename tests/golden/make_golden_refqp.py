#!/usr/bin/env python3
"""Fixtures of stage S4 AS THE REFERENCE ASSEMBLES IT: tests/golden/ref_qp_T{13,20,30,40}.npz + ref_qp_sweep.json.

Run in the build container only (needs /root/reference):    python tests/golden/make_golden_refqp.py [T ...]

What runs.  The reference's `lib.mpc` is imported from /root/reference/main with tests/golden/cvxpy_recorder.py registered as
`cvxpy` (the real cvxpy / ECOS are not installed, pinned or fetchable: SURVEY.md D5), and for every case the reference's own
`MPC.step(state)` (main/lib/mpc.py:284-303) is called UNMODIFIED: `_calc_ref_trajectory`, `_predict_motion`,
`_linear_mpc_control` (:141-211 -- the QP assembly that had never executed before round 3) and the S5 lines that unpack
`x.value` / `u.value`.  The horizon is a module constant the reference derives from its JSON at import (`T`, and `Qf = diag(...) * T`,
:22,:28): the module is re-imported per horizon with `json.load` patched to return the stock config with "T" replaced, so
the reference's own module-level lines compute every derived constant.

What is stored per case (data only: inputs, emitted matrices, outputs):
  inputs   x0 [x, y, v, yaw], path_id / path_len on the synthetic route table (synth.make_route_table, yaw smoothed),
           target_ind_in, speed, warm start oa_in / od_in;
  emitted  the reference's problem  min z'Pz + q'z + c0  s.t.  A z = b,  G z <= h  in COO form (exact zeros dropped), rows in the
           order of its `constraints` list, z = [x(:,0), ..., x(:,T), u(:,0), ..., u(:,T-1)]; `eq_src` / `in_src` = position in
           that list of the constraint each row came from (identical for all cases of a horizon, stored once);
  outputs  what the reference's S5 lines produced from the stand-in's optimum: status (0 solved / 1 "Cannot solve mpc"),
           oa, odelta, ox, oy, ov, oyaw, xref, target_ind, (di, ai); the optimum's multipliers `lam`, `active` (multiplier > 1e-9
           max(1, |g|_inf); canonical row id = emitted row id), `tight` (rows holding with equality) and `degenerate` (the tight rows'
           normals are linearly dependent: multipliers not unique, only `active subset of tight` is well defined); the stand-in's
           KKT residuals.
Cases: the 54 G3 cases of make_golden_qp.py (every constraint family active, the two infeasible starts, the coincident-rows
start) + 96 more seeded random egos (truncated paths, 30 % near the path end).

Checked here, before anything is written (the same checks are tests: tests/test_ref_qp_cpu.py on the fixtures,
tests/test_ref_qp_live.py on fresh cases when /root/reference is present):
  * the oracle's condensed (H, g, G, h) equal the emitted problem after GENERIC elimination of x through A z = b
    (tests/qp_sparse_numpy.condense) to <= 1e-12 relative (g: 1e-11), row for row in the emitted order;
  * the oracle's status, target_ind, xref equal the reference's bit for bit; u*, predicted states <= 1e-8; active sets equal.
`ref_qp_sweep.json` holds the same comparison over 1000 more random egos per horizon (summary only).

STILL UNPINNED after this: what ECOS itself would return for the emitted problem (stopping tolerance 1e-8, OPTIMAL_INACCURATE
accepted).  The optimum is unique, so that is a tolerance band around these numbers, not a different answer.
"""
import contextlib
import importlib
import io
import json
import os
import sys
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_MAIN = "/root/reference/main"
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

TS = (13, 20, 30, 40)
N_EXTRA = 96
N_SWEEP = 1000
DL, DT = 0.083, 0.2


def import_reference(T):
    """`lib.mpc` of the reference for horizon T, its own module-level code run on the stock config with "T" replaced."""
    import cvxpy_recorder
    sys.modules["cvxpy"] = cvxpy_recorder
    if REF_MAIN not in sys.path:
        sys.path.insert(0, REF_MAIN)
    import matplotlib
    matplotlib.use("Agg")
    real_load = json.load

    def load(f):
        cfg = real_load(f)
        cfg["T"] = T
        return cfg

    with mock.patch.object(json, "load", load):
        import lib.mpc as refmpc
        refmpc = importlib.reload(refmpc)
    assert refmpc.T == T and refmpc.cvxpy is cvxpy_recorder
    return refmpc, cvxpy_recorder


def reference_step(refmpc, rec, car, State, route, case):
    """One unmodified MPC.step of the reference on `case`; returns (emitted problem record, outputs dict)."""
    x0, rid, plen, tind, speed, oa, od = case[:7]
    full = route.copy()
    mpc = refmpc.MPC(full[:, 0], full[:, 1], full[:, 2], DL, car, speed=speed, dt=DT)
    mpc.set_trajectory_fromarray(full[:plen])                    # the loop's truncated path (scenarios/mpc_intersection.py:143)
    mpc.target_ind = int(tind)
    mpc.oa, mpc.odelta = np.array(oa, dtype=np.float64), np.array(od, dtype=np.float64)
    di_before = 0.123                                            # a failed solve keeps the previous di (:298-301)
    mpc.di = di_before
    del rec.RECORDS[:]
    err = io.StringIO()
    with contextlib.redirect_stderr(err):
        di, ai = mpc.step(State(x=x0[0], y=x0[1], yaw=x0[3], v=x0[2]))
    assert len(rec.RECORDS) == 1
    r = rec.RECORDS[0]
    T = refmpc.T
    ok = mpc.odelta is not None
    assert ok == (r["status"] == rec.OPTIMAL) and (ok or "Cannot solve mpc" in err.getvalue())
    z = lambda n: np.zeros(n)
    out = dict(status=0 if ok else 1, di=float(di), ai=float(ai), target_ind=int(mpc.target_ind), xref=np.array(mpc.xref),
               oa=np.array(mpc.oa) if ok else z(T), od=np.array(mpc.odelta) if ok else z(T),
               ox=np.array(mpc.ox) if ok else z(T + 1), oy=np.array(mpc.oy) if ok else z(T + 1),
               ov=np.array(mpc.ov) if ok else z(T + 1), oyaw=np.array(mpc.oyaw) if ok else z(T + 1),
               lam=r["lam"] if ok else z(8 * T), di_kept=(not ok and di == di_before))
    return r, out


def compare_with_oracle(O, QS, pkg, T, route, case, r, out):
    """The oracle on the same inputs against the reference's emitted problem and outputs.  Returns a dict of error figures."""
    x0, rid, plen, tind, speed, oa, od = case[:7]
    p = O.make_params(T=T)
    rt = route[:plen]
    res = O.mpc_step(p, (x0[0], x0[1], x0[3], x0[2]), rt[:, 0], rt[:, 1], rt[:, 2], tind, speed, oa=oa, od=od, want_qp=True)
    fig = dict(status_equal=res["status"] == out["status"], target_equal=res["target_ind"] == out["target_ind"],
               xref_equal=bool(np.array_equal(res["xref"], out["xref"])))
    # the emitted problem, x eliminated generically, against the oracle's staged build
    st_, xref, idx, rend, t2 = O.calc_ref_trajectory(p, x0[0], x0[1], x0[2], rt[:, 0], rt[:, 1], rt[:, 2], tind)
    xbar = O.predict_motion(p, x0, oa, od)
    st3, H, g, G, h, skip, fresp, Sens = O.build_qp(p, xref, xbar, x0, rend, speed)
    Hr, gr, Gr, hr, Phi, phi = QS.condense(r["P"], r["q"], r["A"], r["b"], r["G"], r["h"], T)
    sH, sg = np.abs(Hr).max(), max(1.0, np.abs(gr).max())
    keep = ~skip.astype(bool)
    fig.update(dH=float(np.abs(H - Hr).max() / sH), dg=float(np.abs(g - gr).max() / sg),
               dG=float(np.abs(G[keep] - Gr[keep]).max()), dh=float(np.abs(h[keep] - hr[keep]).max()),
               skip_rows_constant=bool(np.abs(Gr[~keep]).max(initial=0.0) <= 1e-13), n_skip=int((~keep).sum()))
    if out["status"] == 0 and res["status"] == 0:
        du = max(np.abs(res["oa"] - out["oa"]).max(), np.abs(res["od"] - out["od"]).max())
        dx = max(np.abs(res[k] - out[k]).max() for k in ("ox", "oy", "ov", "oyaw"))
        thr = 1e-9 * max(1.0, np.abs(gr).max())
        act_ref = sorted(int(i) for i in np.flatnonzero(out["lam"] > thr))
        # rows that hold with equality at the optimum; if their normals are linearly dependent the multipliers -- and with them
        # "the rows with a positive multiplier" -- are not unique (e.g. delta_1 = delta_3 = -MAX_STEER with both steer-rate rows
        # around delta_2 tight: four tight rows on three variables), and only `active subset of tight` can be asked of anyone
        u_ref = np.empty(2 * T); u_ref[0::2] = out["oa"]; u_ref[1::2] = out["od"]
        tight = keep & (hr - Gr @ u_ref <= 1e-9 * np.maximum(1.0, np.abs(hr)))
        degenerate = bool(np.linalg.matrix_rank(Gr[tight], tol=1e-9) < int(tight.sum())) if tight.any() else False
        out["tight"], out["degenerate"] = tight, degenerate
        out["active"] = out["lam"] > thr                      # the rule of oracle and kernels: multiplier > 1e-9 max(1, |g|_inf)
        fig.update(du=float(du), dx=float(dx), active_equal=act_ref == sorted(res["active"]), n_active=len(act_ref),
                   degenerate=degenerate, active_in_tight=bool(tight[res["active"]].all()),
                   dlam=float(np.abs(res["lam"] - out["lam"]).max() / sg))
    return fig


def coo(M):
    i, j = np.nonzero(M)
    return i.astype(np.int16), j.astype(np.int16), M[i, j]


def main():
    if not os.path.isdir(REF_MAIN):
        raise SystemExit("reference not present; these fixtures can only be generated in the build container")
    import oracle_py as O
    import qp_sparse_numpy as QS
    import make_golden_qp as G3
    O.build()
    pkg = importlib.import_module("av-simulation-at-intersections_amd")
    routes = pkg.synth.make_route_table()
    for r_ in routes:
        pkg.synth.smooth_yaw_inplace(r_[:, 2])
    sweep = {}
    for T in (tuple(int(a) for a in sys.argv[1:]) or TS):
        refmpc, rec = import_reference(T)
        from lib.car_dimensions import BicycleModelDimensions
        from lib.simulation import State
        car = BicycleModelDimensions()
        cases = G3.build_cases(pkg, routes, T)
        b = pkg.synth.make_ego_batch(routes, N_EXTRA, T, seed=300 + T, truncate=True, near_end_frac=0.3)
        cases += [(b.x0[i].copy(), int(b.path_id[i]), int(b.path_len[i]), int(b.target_ind[i]), float(b.speed[i]), b.oa[i].copy(),
                   b.od[i].copy(), "random (extra)") for i in range(N_EXTRA)]
        recs, outs, figs = [], [], []
        for c in cases:
            r, out = reference_step(refmpc, rec, car, State, routes[c[1]], c)
            f = compare_with_oracle(O, QS, pkg, T, routes[c[1]], c, r, out)
            assert f["status_equal"] and f["target_equal"] and f["xref_equal"], (T, c[7], f)
            # (g: 1e-11 -- S'Q(f - xref) cancels between its terms and the generic elimination is a dense solve; seen <= 5e-12 at T = 40)
            assert max(f["dH"], f["dG"], f["dh"]) <= 1e-12 and f["dg"] <= 1e-11 and f["skip_rows_constant"] and f["n_skip"] == 2, (T, c[7], f)
            if out["status"] == 0:
                assert f["du"] <= 1e-8 and f["dx"] <= 1e-8, (T, c[7], f)
                assert f["active_in_tight"] and (f["active_equal"] or f["degenerate"]), (T, c[7], f)
                assert f["degenerate"] or "coincident" not in c[7]
            else:
                out["tight"], out["degenerate"], out["active"] = np.zeros(8 * T, dtype=bool), False, np.zeros(8 * T, dtype=bool)
            recs.append(r); outs.append(out); figs.append(f)
        n = len(cases)
        src_eq, src_in = recs[0]["eq_src"], recs[0]["in_src"]
        assert all(np.array_equal(r["eq_src"], src_eq) and np.array_equal(r["in_src"], src_in) for r in recs)
        parts = {k: [coo(r[k]) for r in recs] for k in ("P", "A", "G")}
        save = {}
        for k, lst in parts.items():
            save[f"{k}_ptr"] = np.cumsum([0] + [len(t[2]) for t in lst]).astype(np.int64)
            save[f"{k}_i"] = np.concatenate([t[0] for t in lst]); save[f"{k}_j"] = np.concatenate([t[1] for t in lst])
            save[f"{k}_v"] = np.concatenate([t[2] for t in lst])
        solved = np.array([o["status"] == 0 for o in outs])
        kk = lambda name: np.array([r["kkt"][name] if r["kkt"] else 0.0 for r in recs])
        np.savez_compressed(
            os.path.join(HERE, f"ref_qp_T{T}.npz"),
            x0=np.array([c[0] for c in cases]), path_id=np.array([c[1] for c in cases], dtype=np.int32),
            path_len=np.array([c[2] for c in cases], dtype=np.int32), target_ind_in=np.array([c[3] for c in cases], dtype=np.int64),
            speed=np.array([c[4] for c in cases]), oa_in=np.array([c[5] for c in cases]), od_in=np.array([c[6] for c in cases]),
            degenerate=np.array([o["degenerate"] for o in outs]), tight=np.array([o["tight"] for o in outs]),
            active=np.array([o["active"] for o in outs]),
            n_z=np.int64(recs[0]["n"]), eq_src=src_eq, in_src=src_in, q=np.array([r["q"] for r in recs]), c0=np.array([r["c0"] for r in recs]),
            b=np.array([r["b"] for r in recs]), h=np.array([r["h"] for r in recs]), **save,
            status=np.array([o["status"] for o in outs], dtype=np.int32), di=np.array([o["di"] for o in outs]),
            ai=np.array([o["ai"] for o in outs]), di_kept=np.array([o["di_kept"] for o in outs]),
            target_ind_out=np.array([o["target_ind"] for o in outs], dtype=np.int64), xref=np.array([o["xref"] for o in outs]),
            oa=np.array([o["oa"] for o in outs]), od=np.array([o["od"] for o in outs]), ox=np.array([o["ox"] for o in outs]),
            oy=np.array([o["oy"] for o in outs]), ov=np.array([o["ov"] for o in outs]), oyaw=np.array([o["oyaw"] for o in outs]),
            lam=np.array([o["lam"] for o in outs]),
            kkt_stationarity=kk("stationarity"), kkt_primal_eq=kk("primal_eq"), kkt_primal_in=kk("primal_in"), kkt_dual_min=kk("dual_min"),
            du_ipm_vs_polish=kk("du_ipm"))
        mx = lambda name: max(f.get(name, 0.0) for f in figs)
        print(f"T={T}: {n} cases ({int(solved.sum())} solved, {int((~solved).sum())} infeasible); oracle vs the reference's emitted problem: "
              f"dH {mx('dH'):.1e} dg {mx('dg'):.1e} dG {mx('dG'):.1e} dh {mx('dh'):.1e}; vs its outputs: du {mx('du'):.1e} dx {mx('dx'):.1e} "
              f"dlam(non-degenerate) {max(f.get('dlam', 0.0) for f in figs if not f.get('degenerate')):.1e}; "
              f"{sum(bool(f.get('degenerate')) for f in figs)} degenerate cases (dependent tight rows), active sets equal on "
              f"{sum(f.get('active_equal', True) for f in figs)}/{n}, on every non-degenerate one")
        # ---- the sweep: 1000 more egos, compared and summarised, not stored
        b = pkg.synth.make_ego_batch(routes, N_SWEEP, T, seed=700 + T, truncate=True, near_end_frac=0.2)
        sf = []
        for i in range(N_SWEEP):
            c = (b.x0[i].copy(), int(b.path_id[i]), int(b.path_len[i]), int(b.target_ind[i]), float(b.speed[i]), b.oa[i].copy(),
                 b.od[i].copy(), "sweep")
            r, out = reference_step(refmpc, rec, car, State, routes[c[1]], c)
            sf.append(compare_with_oracle(O, QS, pkg, T, routes[c[1]], c, r, out))
        smx = lambda name: float(max(f.get(name, 0.0) for f in sf))
        nd = [f for f in sf if not f.get("degenerate")]
        sweep[str(T)] = dict(egos=N_SWEEP, seed=700 + T, solved=int(sum("du" in f for f in sf)),
                             status_equal=int(sum(f["status_equal"] for f in sf)), target_equal=int(sum(f["target_equal"] for f in sf)),
                             xref_equal=int(sum(f["xref_equal"] for f in sf)), degenerate=N_SWEEP - len(nd),
                             active_equal_nondegenerate=int(sum(f.get("active_equal", True) for f in nd)),
                             active_in_tight=int(sum(f.get("active_in_tight", True) for f in sf)),
                             active_equal=int(sum(f.get("active_equal", True) for f in sf)),
                             max_dH=smx("dH"), max_dg=smx("dg"), max_dG=smx("dG"), max_dh=smx("dh"), max_du=smx("du"), max_dx=smx("dx"),
                             max_dlam_nondegenerate=float(max(f.get("dlam", 0.0) for f in nd)), mean_active=float(np.mean([f["n_active"] for f in sf if "n_active" in f])))
        s = sweep[str(T)]
        assert s["status_equal"] == s["target_equal"] == s["xref_equal"] == s["active_in_tight"] == N_SWEEP, s
        assert s["active_equal_nondegenerate"] == len(nd), s
        assert max(s["max_dH"], s["max_dG"], s["max_dh"]) <= 1e-12 and s["max_dg"] <= 1e-11 and s["max_du"] <= 1e-8, s
        print(f"   sweep of {N_SWEEP}: {s}")
    path = os.path.join(HERE, "ref_qp_sweep.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    old.update(sweep)
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
