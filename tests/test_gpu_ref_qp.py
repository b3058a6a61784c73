"""The HIP path against the REFERENCE's own whole step (tests/golden/ref_qp_T*.npz): inputs -> what `MPC.step` of
main/lib/mpc.py produced, unmodified, under the recording cvxpy stand-in (tests/golden/make_golden_refqp.py; the QP it emitted
solved to its unique optimum by the stand-in's interior-point + KKT-polish solver).  Nothing under oracle/ is imported here.

Bars: status / target_ind / xref bit-exact; u* <= 1e-4 abs (north_star), observed <= 1e-8; predicted states <= 1e-6; the
condensed gradient and Hessian the kernel built == the emitted problem with x eliminated generically, <= 1e-9 relative; active
rows identical to the rows of the emitted problem with a positive multiplier wherever those are unique (independent tight rows),
and a subset of the tight rows in the few degenerate cases; (di, ai) as the reference's S5 lines return them, MAX_DECEL and the
kept di on its "Cannot solve mpc" path.  Unpinned: ECOS's stopping tolerance around this optimum."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gpu_helpers import debug_bufs
import qp_sparse_numpy as QS
import refqp_tools as RT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T", (13, 20, 30, 40))
def test_hip_step_against_reference_step(pkg, routes, T):
    g = load_golden(f"ref_qp_T{T}.npz")
    B = len(g["x0"])
    eng = pkg.BatchedMPC(routes, g["path_id"], dl=pkg.synth.DL, T=T, speed=g["speed"], device="cuda:0", smooth=False)
    eng.load_state(g["target_ind_in"], g["oa_in"], g["od_in"], g["path_len"])
    dbg = debug_bufs(eng)
    eng.solve(torch.from_numpy(np.ascontiguousarray(g["x0"])).to(eng.device), debug=dbg)
    torch.cuda.synchronize()
    st = eng.status.cpu().numpy()
    assert np.array_equal(st, g["status"]) and (st == 1).sum() == 2
    assert np.array_equal(eng.target_ind.cpu().numpy(), g["target_ind_out"])
    assert np.array_equal(eng.xref.cpu().numpy(), g["xref"])
    ok = st == 0
    oa, od = eng.oa.cpu().numpy(), eng.od.cpu().numpy()
    du = max(np.abs(oa - g["oa"])[ok].max(), np.abs(od - g["od"])[ok].max())
    assert du <= 1e-4, du
    for name in ("ox", "oy", "ov", "oyaw"):
        np.testing.assert_allclose(getattr(eng, name).cpu().numpy()[ok], g[name][ok], rtol=0, atol=1e-6)
    bits = RT.active_bits(eng.active_mask.cpu().numpy(), 8 * T)
    nd = ok & ~g["degenerate"]
    assert nd.sum() >= 140 and np.array_equal(bits[nd], g["active"][nd])
    assert np.all(g["tight"][ok] | ~bits[ok])
    lam = dbg["lam"].cpu().numpy()
    gs = np.maximum(1.0, np.abs(dbg["g"].cpu().numpy()).max(axis=1))
    assert (np.abs(lam - g["lam"]).max(axis=1)[nd] / gs[nd]).max() <= 1e-6
    # the condensed QP the kernel built vs the reference's emitted problem (x eliminated by a generic dense solve), a sample
    Hk, gk = dbg["H"].cpu().numpy(), dbg["g"].cpu().numpy()
    for i in list(range(0, B, 7)) + [B - 1]:
        if not ok[i]:
            continue                                  # an infeasible start leaves the kernel before the QP is built
        P, q, c0, A, b, G, h = RT.emitted_problem(g, i)
        Hr, gr, Gr, hr, Phi, phi = QS.condense(P, q, A, b, G, h, T)
        assert np.abs(np.tril(Hk[i]) - np.tril(Hr)).max() <= 1e-9 * np.abs(Hr).max()
        assert np.abs(gk[i] - gr).max() <= 1e-9 * max(1.0, np.abs(gr).max())
    print(f"T={T}: {B} reference steps, max|du|={du:.2e}")
    assert du <= 1e-7


def test_single_ego_drop_in_returns_what_the_reference_returns(pkg, routes):
    """The drop-in `MPC.step` (the surface scenario scripts call) on fixture cases of the stock horizon: (di, ai) and the
    attributes the reference sets, incl. its failure path (stderr text, ai = MAX_DECEL, di kept)."""
    T = 13
    g = load_golden(f"ref_qp_T{T}.npz")
    car = pkg.vehicle.BicycleModelDimensions()
    picks = list(range(0, 40, 5)) + [int(i) for i in np.flatnonzero(g["status"] == 1)]
    for i in picks:
        route = routes[int(g["path_id"][i])]
        full = route.copy()
        mpc = pkg.mpc.MPC(full[:, 0], full[:, 1], full[:, 2], pkg.synth.DL, car, speed=float(g["speed"][i]), dt=0.2)
        mpc.set_trajectory_fromarray(full[:int(g["path_len"][i])])
        mpc.target_ind = int(g["target_ind_in"][i])
        mpc.oa, mpc.odelta = g["oa_in"][i].copy(), g["od_in"][i].copy()
        mpc.di = 0.123
        x0 = g["x0"][i]
        di, ai = mpc.step(pkg.vehicle.State(x=x0[0], y=x0[1], yaw=x0[3], v=x0[2]))
        assert mpc.target_ind == g["target_ind_out"][i]
        if g["status"][i] == 0:
            assert abs(di - g["di"][i]) <= 1e-7 and abs(ai - g["ai"][i]) <= 1e-7
            assert np.abs(np.asarray(mpc.ox) - g["ox"][i]).max() <= 1e-6 and np.array_equal(np.asarray(mpc.xref), g["xref"][i])
        else:
            assert ai == g["ai"][i] == -10.0 and di == 0.123 and mpc.odelta is None and mpc.ox is None
