"""The HIP path against the committed G3 fixtures (tests/golden/qp_T*.npz, made by tests/golden/make_golden_qp.py) -- whole
MPC steps with every constraint family active, the two infeasible starts and the coincident-rows start -- WITHOUT the oracle:
nothing under oracle/ is imported or loaded here.  The fixtures are the oracle's solutions cross-checked by scipy
trust-constr / KKT at generation time; against ECOS (the reference's solver, absent offline) they stay parity unpinned.

Bars: status / target_ind bit-exact; u* <= 1e-4 abs (north_star; observed ~1e-9); active-constraint indices identical;
multipliers <= 1e-6 relative; g <= 1e-9 relative; H (the stored cases) <= 1e-9 relative."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden
from gpu_helpers import debug_bufs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T", (13, 20, 30, 40))
def test_hip_step_against_golden_qp(pkg, routes, T):
    assert "oracle_py" not in sys.modules or True   # (other test modules of the session may have loaded it; this one never does)
    g = load_golden(f"qp_T{T}.npz")
    B = len(g["x0"])
    assert B >= 50
    eng = pkg.BatchedMPC(routes, g["path_id"], dl=pkg.synth.DL, T=T, speed=g["speed"], device="cuda:0", smooth=False)
    eng.load_state(g["target_ind_in"], g["oa_in"], g["od_in"], g["path_len"])
    dbg = debug_bufs(eng)
    eng.solve(torch.from_numpy(np.ascontiguousarray(g["x0"])).to(eng.device), debug=dbg)
    torch.cuda.synchronize()
    st = eng.status.cpu().numpy()
    assert np.array_equal(st, g["status"])
    assert (st == 1).sum() >= 2                                    # v0 > speed and v0 < MIN_SPEED: the reference's failure path
    assert np.array_equal(eng.target_ind.cpu().numpy(), g["target_ind_out"])
    ok = st == 0
    oa, od = eng.oa.cpu().numpy(), eng.od.cpu().numpy()
    du = max(np.abs(oa - g["oa"])[ok].max(), np.abs(od - g["od"])[ok].max())
    assert du <= 1e-4, du
    assert np.all(oa[~ok] == 0) and np.all(od[~ok] == 0)
    assert np.array_equal(eng.active_mask.cpu().numpy().view(np.uint32)[ok], g["active_mask"][ok])
    gg = dbg["g"].cpu().numpy()
    gs = np.maximum(1.0, np.abs(g["g"]).max(axis=1))
    assert (np.abs(gg - g["g"]).max(axis=1)[ok] / gs[ok]).max() <= 1e-9
    lam = dbg["lam"].cpu().numpy()
    assert (np.abs(lam - g["lam"]).max(axis=1)[ok] / gs[ok]).max() <= 1e-6
    H = dbg["H"].cpu().numpy()
    for k, b in enumerate(g["H_idx"]):
        ref = g["H"][k]
        assert np.abs(np.tril(H[b]) - np.tril(ref)).max() <= 1e-9 * np.abs(ref).max()
    for name in ("ox", "oy", "ov", "oyaw"):
        np.testing.assert_allclose(getattr(eng, name).cpu().numpy()[ok], g[name][ok], rtol=0, atol=1e-6)
    same = (eng.n_iter.cpu().numpy()[ok] == g["n_iter"][ok]).mean()
    # every constraint family is active somewhere in the fixture (checked on the HIP result, not only on the file)
    words = eng.active_mask.cpu().numpy().view(np.uint32)
    bits = ((words[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(B, -1)[:, :8 * T].astype(bool)
    fam = {"D": bits[:, :2 * T - 2].any(), "VU": bits[:, 2 * T - 2:3 * T - 1].any(), "VL": bits[:, 3 * T - 1:4 * T].any(),
           "AU": bits[:, 4 * T:5 * T].any(), "AL": bits[:, 5 * T:6 * T].any(), "S": bits[:, 6 * T:].any()}
    assert all(fam.values()), fam
    print(f"T={T}: {B} golden cases, max|du|={du:.2e}, identical iteration counts {same * 100:.1f}%")
    assert du <= 1e-7 and same >= 0.9
