"""LIVE: the reference's own `MPC.step` (main/lib/mpc.py, imported from /root/reference under the recording cvxpy stand-in)
on FRESH seeded egos, against the oracle -- the same comparison tests/golden/make_golden_refqp.py makes before it writes the
fixtures, on cases no fixture holds.  Runs only where /root/reference exists (the build container); the GPU box has no
reference and the committed fixtures (tests/test_ref_qp_cpu.py, tests/test_gpu_ref_qp.py) stand in for it there."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN

REF_MAIN = "/root/reference/main"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_MAIN), reason="reference tree not present on this machine")


@pytest.mark.parametrize("T,n", ((13, 120), (20, 100), (30, 70), (40, 50)))
def test_reference_step_live_vs_oracle(oracle, pkg, routes, T, n):
    sys.path.insert(0, GOLDEN)
    import make_golden_refqp as M
    import qp_sparse_numpy as QS
    saved = {k: sys.modules.get(k) for k in ("cvxpy",)}
    try:
        refmpc, rec = M.import_reference(T)
        from lib.car_dimensions import BicycleModelDimensions
        from lib.simulation import State
        car = BicycleModelDimensions()
        b = pkg.synth.make_ego_batch(routes, n, T, seed=9000 + T, truncate=True, near_end_frac=0.25)
        figs = []
        for i in range(n):
            c = (b.x0[i].copy(), int(b.path_id[i]), int(b.path_len[i]), int(b.target_ind[i]), float(b.speed[i]), b.oa[i].copy(),
                 b.od[i].copy(), "live")
            r, out = M.reference_step(refmpc, rec, car, State, routes[c[1]], c)
            figs.append(M.compare_with_oracle(oracle, QS, pkg, T, routes[c[1]], c, r, out))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        # leave no trace of the reference's packages: other tests import the repo's own `lib` shim under the same name
        for k in [k for k in sys.modules if k == "lib" or k.startswith("lib.") or k.split(".")[0] in ("bicycle", "envs")]:
            if getattr(sys.modules[k], "__file__", "") and "/root/reference" in (sys.modules[k].__file__ or ""):
                del sys.modules[k]
        while REF_MAIN in sys.path:
            sys.path.remove(REF_MAIN)
    mx = lambda k: max(f.get(k, 0.0) for f in figs)
    assert all(f["status_equal"] and f["target_equal"] and f["xref_equal"] for f in figs)
    assert mx("dH") <= 1e-12 and mx("dG") <= 1e-12 and mx("dh") <= 1e-12 and mx("dg") <= 1e-11
    assert mx("du") <= 1e-8 and mx("dx") <= 1e-8
    assert all(f.get("active_in_tight", True) for f in figs)
    assert all(f.get("active_equal", True) for f in figs if not f.get("degenerate"))
    assert sum("du" in f for f in figs) >= n - 2
