#!/usr/bin/env python3
"""Dev: the SAME 256 egos (config 2's batch) tiled x1, x2, x4, x8, x16 -- identical work per ego, so the launch time
tells how egos that share a CU slow each other down (no straggler effect: every copy takes the same iterations).
usage: tile_scaling.py [T=20] [ticks=100]"""
import importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
pkg = importlib.import_module("av-simulation-at-intersections_amd")
WL, S = pkg.workloads, pkg.synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
routes, _ = WL.route_table(T == 40, source="planner")
base = WL.ego_batch(routes, 256, T, rank=0)
for rep in (1, 2, 3, 4, 6, 8, 16):
    b = S.EgoBatch(**{k: np.concatenate([getattr(base, k)] * rep) for k in ("x0", "path_id", "path_len", "target_ind", "speed", "oa", "od")})
    eng, x0 = WL.make_engine(routes, b, T, "cuda:0")
    pkg._cabi.check(eng.lib.jsim_mpc_set_launch_order(eng._ctx, 0), eng._ctx)
    loop = pkg.ClosedLoop(eng, x0, hist_cap=K + 16, max_age=400)
    loop.run(5); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        eng.load_state(b.target_ind, b.oa, b.od, b.path_len); loop.x0.copy_(torch.from_numpy(b.x0).to(eng.device)); loop.age.zero_()
        loop.tick_counter.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter(); loop.run(K); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(f"T={T} {256 * rep:5d} egos ({rep:2d} x 256): {t * 1e3:8.3f} ms per {K} ticks, {256 * rep * K / t / 1e6:7.3f} M steps/s, {t / K * 1e6:7.2f} us per tick", flush=True)
    eng.close()
