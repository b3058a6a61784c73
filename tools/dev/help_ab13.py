#!/usr/bin/env python3
"""Dev: the closed loop of 256 egos at horizon T (planned routes) with the kernel the environment selects (JSIM_HELP_MAX_B=0: no
helper wavefronts): launch time and a checksum.  usage: help_ab13.py [T=13] [ticks=100]"""
import hashlib, importlib, os, sys, time
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
pkg = importlib.import_module("av-simulation-at-intersections_amd")
WL = pkg.workloads
T = int(sys.argv[1]) if len(sys.argv) > 1 else 13
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
routes, _ = WL.route_table(False, source="planner")
b = WL.ego_batch(routes, 256, T, rank=0)
eng, x0 = WL.make_engine(routes, b, T, "cuda:0")
loop = pkg.ClosedLoop(eng, x0, hist_cap=K + 16, max_age=400)
loop.run(5); torch.cuda.synchronize()
ts = []
for _ in range(5):
    eng.load_state(b.target_ind, b.oa, b.od, b.path_len); loop.x0.copy_(torch.from_numpy(b.x0).to(eng.device)); loop.age.zero_()
    loop.tick_counter.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter(); loop.run(K); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
h = hashlib.sha256(loop.hist[:K].cpu().numpy().tobytes() + loop.x0.cpu().numpy().tobytes() + eng.target_ind.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"T={T} JSIM_HELP_MAX_B={os.environ.get('JSIM_HELP_MAX_B', 'default')}: {min(ts) * 1e3:.3f} ms per {K} ticks ({256 * K / min(ts) / 1e6:.3f} M steps/s), sha {h}")
