"""GPU tests at the per-GPU shares of BASELINE.json's multi-GPU configurations and of the N > 1 path itself:

  * config 4 (32768 egos / 8 GPUs, T = 20): one rank's 4096 egos -- KKT property on all of them, a 512-ego slice against the
    oracle;
  * config 5 (8192 egos / 8 GPUs, T = 40, multi-lane geometry): one rank's 1024 egos on the 48 two-lane routes -- every ego
    against the oracle, KKT on all;
  * shard -> solve -> gather == the unsharded job: two processes sharing the one GPU of the box, gloo for the gather
    (RCCL needs one GPU per rank), the HIP path doing the solves;
  * `python bench.py --gpus 2` started plainly: spawns its two ranks itself (rehearsal mode on the one GPU), exits 0 and
    prints one JSON line with n_gpus = 2;
  * the fused closed loop (jsim_mpc_run_ticks) against the oracle's closed loop (orc_closed_loop) tick by tick.

Run with -m gpu on an MI355X."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO
from gpu_helpers import debug_bufs, engine, kkt_check, oracle_batch

pytestmark = pytest.mark.gpu
U_TOL = 1e-4   # north_star tolerance on u*


def _smoothed(pkg, multi_lane):
    rs = pkg.synth.make_route_table(multi_lane=multi_lane)
    for r in rs:
        pkg.synth.smooth_yaw_inplace(r[:, 2])
    return rs


def _compare(eng, ref, sl=slice(None)):
    st = eng.status.cpu().numpy()[sl]
    assert np.array_equal(st, ref["status"])
    assert np.array_equal(eng.target_ind.cpu().numpy()[sl], ref["target_ind"])
    np.testing.assert_array_equal(eng.xref.cpu().numpy()[sl], ref["xref"])
    ok = st == 0
    err = max(np.abs(eng.oa.cpu().numpy()[sl] - ref["oa"])[ok].max(), np.abs(eng.od.cpu().numpy()[sl] - ref["od"])[ok].max())
    assert err <= U_TOL, err
    assert np.array_equal(eng.active_mask.cpu().numpy().view(np.uint32)[sl], ref["active_mask"])   # bit-exact active sets
    return err, ok


def test_config4_share_4096_egos_T20(pkg, oracle, routes):
    B, T = 4096, 20
    batch = pkg.synth.make_ego_batch(routes, B, T, seed=4, truncate=True, near_end_frac=0.15)
    eng = engine(pkg, routes, batch, T)
    dbg = debug_bufs(eng)
    eng.solve(torch.from_numpy(batch.x0).to(eng.device), debug=dbg)
    torch.cuda.synchronize()
    assert int((eng.status == 0).sum()) >= B - 4
    kkt_check(eng, batch, dbg)
    S = pkg.synth
    sub = S.EgoBatch(**{k: getattr(batch, k)[1024:1536] for k in ("x0", "path_id", "path_len", "target_ind", "speed", "oa", "od")})
    _, ref = oracle_batch(oracle, pkg, routes, sub, T, n_threads=8)
    err, ok = _compare(eng, ref, slice(1024, 1536))
    same = (eng.n_iter.cpu().numpy()[1024:1536] == ref["n_iter"]).mean()
    print(f"config 4 share: 4096x20 KKT ok; 512-ego oracle slice max|du|={err:.2e}, identical iteration counts {same * 100:.1f}%")
    assert err <= 1e-7 and same >= 0.9


def test_config5_share_1024_egos_T40_multi_lane(pkg, oracle):
    B, T = 1024, 40
    routes = _smoothed(pkg, multi_lane=True)
    batch = pkg.synth.make_ego_batch(routes, B, T, seed=5, truncate=True, near_end_frac=0.15)
    assert len(np.unique(batch.path_id)) >= 40          # the 48 two-lane routes are actually used
    eng = engine(pkg, routes, batch, T)
    dbg = debug_bufs(eng)
    eng.solve(torch.from_numpy(batch.x0).to(eng.device), debug=dbg)
    torch.cuda.synchronize()
    assert int((eng.status == 0).sum()) >= B - 4
    kkt_check(eng, batch, dbg)
    _, ref = oracle_batch(oracle, pkg, routes, batch, T, n_threads=8)
    err, ok = _compare(eng, ref)
    same = (eng.n_iter.cpu().numpy() == ref["n_iter"]).mean()
    print(f"config 5 share: 1024x40 multi-lane max|du|={err:.2e}, identical iteration counts {same * 100:.1f}%, "
          f"mean n_iter {ref['n_iter'][ok].mean():.1f}")
    assert err <= 1e-6 and same >= 0.9
    # and the fused closed loop on these routes keeps running: 20 ticks, egos stay on their routes
    loop = pkg.ClosedLoop(eng, torch.from_numpy(batch.x0).to(eng.device), hist_cap=20, max_age=400)
    loop.run(20)
    torch.cuda.synchronize()
    assert int((eng.status == 0).sum()) >= B - 8
    dev, _ = eng.xref_deviation_and_goal(loop.x0)
    assert float(dev[eng.status == 0].median()) < 1.0


@pytest.mark.parametrize("T", (20, 40))
def test_fused_closed_loop_vs_oracle_closed_loop(pkg, oracle, routes, T):
    """jsim_mpc_run_ticks against the oracle's closed loop on the same egos.  Two free-running loops drift apart by the
    loop's own sensitivity (a 1e-9 difference in u* can grow every tick), so: the first ticks must agree to 1e-7 on the
    applied controls for every ego, all K ticks to U_TOL for at least 99 % of the egos (95 % at T = 40, where a hard-braking
    ego amplifies a difference ~40x per tick: DESIGN.md section 4), respawn counts within 1 %."""
    B, K = 256, 25
    batch = pkg.synth.make_ego_batch(routes, B, T, seed=21, near_end_frac=0.3)
    eng = engine(pkg, routes, batch, T)
    loop = pkg.ClosedLoop(eng, torch.from_numpy(batch.x0).to(eng.device), hist_cap=K, max_age=60)
    loop.run(K)
    torch.cuda.synchronize()
    p = oracle.make_params(T=T)
    cx, cy, cyaw, off = pkg.synth.pack_paths(routes)
    st = oracle.loop_state_from_batch(batch, T)
    r = oracle.closed_loop(p, st, cx, cy, cyaw, off, K, max_age=60, n_threads=8)
    hist = loop.hist[:K].cpu().numpy()
    d = np.abs(hist - r["hist"]).max(axis=2)                # [K, B]
    assert d[:3].max() <= 1e-7, d[:3].max()
    good = (d.max(axis=0) <= U_TOL).mean()
    print(f"T={T}: closed loop {K} ticks, first-3-tick max diff {d[:3].max():.2e}, egos within 1e-4 over all ticks: {good * 100:.1f}%, "
          f"respawns {int(loop.n_respawn.item())} vs {r['n_respawn']}")
    assert good >= (0.99 if T <= 20 else 0.95)
    assert abs(int(loop.n_respawn.item()) - r["n_respawn"]) <= max(2, 0.01 * B)


_REHEARSAL_WORKER = r"""
import importlib, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["JSIM_REPO"])
pkg = importlib.import_module("av-simulation-at-intersections_amd")
S = pkg.synth
dist.init_process_group("gloo")                      # two ranks on ONE GPU: RCCL needs a GPU per rank, gloo does not
rank, world = dist.get_rank(), dist.get_world_size()
T, B, K = 20, 301, 12                                # ragged shards: 151 + 150
routes = S.make_route_table()
for r in routes:
    S.smooth_yaw_inplace(r[:, 2])
batch = S.make_ego_batch(routes, B, T, seed=9, near_end_frac=0.3)

def run(b):
    eng = pkg.BatchedMPC(routes, b.path_id, dl=S.DL, T=T, speed=b.speed, device="cuda:0", smooth=False)
    eng.load_state(b.target_ind, b.oa, b.od, b.path_len)
    loop = pkg.ClosedLoop(eng, torch.from_numpy(b.x0).cuda(), hist_cap=K, max_age=50)
    loop.run(K)
    torch.cuda.synchronize()
    return loop.hist[:K].permute(1, 0, 2).contiguous(), loop.x0.clone()

lo, hi = pkg.sharding.shard_range(B, rank, world)
sub = S.EgoBatch(**{k: getattr(batch, k)[lo:hi] for k in ("x0", "path_id", "path_len", "target_ind", "speed", "oa", "od")})
h_local, x_local = run(sub)                           # this rank's shard: no exchange on the solve path
h_all = pkg.sharding.gather_rows(h_local, B)          # the job's one collective
x_all = pkg.sharding.gather_rows(x_local, B)
h_ref, x_ref = run(batch)                             # the unsharded job, same GPU
assert h_all.shape == (B, K, 2) and x_all.shape == (B, 4)
assert torch.equal(h_all, h_ref), "sharded history != unsharded"
assert torch.equal(x_all, x_ref), "sharded final states != unsharded"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_rank_rehearsal_equals_unsharded(tmp_path):
    """shard -> solve (HIP) -> gather == unsharded, bit for bit, two processes on the one GPU of the box."""
    script = tmp_path / "w.py"
    script.write_text(_REHEARSAL_WORKER)
    import socket
    with socket.socket() as sk:                      # a free port, like bench.spawn_ranks (a fixed one collides with a lingering rendezvous)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, JSIM_REPO=REPO, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


def test_bench_gpus2_started_plainly_spawns_its_ranks():
    """The way the driver starts it: `python bench.py --gpus 2 ...` with no launcher.  JSIM_BENCH_REHEARSAL=1 puts both ranks
    on GPU 0 with a gloo gather (one-GPU box); everything else is the real N > 1 path, extras (config 4 / 5 shares) included."""
    env = dict(os.environ, JSIM_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                          "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 20 and j["config"]["egos_total"] == 512 and j["value"] > 0
    assert j["roofline"]["bound"] == "issue-latency" and "traffic_source" in j["roofline"]
    # the collective block says what the communicator itself reports (here: gloo, two ranks, both blocks arrived with data)
    c = j["collective"]
    assert c["ranks"] == 2 and "gloo" in c["backend"] and c["rank_blocks_with_data"] == 2
    assert c["bytes_per_rank"] == 256 * 20 * 2 * 8 and c["bytes_total"] == 2 * c["bytes_per_rank"] and c["ms"] >= 0.0
    assert set(j["extra"]) == {"config4", "config5"}
    assert j["extra"]["config4"]["egos_total"] == 8192 and j["extra"]["config5"]["horizon"] == 40


def test_rebind_to_a_replanned_path_keeps_the_warm_start(pkg, oracle, routes):
    """set_trajectory_fromarray with a path that is NOT a prefix of the bound one (a re-plan, as interactive_mpc /
    ego_instance do) re-uploads the path table; the controller state -- target_ind, warm start oa / odelta, di -- must
    survive like in the reference (main/lib/mpc.py:279-282): the next solve equals the oracle fed the OLD warm start."""
    r = routes[0].copy()
    car = pkg.BicycleModelDimensions()
    mpc = pkg.MPC(cx=r[:, 0], cy=r[:, 1], cyaw=r[:, 2].copy(), dl=pkg.synth.DL, car_dimensions=car, speed=30 / 3.6, dt=0.2)
    p = oracle.make_params(T=13)
    st = np.array([r[40, 0], r[40, 1], 4.0, r[40, 2]])
    for _ in range(4):                                      # a few ticks on the original path: builds a non-trivial warm start
        di, ai = mpc.step(pkg.State(x=st[0], y=st[1], yaw=st[3], v=st[2]))
        st = oracle.plant_step(p, st, ai, di)
    oa_old, od_old, tind_old = mpc.oa.copy(), mpc.odelta.copy(), mpc.target_ind
    assert np.abs(oa_old).max() > 0.1
    # the re-planned path: the same route shifted sideways by 0.4 m (not a prefix of the bound table)
    r2 = r.copy()
    r2[:, 0] -= 0.4 * np.sin(r[:, 2]); r2[:, 1] += 0.4 * np.cos(r[:, 2])
    mpc.set_trajectory_fromarray(r2)
    assert mpc.oa is not None and mpc.target_ind == tind_old
    di, ai = mpc.step(pkg.State(x=st[0], y=st[1], yaw=st[3], v=st[2]))
    warm = oracle.mpc_step(p, (st[0], st[1], st[3], st[2]), r2[:, 0], r2[:, 1], r2[:, 2], tind_old, 30 / 3.6, oa=oa_old, od=od_old)
    cold = oracle.mpc_step(p, (st[0], st[1], st[3], st[2]), r2[:, 0], r2[:, 1], r2[:, 2], tind_old, 30 / 3.6)
    assert mpc.status == warm["status"] == 0 and mpc.target_ind == warm["target_ind"]
    np.testing.assert_allclose(mpc.oa, warm["oa"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(mpc.odelta, warm["od"], rtol=0, atol=1e-7)
    assert mpc.active_constraints == warm["active"]
    # the test would not notice a dropped warm start if cold and warm solves coincided
    assert max(np.abs(warm["oa"] - cold["oa"]).max(), np.abs(warm["od"] - cold["od"]).max()) > 1e-5
    # a caller who sets the warm start by hand is honoured too
    mpc.oa, mpc.odelta = oa_old.copy(), od_old.copy()
    mpc.target_ind = tind_old
    mpc.step(pkg.State(x=st[0], y=st[1], yaw=st[3], v=st[2]))
    np.testing.assert_allclose(mpc.oa, warm["oa"], rtol=0, atol=1e-7)


def test_two_contexts_leave_the_callers_device_alone(pkg, routes):
    """Entry points run on their context's device and restore the caller's current device (one GPU here: the guard must at
    least be a no-op that keeps torch's current device)."""
    before = torch.cuda.current_device()
    batch = pkg.synth.make_ego_batch(routes, 8, 13, seed=2)
    e1, e2 = engine(pkg, routes, batch, 13), engine(pkg, routes, batch, 13)
    x0 = torch.from_numpy(batch.x0).cuda()
    e1.solve(x0); e2.solve(x0)
    torch.cuda.synchronize()
    assert torch.cuda.current_device() == before
    assert torch.equal(e1.oa, e2.oa) and torch.equal(e1.active_mask, e2.active_mask)
    out = e1.read_back()
    assert np.array_equal(out["oa"], e1.oa.cpu().numpy()) and np.array_equal(out["status"], e1.status.cpu().numpy())
    assert np.array_equal(out["xref"], e1.xref.cpu().numpy()) and np.array_equal(out["target_ind"], e1.target_ind.cpu().numpy())


def test_cabi_gather_single_rank(pkg, routes):
    """jsim_comm_unique_id / jsim_comm_init / jsim_mpc_gather / jsim_comm_destroy (RCCL's ncclAllGather called by the library
    itself) on the one GPU of the box: a communicator of one rank, the gather is then a device copy -- what can be checked
    without a second GPU is that RCCL loads, the communicator comes up on the context's device and the bytes arrive.
    (The N-rank form is the same call; bench.py takes it with JSIM_GATHER=cabi.)"""
    batch = pkg.synth.make_ego_batch(routes, 32, 13, seed=3)
    eng = engine(pkg, routes, batch, 13)
    g = pkg.sharding.CabiGather(eng, rank=0, world=1)
    local = torch.arange(32 * 6, dtype=torch.float64, device=eng.device).reshape(32, 3, 2)
    out = g.gather_rows(local, 32)
    torch.cuda.synchronize()
    assert out.shape == (32, 3, 2) and torch.equal(out, local) and out.data_ptr() != local.data_ptr()
    # a second init on the same context is refused, destroy is idempotent
    buf = (__import__("ctypes").c_char * 128)()
    assert eng.lib.jsim_comm_init(eng._ctx, buf, 1, 0) < 0
    g.close(); g.close()
    assert eng.lib.jsim_mpc_gather(eng._ctx, None, local.data_ptr(), out.data_ptr(), 8, None) < 0   # no communicator any more
    assert b"no communicator" in eng.lib.jsim_last_error(eng._ctx)


@pytest.mark.parametrize("T,scenario", ((13, False), (20, True), (30, True), (40, False), (40, True)))
def test_launch_order_changes_when_not_what(pkg, routes, T, scenario):
    """jsim_mpc_set_launch_order: with B >= 512 the fused launches hand workgroup b the ego that ranked b-th by the
    iterations of the previous launch.  (1) the history, final state and per-ego outputs are those of the identity order bit
    for bit (three launches, plain loop and scenario loop with its in-kernel glue); (2) the order used by launch k+1 is
    the stable descending sort of the iteration counts launch k recorded."""
    import ctypes as C
    B, K = 640, 4
    batch = pkg.synth.make_ego_batch(routes, B, T, seed=23, near_end_frac=0.2)
    specs = [dict(direction=1, turning=False, speed=25 / 3.6, offset=None), dict(direction=-1, turning=True, speed=20 / 3.6, offset=1.0)]

    def make(enabled):
        eng = engine(pkg, routes, batch, T)
        pkg._cabi.check(eng.lib.jsim_mpc_set_launch_order(eng._ctx, 1 if enabled else 0), eng._ctx)
        x0 = torch.from_numpy(batch.x0).to(eng.device)
        if scenario:
            sc = pkg.ScenarioLoop(eng, x0, specs, hist_cap=3 * K, max_age=400)
            return eng, sc.loop, sc.run
        loop = pkg.ClosedLoop(eng, x0, hist_cap=3 * K, max_age=400)
        return eng, loop, loop.run

    def order_and_work(eng):
        o = np.zeros(B, dtype=np.int32); w = np.zeros(B, dtype=np.uint32)
        pkg._cabi.check(eng.lib.jsim_mpc_get_launch_order(eng._ctx, B, o.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p)), eng._ctx)
        return o, w

    e0, l0, run0 = make(False)
    e1, l1, run1 = make(True)
    prev_work = None
    for launch in range(3):
        run0(K); run1(K)
        order, work = order_and_work(e1)
        assert sorted(order.tolist()) == list(range(B))
        if prev_work is None:
            assert np.array_equal(order, np.arange(B))               # nothing known yet: identity
        else:
            assert np.array_equal(order, np.argsort(-prev_work.astype(np.int64), kind="stable"))
            assert not np.array_equal(order, np.arange(B))
        assert work.sum() > 0 and work.max() <= K * (50 * 2 * T + 100)
        prev_work = work
    torch.cuda.synchronize()
    assert torch.equal(l0.hist, l1.hist)
    assert torch.equal(l0.x0, l1.x0) and torch.equal(l0.age, l1.age)
    for name in ("oa", "od", "ox", "oy", "ov", "oyaw", "xref", "target_ind", "status", "n_iter", "active_mask", "di_ai", "path_len"):
        assert torch.equal(getattr(e0, name), getattr(e1, name)), name
    with pytest.raises(pkg._cabi.JsimError):     # the identity-order context never ranked anything
        pkg._cabi.check(e0.lib.jsim_mpc_get_launch_order(e0._ctx, B, None, None), e0._ctx)
