"""CPU-side checks of the product's host logic: the C-ABI library loads and exports every symbol the header
declares (no compute calls without a GPU), config parsing, sharding arithmetic, the N>1 gather with gloo,
and the loud failure when no HIP device is present."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO


def test_cabi_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(REPO, "include", "jsim_mpc.h")).read()
    declared = set(re.findall(r"\b(jsim_[a-z_0-9]+)\s*\(", hdr)) - {"jsim_lds_doubles"}
    assert declared == set(pkg._cabi.EXPORTS)
    lib = pkg.build.build()              # (re)build if stale; hipcc cross-compiles without a GPU
    so = ctypes.CDLL(lib)
    for name in declared:
        assert hasattr(so, name), name
    assert pkg._cabi.load().jsim_abi_version() == 2


def test_cfg_struct_matches_header_layout(pkg):
    cfg = pkg._cabi.make_cfg(pkg.MPCConfig.from_json(), T=13, dt=0.2, dl=0.083, L=2.86)
    assert ctypes.sizeof(cfg) == 8 + 8 * (3 + 2 + 2 + 2 + 2 + 4 + 2 + 3 + 3 + 1 + 2) + 8 + 8   # .. + (nx, reserved_) + jerk_weight
    assert cfg.T == 13 and cfg.max_iter == 1 and cfg.nx == 4
    assert list(cfg.Qf) == [1.0, 1.0, 0.0, 0.5] and list(cfg.R_end) == [10.0, 10.0]
    assert cfg.max_dsteer == np.deg2rad(30.0) and cfg.max_steer == np.deg2rad(45.0)


def test_create_argument_errors_without_gpu(pkg):
    lib = pkg._cabi.load()
    ctx = ctypes.c_void_p()
    cfg = pkg._cabi.make_cfg(pkg.MPCConfig.from_json(), T=49, dt=0.2, dl=0.083, L=2.86)
    assert lib.jsim_mpc_create(ctypes.byref(cfg), 0, ctypes.byref(ctx)) < 0
    assert b"T=49" in lib.jsim_last_error(None)
    cfg = pkg._cabi.make_cfg(pkg.MPCConfig.from_json(), T=13, dt=0.2, dl=0.083, L=2.86)
    cfg.nx = 6
    assert lib.jsim_mpc_create(ctypes.byref(cfg), 0, ctypes.byref(ctx)) < 0
    assert b"NX=6" in lib.jsim_last_error(None)
    assert lib.jsim_mpc_create(None, 0, ctypes.byref(ctx)) < 0
    assert lib.jsim_mpc_step(None, 1, *([None] * 16)) < 0


def test_module_constants_mirror_reference_names(pkg):
    m = pkg.mpc
    assert (m.NX, m.NU, m.T, m.MAX_ITER) == (4, 2, 13, 1)
    assert m.MAX_ACCEL == 2.0 and m.MAX_DECEL == -10
    assert m.MAX_DSTEER == np.deg2rad(30.0)
    assert np.array_equal(m.Qf, np.diag([1.0, 1.0, 0.0, 0.5]) * 13)
    assert issubclass(m.MPCSolutionNotFoundException, Exception)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_hip_device(pkg, routes):
    with pytest.raises(pkg._cabi.JsimError, match="no CPU fallback"):
        pkg.BatchedMPC(routes, [0], dl=0.083)


def test_shard_ranges(pkg):
    S = pkg.sharding
    for B in (0, 1, 7, 256, 32768, 1000):
        for w in (1, 2, 3, 8):
            rs = [S.shard_range(B, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == B
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert S.shard_sizes(B, w) == [hi - lo for lo, hi in rs]
    assert S.shard_range(32768, 3, 8) == (12288, 16384)


_WORKER = r"""
import importlib, os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["JSIM_REPO"])
S = importlib.import_module("av-simulation-at-intersections_amd").sharding
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
for B in (10, 7):
    lo, hi = S.shard_range(B, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float64)[:, None] * torch.ones(1, 3, dtype=torch.float64)
    full = S.gather_rows(local, B)
    assert full.shape == (B, 3), full.shape
    assert torch.equal(full[:, 0], torch.arange(B, dtype=torch.float64)), full
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_gather_rows_world2_gloo(tmp_path):
    """The N>1 path (shard -> solve -> gather) with two CPU processes over gloo, equal and ragged shards."""
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, JSIM_REPO=REPO, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_integration_doc_struct_matches_binding(pkg):
    """INTEGRATION.md shows the ctypes struct a maintainer would write: its field list must be the binding's (and so the
    header's) -- name, type and order."""
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    block = doc[doc.index("class jsim_cfg(C.Structure):"):]
    block = block[:block.index("\n\n")]
    fields = re.findall(r'\("(\w+)",\s*C\.(c_\w+)(?:\s*\*\s*(\d+))?\)', block)
    want = []
    for name, ct in pkg._cabi.JsimCfg._fields_:
        if hasattr(ct, "_length_"):
            want.append((name, ct._type_.__name__, str(ct._length_)))
        else:
            want.append((name, ct.__name__, ""))
    # ctypes spells c_int32 as c_int on this platform: compare sizes, not alias names
    size = {"c_int32": 4, "c_int": 4, "c_double": 8}
    assert [(n, size[t], k) for n, t, k in fields] == [(n, size[t], k) for n, t, k in want]
    hdr = open(os.path.join(REPO, "include", "jsim_mpc.h")).read()
    body = hdr[hdr.index("typedef struct jsim_cfg {"):hdr.index("} jsim_cfg;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    hnames = []
    for decl in re.findall(r"(?:int32_t|double)\s+([^;]+);", body):
        hnames += [re.sub(r"\[\d+\]", "", v).strip() for v in decl.split(",")]
    assert hnames == [n for n, _, _ in fields]
    assert f"jsim_abi_version() == {pkg._cabi.ABI_VERSION}" in doc
    # the table of entry points names every export of the library (and the header declares each: test_cabi_exports)
    assert [e for e in pkg._cabi.EXPORTS if f"`{e}`" not in doc and f"`{e}` " not in doc and e not in doc] == []


def test_lib_shim_resolves_like_the_scenario_scripts_expect(tmp_path):
    """`from lib.mpc import MPC, MAX_ACCEL` (main/scenarios/mpc_intersection.py:20) with <repo>/shim ahead on sys.path gives
    the HIP drop-ins, while other `lib.*` modules still come from the other `lib` directory on sys.path (here a stand-in
    for the reference's main/lib with a simulation.py and its own mpc.py that must NOT win)."""
    other = tmp_path / "main" / "lib"
    other.mkdir(parents=True)
    (other / "__init__.py").write_text("")
    (other / "simulation.py").write_text("class State:\n    pass\nWHO = 'reference-side lib'\n")
    (other / "mpc.py").write_text("raise ImportError('the reference-side lib.mpc must be shadowed by the shim')\n")
    scen = tmp_path / "main" / "scenarios"
    scen.mkdir()
    code = (
        "import sys\n"
        "sys.path.append('..')\n"                                   # what every scenario script does first
        "from lib.mpc import MPC, MAX_ACCEL, MPCSolutionNotFoundException\n"
        "from lib.simulation import State, WHO\n"
        "import lib.mpc_with_speed, lib.mpc_sensitivity, lib.mpc_jerk\n"
        "from lib.mp_search_ww_generic import MotionPrimitiveSearch\n"            # main/scenarios/mpc_intersection.py:17
        "assert MotionPrimitiveSearch.__module__ == 'av-simulation-at-intersections_amd.planner'\n"
        "assert MPC.__module__ == 'av-simulation-at-intersections_amd.mpc', MPC.__module__\n"
        "assert MAX_ACCEL == 2.0 and WHO == 'reference-side lib'\n"
        "assert lib.mpc_jerk.NX == 5 and lib.mpc_with_speed.MAX_DECEL == -5\n"
        "assert lib.mpc_sensitivity.MPC.__module__.endswith('mpc_sensitivity')\n"
        "print('shim ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(REPO, "shim"))
    out = subprocess.run([sys.executable, "-c", code], cwd=str(scen), env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim ok" in out.stdout


_SHARD_WORKER = r"""
import importlib, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["JSIM_REPO"]); sys.path.insert(0, os.path.join(os.environ["JSIM_REPO"], "oracle"))
pkg = importlib.import_module("av-simulation-at-intersections_amd")
import oracle_py as O
S = pkg.synth
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
T, B, K = 13, 21, 3                                  # ragged shards: 11 + 10
routes = S.make_route_table()
for r in routes:
    S.smooth_yaw_inplace(r[:, 2])
batch = S.make_ego_batch(routes, B, T, seed=5)
p = O.make_params(T=T)
cx, cy, cyaw, off = S.pack_paths(routes)
lo, hi = pkg.sharding.shard_range(B, rank, world)
sub = S.EgoBatch(**{k: getattr(batch, k)[lo:hi] for k in ("x0", "path_id", "path_len", "target_ind", "speed", "oa", "od")})
st = O.loop_state_from_batch(sub, T)
r = O.closed_loop(p, st, cx, cy, cyaw, off, K)      # this rank's egos only: no exchange on the solve path
full = pkg.sharding.gather_rows(torch.from_numpy(np.ascontiguousarray(r["hist"].transpose(1, 0, 2))), B)   # [B, K, 2]
stw = O.loop_state_from_batch(batch, T)
rw = O.closed_loop(p, stw, cx, cy, cyaw, off, K)    # the unsharded job
assert full.shape == (B, K, 2)
assert np.array_equal(full.numpy(), rw["hist"].transpose(1, 0, 2)), "sharded != unsharded"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_shard_solve_gather_equals_unsharded_world2_gloo(tmp_path, oracle):
    """shard -> solve -> gather == the unsharded job, two CPU processes over gloo, ragged shards.  The product has no CPU
    solve, so the CPU oracle stands in for the per-rank solve here (what is under test is the sharding arithmetic and the
    gather); the same check with the HIP solve is tests/test_gpu_configs.py::test_two_rank_rehearsal_equals_unsharded."""
    script = tmp_path / "w.py"
    script.write_text(_SHARD_WORKER)
    env = dict(os.environ, JSIM_REPO=REPO, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_bench_spawns_its_own_ranks_before_touching_torch():
    """`python bench.py --gpus N` started plainly must start its N ranks as a child job before importing torch (here: the
    spawn command is built and handed to subprocess; no GPU involved)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 0
    real = bench.subprocess.call
    bench.subprocess.call = fake_call
    argv = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "4", "--steps", "20"]
        rc = bench.spawn_ranks(bench.parse_args(["--gpus", "4", "--steps", "20"]))
    finally:
        bench.subprocess.call = real
        sys.argv = argv
    assert rc == 0
    cmd = calls["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "20"] and cmd[-5].endswith("bench.py")
    assert calls["env"]["MASTER_ADDR"] == "127.0.0.1" and calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    src = open(os.path.join(REPO, "bench.py")).read()
    assert src.index("sys.exit(spawn_ranks(args))") < src.index("    import torch\n")


def test_multi_lane_route_table_geometry(pkg):
    """Config 5's routes: two-lane geometry of main/envs/intersection_multi_lanes.py (lane centres 3 m / 7 m from the axis,
    start / goal distance 30 m), uniform spacing dl, [x, y, yaw] with yaw wrapped like the planner's output."""
    S = pkg.synth
    rs = S.make_route_table(multi_lane=True)
    assert len(rs) == 48
    k = 0
    for sp in (1, 2, 3, 4):
        for tn in (1, 2, 3):
            for sl in (1, 2):
                for gl in (1, 2):
                    r = rs[k]; k += 1
                    d = np.hypot(np.diff(r[:, 0]), np.diff(r[:, 1]))
                    assert abs(d - S.DL).max() < 5e-4
                    assert np.all(np.abs(r[:, 2]) <= np.pi + 1e-12)
                    if sp == 1:
                        assert abs(r[0, 0] - (3.0 + 4.0 * (sl - 1))) < 1e-9 and abs(r[0, 1] + 30.0) < 1e-9
                        end = {1: (-30.0, 3.0 + 4.0 * (gl - 1)), 2: (3.0 + 4.0 * (gl - 1), 30.0), 3: (30.0, -(3.0 + 4.0 * (gl - 1)))}[tn]
                        assert np.hypot(r[-1, 0] - end[0], r[-1, 1] - end[1]) < 0.1
    assert len(S.make_route_table()) == 12


def test_gather_entry_points_argument_errors_without_gpu(pkg):
    lib = pkg._cabi.load()
    assert lib.jsim_mpc_gather(None, None, None, None, 8, None) < 0 and b"null ctx" in lib.jsim_last_error(None)
    assert lib.jsim_comm_init(None, None, 1, 0) < 0
    assert lib.jsim_comm_unique_id(None) < 0
    assert lib.jsim_comm_destroy(None) < 0


def test_plan_routes_argument_errors_without_gpu(pkg):
    """jsim_plan_routes checks sizes, null pointers and its offset tables on the host, before anything touches a device."""
    import ctypes as C
    lib = pkg._cabi.load()
    PL = pkg.planner
    pts, length = PL.make_motion_primitives()
    rad, cen = PL.car_circles()
    cc = np.concatenate([PL.collision_points(p, cen, rad) for p in pts], axis=0)
    cc_off = np.concatenate([[0], np.cumsum([len(PL.collision_points(p, cen, rad)) for p in pts])]).astype(np.int32)
    q = PL.intersection_query(1, 1, rad)
    hp = np.ascontiguousarray(np.concatenate(q.obstacles, axis=0))
    hp_off = np.concatenate([[0], np.cumsum([len(o) for o in q.obstacles])]).astype(np.int32)
    r_off = np.array([0, len(q.obstacles)], dtype=np.int32)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    start, goal, box, tol = f64([q.start]), f64([q.goal]), f64([q.goal_box]), f64([q.tol])
    wh, wc = f64(PL.WH_DEFAULT), f64(PL.WC_DEFAULT)
    mp, ml = f64(pts), f64(length)
    max_path = 8
    status = np.zeros(1, np.int32); cost = np.zeros(1); n_prims = np.zeros(1, np.int32); prims = np.zeros((1, max_path), np.int32)
    nodes = np.zeros((1, max_path + 1, 3)); traj = np.zeros((1, max_path * 60, 3)); n_exp = np.zeros(1, np.int32)
    p = lambda a: C.c_void_p(a.ctypes.data)

    def call(hp_off_=hp_off, r_off_=r_off, cc_off_=cc_off, n_prim=9, node_cap=4096, start_=start):
        return lib.jsim_plan_routes(0, 1, p(start_) if start_ is not None else None, p(goal), p(box), p(tol), p(hp), p(hp_off_), len(hp_off) - 1,
                                    p(r_off_), p(mp), p(ml), n_prim, 61, p(cc), p(cc_off_), p(wh), p(wc), max_path, node_cap, p(status), p(cost),
                                    p(n_prims), p(prims), p(nodes), p(traj), p(n_exp))
    assert call(n_prim=0) == -22 and b"bad sizes" in lib.jsim_last_error(None)
    assert call(node_cap=8) == -22
    assert call(start_=None) == -22 and b"null argument" in lib.jsim_last_error(None)
    bad = hp_off.copy(); bad[2] = bad[1] - 1
    assert call(hp_off_=bad) == -22 and b"hp_off" in lib.jsim_last_error(None)
    assert call(r_off_=np.array([0, len(q.obstacles) + 1], dtype=np.int32)) == -22 and b"route_obs_off" in lib.jsim_last_error(None)
    bad = cc_off.copy(); bad[3] = bad[2] - 1
    assert call(cc_off_=bad) == -22 and b"cc_off" in lib.jsim_last_error(None)


def test_bench_line_helpers():
    """bench.py's pure pieces: the kernel launch_reg dispatches per (T, scenario, B), the straggler
    statistic from per-ego iteration totals, the flop / byte formulas of SURVEY 8d, the newest committed PMC summary."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.kernel_name(40, False, 1024) == "mpc_step_reg4_kernel<40, false>" and bench.kernel_name(40, True, 64) == "mpc_step_reg4_kernel<40, true>"
    assert bench.kernel_name(20, False, 256) == "mpc_step_reg_kernel<20, false, 1, true>" and bench.kernel_name(20, False, 1024) == "mpc_step_reg_kernel<20, false, 1>"
    assert bench.kernel_name(20, False, 257) == "mpc_step_reg_kernel<20, false, 1>" and bench.kernel_name(20, True, 64) == "mpc_step_reg_kernel<20, true, 1, true>" and bench.kernel_name(16, True, 64) == "mpc_step_reg_kernel<16, true, 1>"
    assert bench.kernel_name(20, False, 1025) == "mpc_step_reg_kernel<20, false, 2>" and bench.kernel_name(20, True, 4096) == "mpc_step_reg_kernel<20, true, 1>"
    assert bench.kernel_name(13, False, 8) == "mpc_step_reg_kernel<13, false, 1, true>" and bench.kernel_name(13, False, 257) == "mpc_step_reg_kernel<13, false, 2>"
    assert bench.kernel_name(30, True, 4096) == "mpc_step_reg_kernel<30, true, 1>" and bench.kernel_name(30, False, 8) == "mpc_step_reg_kernel<30, false, 1>"
    assert bench.kernel_name(25, False, 8) == "mpc_step_reg_kernel<25, false, 1, true>" and bench.kernel_name(25, False, 300) == "mpc_step_reg_kernel<25, false, 1>" and bench.kernel_name(32, True, 8) == "mpc_step_reg4_kernel<32, true>"
    assert bench.kernel_name(24, False, 8) == "mpc_step_kernel" and bench.kernel_name(48, False, 8) == "mpc_step_kernel"
    assert bench.waves_per_ego(bench.kernel_name(20, False, 256)) == "1 + 3 helper" and bench.waves_per_ego(bench.kernel_name(20, True, 256)) == "1 + 3 helper" and bench.waves_per_ego(bench.kernel_name(20, True, 4096)) == 1
    assert bench.waves_per_ego(bench.kernel_name(40, False, 8)) == 4 and bench.waves_per_ego(bench.kernel_name(20, False, 4096)) == 1
    src = open(os.path.join(REPO, "av-simulation-at-intersections_amd", "csrc", "jsim_mpc.hip")).read()
    assert "return e ? atoi(e) : 1025;" in src and "hipDeviceAttributeMultiprocessorCount" in src   # the dispatch thresholds kernel_name() mirrors (256 CUs)
    # the horizon lists kernel_name() reads are the library's own
    import re
    cfg = importlib.import_module("av-simulation-at-intersections_amd.config")
    for macro, lst in (("JSIM_ONE_WAVE_HORIZONS", cfg.ONE_WAVE_HORIZONS), ("JSIM_FOUR_WAVE_HORIZONS", cfg.FOUR_WAVE_HORIZONS), ("JSIM_HELP_HORIZONS", cfg.HELP_HORIZONS), ("JSIM_HELP_PRE_HORIZONS", cfg.HELP_PRE_HORIZONS)):
        line = re.search(r"#define %s\(X\)(.*)" % macro, src).group(1)
        assert tuple(int(t) for t in re.findall(r"X\((\d+)\)", line)) == tuple(lst), macro
    # the split build's per-horizon units: the `#if JSIM_KERNEL_TU == ..` lists and build.py's unit list repeat the same horizons
    for tag, lst in (("four-wave horizons", cfg.FOUR_WAVE_HORIZONS), ("one-wave horizons", cfg.ONE_WAVE_HORIZONS), ("helper-wavefront horizons */", cfg.HELP_HORIZONS), ("helper-wavefront horizons with the glue", cfg.HELP_PRE_HORIZONS)):
        line = [ln for ln in src.splitlines() if ln.startswith(("#if JSIM_KERNEL_TU ==", "#elif JSIM_KERNEL_TU ==")) and tag in ln]
        assert len(line) == 1 and tuple(int(t) for t in re.findall(r"JSIM_KERNEL_TU == (\d+)", line[0])) == tuple(lst), tag
    build_mod = importlib.import_module("av-simulation-at-intersections_amd.build")
    assert tuple(build_mod.KERNEL_TUS) == tuple(sorted(cfg.ONE_WAVE_HORIZONS + cfg.FOUR_WAVE_HORIZONS))
    mean, s = bench.straggler_stats(np.array([100, 100, 100, 700], dtype=np.uint64), 10)
    assert mean == 25.0 and s == {"max_ego_iters_per_tick": 70.0, "slowest_over_mean": 2.8, "egos_above_3x_mean": 0}
    mean, s = bench.straggler_stats(np.array([10] * 99 + [1000], dtype=np.uint64), 10)
    assert abs(mean - 1.99) < 1e-12 and s["egos_above_3x_mean"] == 1 and s["max_ego_iters_per_tick"] == 100.0
    assert bench.algorithmic_bytes_per_step(20) == 2564 and bench.algorithmic_bytes_per_step(13) == 1717
    assert abs(bench.algorithmic_flops_per_step(20, 20) - 0.4826e6) < 1e3                                      # 59 T^3-ish at n_iter = T
    pm = bench.latest_pmc_summary(2)
    assert pm and os.path.basename(pm) >= "r03_config2_pmc_summary.json" and "driver" not in pm
    pm20 = bench.latest_pmc_summary(2, 20)                         # the driver's --steps 20 invocation has its own collection
    assert "driver_invocation" in pm20 and json.load(open(pm20))["ticks_per_launch"] == 20
    assert bench.latest_pmc_summary(2, 100) == pm and bench.latest_pmc_summary(2, 37) == pm
    d = json.load(open(pm))
    assert "FETCH_SIZE" in d and "WRITE_SIZE" in d and d.get("commit", "?") != "?"
