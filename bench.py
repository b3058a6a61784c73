#!/usr/bin/env python3
"""bench.py -- MPC receding-horizon steps/s of the HIP path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}]

`--gpus N` (N > 1) started plainly spawns its own N ranks (one per GPU, `python -m torch.distributed.run`, rendezvous on
127.0.0.1) BEFORE this process imports torch or touches a GPU, relays their output and exits with their code; started
under torch.distributed.run (WORLD_SIZE set) it is one of the ranks.

Workloads (BASELINE.json `configs`; synthetic random-init egos, everything resident in HBM, CLOSED LOOP: one "step" = one
tick = one MPC.step for every ego + plant update + history + goal test / respawn, all K timed ticks in one fused launch):
  --config 2 (default)  256 egos per GPU, horizon 20, fp64, the 12 synthetic intersection routes        [configs[1], the headline]
  --config 3            4096 egos, horizon 30, the scenario loop with four scripted obstacle vehicles: obstacle prediction ->
                        collision check -> path cut-off inside every tick (the reference's dynamic-obstacle mechanism)
  --config 4            4096 egos per GPU (32768 / 8), horizon 20                                          [configs[3]]
  --config 5            1024 egos per GPU (8192 / 8), horizon 40, multi-lane route geometry                [configs[4]]
Weak scaling: every rank owns the same number of egos, no collective on the solve path; the recorded controls are
all-gathered once at the end (RCCL all-gather; the job's only exchange).  With N > 1 and the default config the line also
carries `extra`: the per-rank shares of configs 4 and 5 measured in the same job (same protocol).

Prints ONE JSON line (rank 0).  `value` = egos x ticks / wall time (max over ranks).
`roofline`     : the dominant kernel against the fp64 peak (the path is bound by single-wave fp64 issue latency, not by HBM
                 and not by MFMA throughput: SURVEY D6); `roofline_hbm` carries the algorithmic-bytes-vs-HBM figure.
`cpu_baseline` : the CPU oracle (a C port of the reference path; the reference's cvxpy/ECOS stack cannot be installed)
                 running the same closed loop on this box's host cores, 1 core and all cores.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == fp64 matrix peak
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md

# The workloads (CONFIGS, OBSTACLE_SPECS, route tables, ego batches, loop objects) live in the package's workloads.py: ONE
# definition shared with tests/test_gpu_bench_workloads.py, which checks these very egos and routes against the oracle.
CONFIG_IDS = (2, 3, 4, 5)


def algorithmic_bytes_per_step(T, w=8):
    """w(15T+16) + 16 + T bytes per ego per MPC step."""
    return w * (15 * T + 16) + 16 + T


def algorithmic_flops_per_step(T, n_iter):
    """16T^3 (Hessian GEMM) + 8T^3/3 (Cholesky) + 40T^2 n_iter (active set) + 32T^2 + 60T."""
    return 16 * T ** 3 + 8 * T ** 3 / 3 + 40 * T ** 2 * n_iter + 32 * T ** 2 + 60 * T


def kernel_name(T, scenario, B):
    """The kernel launch_reg (csrc/jsim_mpc.hip) dispatches: <T, PRE, waves per SIMD the register budget is set for>."""
    pre = "true" if scenario else "false"
    import importlib
    cfg = importlib.import_module("av-simulation-at-intersections_amd.config")
    if T in cfg.ONE_WAVE_HORIZONS:
        if B <= 256 and T in (cfg.HELP_PRE_HORIZONS if scenario else cfg.HELP_HORIZONS):   # one ego per CU at most: three helper wavefronts per ego
            return f"mpc_step_reg_kernel<{T}, {pre}, 1, true>"
        wpe = 2 if (not scenario and (T == 13 or (T == 20 and B > 1024))) else 1
        return f"mpc_step_reg_kernel<{T}, {pre}, {wpe}>"
    if T in cfg.FOUR_WAVE_HORIZONS:
        return f"mpc_step_reg4_kernel<{T}, {pre}>"
    return "mpc_step_kernel"


def waves_per_ego(kname):
    """Wavefronts that work on one ego in the kernel kernel_name() returned (for the roofline note)."""
    if "reg4" in kname:
        return 4
    if kname.startswith("mpc_step_reg_kernel<") and kname.count(",") == 3 and kname.endswith(", true>"):
        return "1 + 3 helper"
    return 1


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", type=int, choices=CONFIG_IDS, default=2, help="BASELINE.json workload (see module docstring)")
    ap.add_argument("--batch", type=int, default=0, help="egos per GPU (0: the config's)")
    ap.add_argument("--horizon", type=int, default=0, help="horizon (0: the config's)")
    ap.add_argument("--mode", choices=("fused", "graph", "eager"), default="fused",
                    help="fused: K ticks per launch inside the kernel (default); graph: one launch pair per tick "
                         "replayed from a hipGraph; eager: one launch pair per tick from Python")
    ap.add_argument("--ticks-per-launch", type=int, default=0,
                    help="fused mode: closed-loop ticks per kernel launch (0 = all K ticks of the timed region in one launch; "
                         "egos only wait for each other at launch boundaries)")
    ap.add_argument("--routes", choices=("planner", "synthetic"), default="planner",
                    help="planner (default): the route table is planned on the GPU (jsim_plan_routes: A* over motion primitives on the "
                         "reference's intersection geometry, 12 routes / 48 two-lane routes, untimed set-up); synthetic: idealised arcs")
    ap.add_argument("--respawn", choices=("initial", "start"), default="initial",
                    help="where an ego that reached its goal (or timed out) re-enters: its own random initial state, or the first point of "
                         "its route at standstill like the reference's scenarios start their vehicle")
    ap.add_argument("--no-extra", action="store_true", help="N > 1: skip the config 4 / config 5 per-rank shares")
    ap.add_argument("--no-respawn-start", action="store_true",
                    help="skip the second, shorter run of the same workload under the other respawn rule (reported beside the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` started plainly: start the N ranks as a CHILD job.  Nothing in this process has imported
    torch or initialised the GPU, and nothing is exec'ed: the parent only waits and hands the child's exit code on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import importlib

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    # JSIM_BENCH_REHEARSAL=1: all ranks share GPU 0 and the final gather runs over gloo -- lets the N>1 code path be
    # exercised on a one-GPU box (the real run is one rank per GPU with RCCL)
    rehearsal = os.environ.get("JSIM_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    pkg = importlib.import_module("av-simulation-at-intersections_amd")
    WL = pkg.workloads
    CONFIGS, OBSTACLE_SPECS = WL.CONFIGS, WL.OBSTACLE_SPECS
    K, W = args.steps, args.warmup

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    _route_cache = {}
    plan_info = {}   # row f4's own measurement: the planner launch that produced the workload's routes (rank 0 reports it)

    def route_table(multi_lane):
        if multi_lane not in _route_cache:
            rs, info = WL.route_table(multi_lane, source=args.routes, device_index=dev_index)
            if info:
                plan_info[multi_lane] = info
            _route_cache[multi_lane] = rs
        return _route_cache[multi_lane]

    def iter_totals(eng, reset):
        tot = np.zeros(eng.B, dtype=np.uint64)
        pkg._cabi.check(eng.lib.jsim_mpc_iter_totals(eng._ctx, eng.B, tot.ctypes.data, 1 if reset else 0), eng._ctx, "jsim_mpc_iter_totals")
        return tot

    def run_workload(cfg_id, B, T, K, W, mode, tpl, respawn):
        """W untimed ticks, then EXACTLY K timed ticks bracketed by barrier + synchronize; returns the figures of this rank
        (elapsed = max over ranks)."""
        cfg = CONFIGS[cfg_id]
        routes = route_table(cfg["multi_lane"])
        batch = WL.ego_batch(routes, B, T, rank=rank)
        eng, x0 = WL.make_engine(routes, batch, T, device)
        sc, loop = WL.make_loop(cfg_id, eng, x0, hist_cap=K + W + 64, routes=routes, batch=batch, respawn=respawn)
        if sc is not None:
            tick, run = sc.tick, sc.run
            mode = "fused" if mode == "graph" else mode
        else:
            tick, run = loop.tick, loop.run

        for _ in range(W):          # warm-up (untimed)
            tick()
        torch.cuda.synchronize(device)
        # the job's one exchange: torch.distributed's all_gather_into_tensor (default), or with JSIM_GATHER=cabi the C-ABI's own
        # jsim_mpc_gather (ncclAllGather called by libjsim_mpc.so; needs one GPU per rank, so not in rehearsal mode)
        cabi_gather = pkg.sharding.CabiGather(eng) if (world > 1 and os.environ.get("JSIM_GATHER") == "cabi" and not rehearsal) else None
        gather = cabi_gather.gather_rows if cabi_gather else pkg.sharding.gather_rows
        if world > 1:   # run once untimed: communicator set-up and buffer registration are not the path
            _ = gather(loop.hist[:max(W, 1)].permute(1, 0, 2).contiguous(), B * world)
            torch.cuda.synchronize(device)

        chunk = next(c for c in (50, 25, 20, 10, 5, 4, 2, 1) if K % c == 0)
        if mode == "fused":
            # the scripted vehicles cross the junction once (~25 s of simulated time) and are gone: config 3 sends them in again
            # at every launch boundary, so that obstacle prediction / collision check / cut-off have work throughout
            chunk = tpl if tpl > 0 else (chunk if cfg["scenario"] else K)
            if K % chunk:
                raise SystemExit(f"--ticks-per-launch {chunk} must divide --steps {K}")
            # untimed: first use of the entry point (the scenario entry sizes its per-launch obstacle buffers on first use, so
            # it is given a launch of the timed size)
            run(chunk if cfg["scenario"] else 1)
            iter_totals(eng, reset=True)    # the counters of the timed launches start at zero (synchronises: outside the timed region)
        elif mode == "graph":
            loop.capture(chunk)     # (capture runs one extra untimed tick)

        # ---- timed region: exactly K ticks.  HIP events are recorded on the stream the kernels are launched on (the
        # engine passes torch's current stream of this device to the C-ABI, and torch.cuda.Event records on that stream)
        evs, ncut = [], []
        pre_evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K // chunk if mode == "fused" else 0)]
        gev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for ea, eb in pre_evs + [gev]:        # (event objects are created lazily on first record: not inside the timed region)
            ea.record(); eb.record()
        if sc is not None:
            # the cut-off statistic below is a torch reduction; its first call loads torch's kernel for it (18 ms seen) -- that is
            # PyTorch's start-up, not the path: first call here, outside the timed region
            _ = (eng.path_len < eng.full_len).sum()
        sync_all()
        t0 = time.perf_counter()
        if mode == "fused":         # closed loop on the device: `chunk` ticks per launch, egos never wait for each other
            for ea, eb in pre_evs:
                if sc is not None:
                    sc.obst.reset()         # a 128-byte device-to-device copy on the launch stream
                    ncut.append((eng.path_len < eng.full_len).sum())     # of the previous launch's last tick (no sync here)
                ea.record(); run(chunk); eb.record()
                evs.append((ea, eb))
        elif mode == "graph":
            for _ in range(K // chunk):
                loop.replay()
        else:
            for _ in range(K):
                tick()
        collective = None
        if world > 1:   # the only exchange of the job: gather every rank's recorded controls (RCCL all-gather)
            hist_local = loop.hist[:K].permute(1, 0, 2).contiguous()          # [B, K, 2]
            gev[0].record()
            hist_all = gather(hist_local, B * world)
            gev[1].record()
            assert hist_all.shape[0] == B * world
        sync_all()
        t1 = time.perf_counter()
        if world > 1:
            # what the communicator itself says it is: backend and rank count come from the process group / the C-ABI's
            # communicator, not from the command line
            collective = {"what": "all-gather of the recorded controls [B, K, 2] f64, once per job",
                          "backend": ("rccl via jsim_mpc_gather (ncclAllGather called by libjsim_mpc.so)" if cabi_gather else
                                      f"torch.distributed {dist.get_backend()}" + (" (= RCCL on ROCm)" if dist.get_backend() == "nccl" else "")),
                          "ranks": int(cabi_gather.world if cabi_gather else dist.get_world_size()),
                          "bytes_per_rank": int(hist_local.numel() * hist_local.element_size()),
                          "bytes_total": int(hist_all.numel() * hist_all.element_size()),
                          # evidence in the data itself: rank blocks of the gathered tensor that arrived non-empty
                          "rank_blocks_with_data": int((hist_all.reshape(world, -1).abs().sum(dim=1) > 0).sum().item()),
                          "ms": float(gev[0].elapsed_time(gev[1]))}
        eng_cut_last = (eng.path_len < eng.full_len).sum() if sc is not None else None
        elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cpu" if rehearsal else device)
        if world > 1:
            dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        elapsed = float(elapsed.item())

        straggler = None
        if mode == "fused":
            # iterations of the TIMED launches, per ego (jsim_mpc_iter_totals: every fused launch adds what each ego needed), and
            # the dominant launch of the timed region IS the fused kernel: HIP events around each one
            mean_iter, straggler = straggler_stats(iter_totals(eng, reset=False), K)
            kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
            launches, ticks_per_launch = len(evs), chunk
        else:
            # non-fused modes: KE more ticks, untimed for `value`, HIP events around the single-tick MPC launch only
            KE = min(K, 100)
            sevs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(KE)]
            n_iter_sum = torch.zeros((), dtype=torch.float64, device=device)
            torch.cuda.synchronize(device)
            for a, b in sevs:
                a.record(); eng.solve(loop.x0); b.record()
                n_iter_sum += eng.n_iter.sum()
                pkg._cabi.check(eng.lib.jsim_loop_advance(
                    eng._ctx, eng.B, loop.x0.data_ptr(), eng.oa.data_ptr(), eng.od.data_ptr(), eng.status.data_ptr(),
                    eng.di_ai.data_ptr(), eng.target_ind.data_ptr(), eng.path_id.data_ptr(), eng.path_len.data_ptr(),
                    loop.x0_spawn.data_ptr(), loop.target_spawn.data_ptr(), loop.age.data_ptr(), loop.max_age, None, None, 0,
                    loop.n_respawn.data_ptr(), eng._stream()), eng._ctx)
            torch.cuda.synchronize(device)
            kern_ms = float(np.mean([a.elapsed_time(b) for a, b in sevs]))   # one single-tick launch
            launches, ticks_per_launch = KE, 1
            mean_iter = float(n_iter_sum.item()) / (KE * B)
        res = dict(cfg_id=cfg_id, B=B, T=T, K=K, W=W, mode=mode, chunk=chunk, elapsed=elapsed, kern_ms=kern_ms,
                   launches=launches, ticks_per_launch=ticks_per_launch, mean_iter=mean_iter, straggler=straggler,
                   n_fail=int((eng.status != 0).sum().item()), respawns=int(loop.n_respawn.item()),
                   value=B * world * K / elapsed, routes=routes, batch=batch, collective=collective,
                   cut=(float(torch.stack(ncut[1:] + [(eng_cut_last)]).double().mean().item()) if sc is not None else None))
        if cabi_gather:
            cabi_gather.close()
        eng.close()
        return res

    cfg = CONFIGS[args.config]
    B = args.batch or cfg["batch"]
    T = args.horizon or cfg["horizon"]
    main_res = run_workload(args.config, B, T, K, W, args.mode, args.ticks_per_launch, args.respawn)
    other = None
    if args.mode == "fused" and not args.no_respawn_start:
        # the same workload and kernels under the OTHER respawn rule, a shorter run of the same protocol: the default rule keeps
        # re-injecting SURVEY 8d's perturbed states (a few egos get stuck beside their path and set the launch time: `straggler`),
        # "start" lets a finished ego re-enter at its route's first point at standstill like the reference's scenarios
        Ko = max(10, min(K, 100))
        other_rule = "start" if args.respawn == "initial" else "initial"
        r = run_workload(args.config, B, T, Ko, min(W, 10), "fused", 0, other_rule)
        other = {"respawn": other_rule, "value": r["value"], "unit": "MPC steps/s", "steps": Ko, "ms_per_step": r["elapsed"] / Ko * 1e3,
                 "mean_active_set_iters": round(r["mean_iter"], 2), "straggler": r["straggler"]}
    extra = {}
    if world > 1 and args.config == 2 and not args.no_extra and not (args.batch or args.horizon):
        for cid in (4, 5):   # the named multi-GPU configurations' per-rank shares, same job, same protocol, fewer ticks
            c = CONFIGS[cid]
            Ke = max(10, min(K, 100 if cid == 4 else 40))
            r = run_workload(cid, c["batch"], c["horizon"], Ke, min(W, 5), "fused", 0, args.respawn)
            extra[f"config{cid}"] = {"workload": c["name"], "value": r["value"], "unit": "MPC steps/s", "egos_total": c["batch"] * world,
                                     "horizon": c["horizon"], "steps": Ke, "ms_per_step": r["elapsed"] / Ke * 1e3,
                                     "mean_active_set_iters": round(r["mean_iter"], 2), "straggler": r["straggler"],
                                     "failed_egos_last_tick": r["n_fail"], "collective": r["collective"]}

    if rank == 0:
        r = main_res
        tpl = r["ticks_per_launch"]
        flops = algorithmic_flops_per_step(T, r["mean_iter"]) * B * tpl
        nbytes = algorithmic_bytes_per_step(T) * B * tpl
        ach_tf = flops / (r["kern_ms"] * 1e-3) / 1e12
        ach_gbs = nbytes / (r["kern_ms"] * 1e-3) / 1e9
        traffic, traffic_source = None, "not measured in this run (PMC counters need their own rocprofv3 passes)"
        pmc_path = latest_pmc_summary(args.config, tpl)
        if r["mode"] == "fused" and (B, T) == (cfg["batch"], cfg["horizon"]) and pmc_path:
            # HBM bytes of one fused launch from the rocprofv3 PMC passes of this same command, collected in a SEPARATE run
            # (tools/collect_profile.sh): FETCH_SIZE doubled (gfx950 tallies 64 B per 128-B request), WRITE_SIZE as is,
            # both in KiB; scaled to this launch's tick count
            pmc = json.load(open(pmc_path))
            if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 * (tpl / float(pmc.get("ticks_per_launch", 100)))
                traffic_source = (f"profiles/{os.path.basename(pmc_path)}: separate rocprofv3 --pmc passes of this command "
                                  f"({pmc.get('ticks_per_launch', 100)} ticks per launch, library at commit {pmc.get('commit', '?')}), "
                                  f"scaled to {tpl} ticks; stale if the kernel changed since")
        kname = kernel_name(T, cfg["scenario"], B)
        waves = waves_per_ego(kname)
        per_simd = 2 if kernel_name(T, cfg["scenario"], B).endswith(", 2>") else 1
        out = {
            "metric": "MPC steps/sec (batch x horizon) at N=20 nu=2",
            "value": r["value"], "unit": "MPC steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": r["elapsed"] / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{B} egos per GPU, horizon N={T}, nu=2, fp64, closed loop -- "
                                   + (cfg["name"] if (B, T) == (cfg["batch"], cfg["horizon"]) else "not a named configuration"),
                       "baseline_config": args.config, "egos_per_gpu": B, "egos_total": B * world, "horizon": T,
                       "routes": (("48 two-lane routes" if cfg["multi_lane"] else "12 intersection routes") +
                                  (" planned on the GPU (A* over motion primitives, the reference's scenario geometry)" if args.routes == "planner"
                                   else ", synthetic arcs")),
                       "respawn": ("an ego that reaches its goal or times out (400 ticks) re-enters in its own initial state (SURVEY 8d's distribution: "
                                   "lateral / heading / speed perturbations keep being injected)" if args.respawn == "initial" else
                                   "an ego that reaches its goal or times out re-enters at the first point of its route at standstill"),
                       "launch": {"fused": f"fused closed loop, {r['chunk']} ticks per launch", "graph": "hipGraph",
                                  "eager": "eager"}[r["mode"]], "parallelism": f"ego-shard x{world}",
                       "mean_active_set_iters": round(r["mean_iter"], 2),
                       "iters_source": ("the timed launches' own per-ego counters (jsim_mpc_iter_totals)" if r["mode"] == "fused" else
                                        "extra single-tick launches after the timed region"),
                       "straggler": r["straggler"], "failed_egos_last_tick": r["n_fail"], "respawns": r["respawns"]},
            "roofline": {"bound": "issue-latency", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel_name(T, cfg["scenario"], B), "kernel_ms": r["kern_ms"],
                         "ticks_per_launch": tpl, "algorithmic_flops_per_launch": flops,
                         "note": ("priced against the fp64 vector = matrix peak; the kernel is bound by the issue latency of "
                                  f"{waves} wave(s) per ego, {per_simd} wave(s) per SIMD ({B} egos on 256 CUs / 1024 SIMDs), not by HBM (roofline_hbm) "
                                  "and not by MFMA throughput (the MFMA pipe is < 1 % busy)")},
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes},
        }
        if other:
            out["other_respawn_rule"] = other
        if r["collective"]:
            out["collective"] = r["collective"]
        if cfg["scenario"]:
            out["config"]["obstacle_vehicles"] = len(OBSTACLE_SPECS)
            out["config"]["egos_cut_off_mean_at_launch_ends"] = r["cut"]
        if extra:
            out["extra"] = extra
        pi = plan_info.get(cfg["multi_lane"])
        if pi:
            out["planner"] = {"what": "jsim_plan_routes (A* over motion primitives, one wavefront per route): the route table of this workload, "
                                      "one call, host arrays in and out (uploads, search, read-back); outside the timed region",
                              "routes": pi["routes"], "wall_ms": pi["wall_ms"], "expanded_nodes": int(sum(pi["expanded"])),
                              "trajectory_points": pi["points"]}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, r["routes"], r["batch"], T, args.cpu_seconds)
            if pi:
                out["planner"]["cpu_port"] = planner_cpu_baseline(pi)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def straggler_stats(totals, K):
    """(mean iterations per step, config.straggler) from the per-ego iteration totals of K timed ticks: a fused launch lasts as
    long as its slowest ego, so the line says who that is -- its iterations per tick, their ratio to the mean, and how many egos
    sit above three times the mean."""
    import numpy as np
    per_tick = np.asarray(totals, dtype=np.float64) / float(K)
    mean = float(per_tick.mean()) if per_tick.size else 0.0
    mx = float(per_tick.max()) if per_tick.size else 0.0
    return mean, {"max_ego_iters_per_tick": round(mx, 2), "slowest_over_mean": round(mx / max(mean, 1e-30), 2),
                  "egos_above_3x_mean": int((per_tick > 3.0 * mean).sum())}


def latest_pmc_summary(config, ticks_per_launch=None):
    """profiles/rNN_config<config>*_pmc_summary.json of the latest round that has one (None if there is none); among that round's
    files the one collected at this launch length wins (the driver's `--steps 20` invocation has its own collection)."""
    import glob
    cands = sorted(glob.glob(os.path.join(REPO, "profiles", f"r[0-9][0-9]_config{config}*_pmc_summary.json")))
    if not cands:
        return None
    rnd = os.path.basename(cands[-1])[:3]
    cands = [c for c in cands if os.path.basename(c).startswith(rnd)]
    if ticks_per_launch is not None:
        for c in cands:
            try:
                if int(json.load(open(c)).get("ticks_per_launch", -1)) == int(ticks_per_launch):
                    return c
            except Exception:
                pass
    plain = [c for c in cands if os.path.basename(c) == f"{rnd}_config{config}_pmc_summary.json"]
    return plain[0] if plain else cands[-1]


def host_cpu_description():
    model, threads = "?", os.cpu_count() or 1
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in txt.splitlines():
            if line.startswith("Model name:"):
                model = line.split(":", 1)[1].strip()
    except Exception:
        pass
    return model, threads


def cpu_baseline(pkg, routes, batch, T, seconds):
    """The oracle (C port of the reference path) running the SAME closed loop on the host cores: MPC.step with the carried
    warm start -> Simulation.step -> goal test / respawn (oracle/mpc_oracle.c orc_closed_loop), OpenMP over egos, on every
    core this process may run on; a bounded sample (~`seconds` s)."""
    import numpy as np
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_py as O
    O.build()
    p = O.make_params(T=T)
    cx, cy, cyaw, off = pkg.synth.pack_paths(routes)
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, cores)
    model, hw_threads = host_cpu_description()
    B = batch.x0.shape[0]

    def take(n):   # the first n egos of the batch, tiled when n > B
        idx = np.arange(n) % B
        sub = pkg.synth.EgoBatch(x0=batch.x0[idx], path_id=batch.path_id[idx], path_len=batch.path_len[idx],
                                 target_ind=batch.target_ind[idx], speed=batch.speed[idx], oa=batch.oa[idx], od=batch.od[idx])
        return O.loop_state_from_batch(sub, T)

    def run(n_egos, nthreads, budget, ticks_per_call):
        st = take(n_egos)
        n, t0 = 0, time.perf_counter()
        while True:
            O.closed_loop(p, st, cx, cy, cyaw, off, ticks_per_call, max_age=400, n_threads=nthreads, record=False)
            n += n_egos * ticks_per_call
            dt = time.perf_counter() - t0
            if dt >= budget:
                return n / dt, n
    one, n1 = run(min(B, 32), 1, seconds * 0.3, 2)
    # "all cores": the affinity mask can be wider than the CPU time this process is actually given (a GPU box hands one GPU's
    # job a share of a 256-thread host), so a short probe picks the thread count that delivers the most, and that count is reported
    cands = sorted({min(c, cores) for c in (8, 16, 32, 64, 128, cores)})
    probe = {}
    for c in cands:
        probe[c], _ = run(max(B, 8 * c), c, seconds * 0.25 / len(cands), 2)
    best = max(probe, key=probe.get)
    n_all = max(B, 8 * best)        # every thread gets whole egos to loop over
    allc, na = run(n_all, best, seconds * 0.45, 4)
    return {"value": allc, "unit": "MPC steps/s", "cores": best, "kind": "port",
            "value_1core": one, "host_cpu": model, "host_hw_threads": hw_threads, "affinity_cores": cores,
            "thread_probe": {str(k): round(v) for k, v in probe.items()},
            "sample": (f"closed loop (MPC.step with carried warm start -> plant -> goal/respawn) of the same synthetic egos at T={T}: "
                       f"{n1} steps of {min(B, 32)} egos on 1 core, {na} steps of {n_all} egos on {best} threads (OpenMP over egos; the "
                       f"thread count that delivered most in a short probe over {cands}); "
                       "C port of the reference path (oracle/mpc_oracle.c, exact active-set QP); cvxpy/ECOS unavailable offline")}


def planner_cpu_baseline(pi, max_nodes=400, budget_s=8.0):
    """The numpy restatement of the reference's planner (oracle/planner_oracle.py) on the same route queries, one core; routes
    whose search the GPU needed more than `max_nodes` expansions for are left out (numpy spends ~3 ms per expansion), and the
    sample stops after `budget_s` seconds."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import planner_oracle as PO
    import importlib
    PL = importlib.import_module("av-simulation-at-intersections_amd").planner
    mps = PO.make_motion_primitives()
    radius, centres = PL.car_circles()
    done, nodes, t0 = 0, 0, time.perf_counter()
    for q, n in zip(pi["queries"], pi["expanded"]):
        if n > max_nodes:
            continue
        orc = PO.PlannerOracle(q.start, q.goal, q.goal_box, q.tol, q.obstacles, mps, centres, radius)
        orc.run()
        done += 1
        nodes += orc.n_expanded
        if time.perf_counter() - t0 > budget_s:
            break
    return {"routes": done, "wall_ms": (time.perf_counter() - t0) * 1e3, "expanded_nodes": nodes, "cores": 1, "kind": "port",
            "sample": f"the first {done} of the workload's routes that need <= {max_nodes} expansions"}


if __name__ == "__main__":
    main()
