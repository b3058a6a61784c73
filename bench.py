#!/usr/bin/env python3
"""bench.py -- MPC receding-horizon steps/s of the HIP path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 256 egos per GPU, kinematic bicycle, horizon T = 20, nu = 2, fp64,
synthetic random-init egos on the 12 synthetic intersection routes, CLOSED LOOP: one "step" = one tick =
one MPC.step for every ego (jsim_mpc_step) + the loop bookkeeping (plant update, history record, respawn
of finished egos; jsim_loop_advance), all resident in HBM.  Weak scaling: every rank owns 256 egos, no
collective on the solve path; the recorded controls are all-gathered once at the end (RCCL).

Prints ONE JSON line (rank 0).  `value` = egos x ticks / wall time (max over ranks).
`roofline`  : the dominant kernel (mpc_step_kernel) against the fp64 peak -- the path is compute/latency
              bound; `roofline_hbm` carries the algorithmic-bytes-vs-HBM figure BASELINE.json asks for.
`cpu_baseline`: the CPU oracle (a C port of the reference path; the reference's cvxpy/ECOS stack cannot be
              installed) timed on this box's host cores on the same synthetic batch.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == fp64 matrix peak
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes_per_step(T, w=8):
    """w(15T+16) + 16 + T bytes per ego per MPC step."""
    return w * (15 * T + 16) + 16 + T


def algorithmic_flops_per_step(T, n_iter):
    """16T^3 (Hessian GEMM) + 8T^3/3 (Cholesky) + 40T^2 n_iter (active set) + 32T^2 + 60T."""
    return 16 * T ** 3 + 8 * T ** 3 / 3 + 40 * T ** 2 * n_iter + 32 * T ** 2 + 60 * T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256, help="egos per GPU")
    ap.add_argument("--horizon", type=int, default=20)
    ap.add_argument("--mode", choices=("fused", "graph", "eager"), default="fused",
                    help="fused: K ticks per launch inside the kernel (default); graph: one launch pair per tick "
                         "replayed from a hipGraph; eager: one launch pair per tick from Python")
    ap.add_argument("--ticks-per-launch", type=int, default=0,
                    help="fused mode: closed-loop ticks per kernel launch (0 = all K ticks of the timed region in one launch; "
                         "egos only wait for each other at launch boundaries)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
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
    S = pkg.synth
    T, B, K, W = args.horizon, args.batch, args.steps, args.warmup

    routes = S.make_route_table()
    for r in routes:
        S.smooth_yaw_inplace(r[:, 2])
    batch = S.make_ego_batch(routes, B, T, seed=1 + rank, truncate=False)
    eng = pkg.BatchedMPC(routes, batch.path_id, dl=S.DL, T=T, speed=batch.speed, device=device, smooth=False)
    eng.load_state(batch.target_ind, batch.oa, batch.od, batch.path_len)
    x0 = torch.from_numpy(batch.x0).to(device)
    loop = pkg.ClosedLoop(eng, x0, hist_cap=K + W + 8, max_age=400)

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---- warm-up (untimed)
    for _ in range(W):
        loop.tick()
    torch.cuda.synchronize(device)
    if world > 1:   # the job's one collective, run once untimed: communicator set-up and buffer registration are not the path
        _ = pkg.sharding.gather_rows(loop.hist[:max(W, 1)].permute(1, 0, 2).contiguous(), B * world)
        torch.cuda.synchronize(device)

    mode = args.mode
    chunk = next(c for c in (50, 25, 20, 10, 5, 4, 2, 1) if K % c == 0)
    if mode == "fused":
        chunk = args.ticks_per_launch if args.ticks_per_launch > 0 else K
        if K % chunk:
            raise SystemExit(f"--ticks-per-launch {chunk} must divide --steps {K}")
    if mode == "graph":
        loop.capture(chunk)   # (capture runs one extra untimed tick)
    elif mode == "fused":
        loop.run(1)           # untimed: first use of the entry point

    # ---- timed region: exactly K ticks
    n_iter_sum = torch.zeros((), dtype=torch.float64, device=device)
    fused_evs = []
    sync_all()
    t0 = time.perf_counter()
    if mode == "fused":       # closed loop on the device: `chunk` ticks per launch, egos never wait for each other
        for _ in range(K // chunk):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record(); loop.run(chunk); eb.record()
            fused_evs.append((ea, eb))
    elif mode == "graph":
        for _ in range(K // chunk):
            loop.replay()
    else:
        for _ in range(K):
            loop.tick()
    if world > 1:   # the only exchange of the job: gather every rank's recorded controls (RCCL all-gather)
        hist_local = loop.hist[:K].permute(1, 0, 2).contiguous()          # [B, K, 2]
        hist_all = pkg.sharding.gather_rows(hist_local, B * world)
        assert hist_all.shape[0] == B * world
    sync_all()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cpu" if rehearsal else device)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    # ---- dominant-kernel duration: HIP events on the launch stream around K more launches of mpc_step_kernel
    # (same closed-loop states keep evolving; events bracket only the MPC kernel, not the bookkeeping kernel)
    KE = min(K, 200)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(KE)]
    n_iter_sum.zero_()
    ok_sum = 0
    torch.cuda.synchronize(device)
    for a, b in evs:
        a.record()
        eng.solve(loop.x0)
        b.record()
        n_iter_sum += eng.n_iter.sum()
        _ = pkg._cabi.check(eng.lib.jsim_loop_advance(
            eng._ctx, eng.B, loop.x0.data_ptr(), eng.oa.data_ptr(), eng.od.data_ptr(), eng.status.data_ptr(),
            eng.di_ai.data_ptr(), eng.target_ind.data_ptr(), eng.path_id.data_ptr(), eng.path_len.data_ptr(),
            loop.x0_spawn.data_ptr(), loop.target_spawn.data_ptr(), loop.age.data_ptr(), loop.max_age, None, None, 0,
            loop.n_respawn.data_ptr(), eng._stream()), eng._ctx)
    torch.cuda.synchronize(device)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))   # one single-tick launch
    launches, ticks_per_launch = KE, 1
    if mode == "fused":   # the dominant launch of the timed region IS the fused kernel: HIP events around each one
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in fused_evs]))
        launches, ticks_per_launch = len(fused_evs), chunk
    mean_iter = float(n_iter_sum.item()) / (KE * B)
    n_fail = int((eng.status != 0).sum().item())

    steps_total = B * world * K
    value = steps_total / elapsed

    if rank == 0:
        flops = algorithmic_flops_per_step(T, mean_iter) * B * ticks_per_launch
        nbytes = algorithmic_bytes_per_step(T) * B * ticks_per_launch
        ach_tf = flops / (kern_ms * 1e-3) / 1e12
        ach_gbs = nbytes / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(REPO, "profiles", "r01_bench_fused_pmc_summary.json")
        if mode == "fused" and T == 20 and B == 256 and os.path.exists(pmc_path):
            # HBM bytes of one fused launch from the rocprofv3 PMC passes of this same command (profiles/r01_SUMMARY.txt):
            # FETCH_SIZE doubled (gfx950 tallies 64 B per 128-B request), WRITE_SIZE as is, both in KiB; scaled to this chunk
            pmc = json.load(open(pmc_path))
            traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0 * (ticks_per_launch / float(pmc.get("ticks_per_launch", 50)))
        out = {
            "metric": "MPC steps/sec (batch x horizon) at N=20 nu=2",
            "value": value, "unit": "MPC steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{B}-ego batch per GPU, kinematic bicycle, horizon N={T}, nu=2, fp64, closed loop "
                                   f"({'BASELINE.json configs[1]' if (B, T) == (256, 20) else 'not the headline config'})", "egos_per_gpu": B, "horizon": T,
                       "launch": {"fused": f"fused closed loop, {chunk} ticks per launch", "graph": "hipGraph",
                                  "eager": "eager"}[mode], "parallelism": f"ego-shard x{world}",
                       "mean_active_set_iters": round(mean_iter, 2), "failed_egos_last_tick": n_fail,
                       "respawns": int(loop.n_respawn.item())},
            "roofline": {"bound": "mfma", "achieved": ach_tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach_tf / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": (f"mpc_step_reg_kernel<{T}, false>" if T in (13, 20, 30) else
                                    f"mpc_step_reg2_kernel<{T}, false>" if T == 40 else "mpc_step_kernel"), "kernel_ms": kern_ms,
                         "ticks_per_launch": ticks_per_launch,
                         "algorithmic_flops_per_launch": flops,
                         "note": ("fp64 vector/matrix peak; latency-bound: " +
                                  (f"{2 if T == 40 else 1} wave(s) per ego, {B} egos on 256 CUs (1024 SIMDs)"))},
            "roofline_hbm": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, routes, batch, T, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(pkg, routes, batch, T, seconds):
    """The oracle (C port of the reference path) on the host cores, same synthetic batch (tick 0)."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle_py as O
    O.build()
    p = O.make_params(T=T)
    cx, cy, cyaw, off = pkg.synth.pack_paths(routes)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, 16))   # the GPU box's CPU share for one GPU is 16 cores
    # tile the batch so that every thread has real work (256 egos over 16 threads would be one malloc-bound
    # scheduling quantum each)
    rep = 8
    tile = lambda a: np.concatenate([a] * rep, axis=0)
    big = pkg.synth.EgoBatch(x0=tile(batch.x0), path_id=tile(batch.path_id), path_len=tile(batch.path_len),
                             target_ind=tile(batch.target_ind), speed=tile(batch.speed), oa=tile(batch.oa),
                             od=tile(batch.od))
    B = batch.x0.shape[0]

    def run(bt, nthreads, budget):
        n, t0 = 0, time.perf_counter()
        while True:
            O.mpc_step_batch(p, bt.x0, bt.path_id, bt.path_len, bt.speed, cx, cy, cyaw, off, bt.target_ind, bt.oa,
                             bt.od, n_threads=nthreads)
            n += bt.x0.shape[0]
            dt = time.perf_counter() - t0
            if dt >= budget:
                return n / dt
    one = run(batch, 1, seconds * 0.4)
    allc = run(big, cores, seconds * 0.6)
    return {"value": allc, "unit": "MPC steps/s", "cores": cores, "kind": "port",
            "value_1core": one,
            "sample": f"tick-0 batch of {B} egos (T={T}; x{rep} tiled for the {cores}-thread run) re-solved for ~{seconds:.0f} s: C port of the reference path "
                      f"(oracle/mpc_oracle.c, exact active-set QP), OpenMP over egos; cvxpy/ECOS unavailable offline"}


if __name__ == "__main__":
    main()
