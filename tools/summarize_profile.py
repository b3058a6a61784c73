#!/usr/bin/env python3
"""Turn profiles/r01_bench_fused_pmc_summary.json (+ the bench line of the profiled run) into profiles/r01_SUMMARY.txt."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(REPO, "profiles")
p = json.load(open(os.path.join(P, "r01_bench_fused_pmc_summary.json")))
bench = json.loads(open(os.path.join(P, "r01_bench_fused_under_rocprof.json")).read())
launch_ms = p["fused_launch_ms_kernel_trace"]
egosteps = 50 * 256
traffic = (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
clk_ghz = p["GRBM_GUI_ACTIVE"] / 8 / launch_ms / 1e6
txt = f"""Round 1 -- rocprofv3 evidence for the default bench (python bench.py --steps 100 --warmup 10, fused mode)
=====================================================================================================
Commands (each its own run, from /tmp with TMPDIR=/tmp, program after `--`):
  rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc FETCH_SIZE ...            (TCC: 3 slots)
  rocprofv3 --kernel-trace --pmc WRITE_SIZE ...            (TCC: 2 slots)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_VALU ...
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE ...

Dominant kernel: mpc_step_reg_kernel<20>(KP, TickP); the timed region consists of 2 launches of 50 ticks x 256 egos
(the other 111 dispatches of the same kernel in r01_bench_fused_kernel_stats.csv are single-tick launches: 10 warm-up
ticks, 1 untimed first use, 100 event-bracketed single ticks after the timed region -- avg {p['single_tick_launch_ms_kernel_trace']:.3f} ms each).

Fused launch (50 ticks, 256 waves, one per CU):
  duration (kernel trace)            {launch_ms:.2f} ms   (bench.py HIP events in the same run: {bench['roofline']['kernel_ms']:.2f} ms)
  => 12,800 MPC steps per launch; bench line of the profiled run: {bench['value'] / 1e6:.2f} M steps/s
  FETCH_SIZE                         {p['FETCH_SIZE']:.0f} KiB   (gfx950 counts 64 B per 128-B request on wide coalesced reads -> x2)
  WRITE_SIZE                         {p['WRITE_SIZE']:.0f} KiB
  HBM traffic (2*FETCH + WRITE)      {traffic / 1e6:.2f} MB per launch = {traffic / egosteps:.0f} B per MPC step
                                     (algorithmic figure of SURVEY 8d: 2564 B/step; state stays in registers across the
                                      50 ticks and the 200-KB path table is cache-resident, so less than that reaches HBM)
  achieved HBM rate                  {traffic / launch_ms / 1e6:.2f} GB/s  = {traffic / launch_ms / 1e6 / 8000 * 100:.4f} % of 8 TB/s   (not the binding resource)
  SQ_INSTS_VALU                      {p['SQ_INSTS_VALU'] / egosteps:.0f} wave-instructions per MPC step
  SQ_INSTS_VALU_FMA_F64              {p['SQ_INSTS_VALU_FMA_F64'] / egosteps:.0f} per MPC step  -> {p['SQ_INSTS_VALU_FMA_F64'] * 128 / launch_ms / 1e9:.2f} TFLOP/s fp64 FMA executed (incl. idle lanes)
  SQ_INSTS_VALU_MFMA_F64             {p['SQ_INSTS_VALU_MFMA_F64'] / egosteps:.0f} per MPC step (v_mfma_f64_16x16x4_f64; = the 56 non-zero (tile, time-step) pairs at T=20)
  SQ_VALU_MFMA_BUSY_CYCLES           {p['SQ_VALU_MFMA_BUSY_CYCLES']:.3g}  = {p['SQ_VALU_MFMA_BUSY_CYCLES'] / p['SQ_INSTS_VALU_MFMA_F64']:.0f} cycles per MFMA
  GRBM_GUI_ACTIVE / 8 XCDs           {p['GRBM_GUI_ACTIVE'] / 8:.3g} cycles -> {clk_ghz:.2f} GHz effective clock
  MFMA utilisation                   {p['SQ_VALU_MFMA_BUSY_CYCLES'] / (p['GRBM_GUI_ACTIVE'] / 8 * 1024) * 100:.2f} % of all 1024 SIMDs ({p['SQ_VALU_MFMA_BUSY_CYCLES'] / (p['GRBM_GUI_ACTIVE'] / 8 * 256) * 100:.2f} % of the 256 SIMDs that hold a wave)
  SQ_WAIT_ANY / SQ_WAVE_CYCLES       {p['SQ_WAIT_ANY'] / p['SQ_WAVE_CYCLES'] * 100:.0f} % of wave lifetime parked on s_waitcnt (LDS / scalar loads), issue-stalled {p['SQ_WAIT_INST_ANY'] / p['SQ_WAVE_CYCLES'] * 100:.0f} %
  SQ_INSTS_LDS                       {p['SQ_INSTS_LDS'] / egosteps:.0f} per MPC step, bank-conflict cycles {p['SQ_LDS_BANK_CONFLICT'] / egosteps:.0f} per step
  SQ_WAVES                           256 (one wave per CU: the shape is latency-bound by construction)

Reading: the path moves ~{traffic / egosteps:.0f} B/step and executes ~{p['SQ_INSTS_VALU'] / egosteps / 1000:.0f}k wave instructions/step on ONE wave per CU; it is bound by the
dependent-issue latency of a single wavefront (SURVEY D6), not by HBM and not by MFMA throughput.

History of the same measurement within round 1 (bench.py, 1 GPU, 256 egos, T = 20):
  v1 LDS-resident kernel, one launch per tick              0.37 M steps/s   (0.67 ms per single-tick launch)   r01_v1_lds_*
  register-resident kernel, one launch per tick            0.82 M           (0.38 ms)
  + fused closed loop (50 ticks per launch)                1.54 M           (26k VALU wave-instr per step)
  + steepest-edge entering rule (-35 % iterations)         2.00 M
  + burst LDS reads behind sched barriers                  2.24 M           (18k VALU wave-instr per step)
Files: r01_bench_fused_kernel_stats.csv (rocprofv3 --stats), r01_bench_fused_pmc_summary.json (per-launch counter means),
r01_bench_fused_under_rocprof.json (the bench line printed in the profiled run), r01_bench_default.json (plain `python bench.py`),
r01_reg_kernel_phase_stamps_T20.txt (diagnostic -DJSIM_STAMPS build: per-phase / per-section cycle shares; that build spills
and its absolute times -- the `u0` phase in particular -- are not those of the shipped kernel), r01_v1_lds_* (first kernel).
"""
open(os.path.join(P, "r01_SUMMARY.txt"), "w").write(txt)
print(txt)
