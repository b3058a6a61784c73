#!/usr/bin/env python3
"""profiles/ from the rocprofv3 runs of tools/collect_profile.sh:
    summarize_profile.py TAG_default [TAG_other ...]
TAG_default = the run of the default bench (256 egos, T = 20): writes r01_bench_fused_pmc_summary.json (read by bench.py
for `roofline.traffic`), r01_bench_fused_kernel_stats.csv, r01_bench_fused_under_rocprof.json and r01_SUMMARY.txt;
every other TAG adds r01_<TAG>_pmc_summary.json and a paragraph."""
import json
import os
import shutil
import glob
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(REPO, "profiles")
tags = sys.argv[1:]


def load(tag):
    return json.load(open(os.path.join(REPO, "gpurun_out", tag, "summary.json")))


def para(p, title):
    b = p["bench_line_under_rocprof"]
    B, T, tpl = b["config"]["egos_per_gpu"], b["config"]["horizon"], b["roofline"]["ticks_per_launch"]
    steps = B * tpl
    ms = p["fused_launch_ms_kernel_trace"]
    traffic = (2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024
    clk = p["GRBM_GUI_ACTIVE"] / 8 / ms / 1e6
    waves = p["SQ_WAVES"]
    simds = min(waves, 1024)
    return f"""{title}
  kernel                             {p['kernel']}   ({p['fused_launches']} fused launches of {tpl} ticks x {B} egos in the timed region)
  duration (kernel trace)            {ms:.3f} ms   (bench.py HIP events in the same run: {b['roofline']['kernel_ms']:.3f} ms)
  => {steps} MPC steps per launch; bench line of the profiled run: {b['value'] / 1e6:.2f} M steps/s, mean {b['config']['mean_active_set_iters']} active-set iterations
  FETCH_SIZE / WRITE_SIZE            {p['FETCH_SIZE']:.0f} / {p['WRITE_SIZE']:.0f} KiB   (gfx950 counts 64 B per 128-B request on wide coalesced reads -> FETCH x2)
  HBM traffic (2*FETCH + WRITE)      {traffic / 1e6:.2f} MB per launch = {traffic / steps:.0f} B per MPC step   (algorithmic, SURVEY 8d: {8 * (15 * T + 16) + 16 + T} B/step)
  achieved HBM rate                  {traffic / ms / 1e6:.2f} GB/s = {traffic / ms / 1e6 / 8000 * 100:.4f} % of 8 TB/s   (not the binding resource)
  SQ_INSTS_VALU / _SALU / _LDS       {p['SQ_INSTS_VALU'] / steps:.0f} / {p['SQ_INSTS_SALU'] / steps:.0f} / {p['SQ_INSTS_LDS'] / steps:.0f} wave-instructions per MPC step
  SQ_INSTS_VALU_FMA_F64              {p['SQ_INSTS_VALU_FMA_F64'] / steps:.0f} per MPC step -> {p['SQ_INSTS_VALU_FMA_F64'] * 128 / ms / 1e9:.2f} TFLOP/s fp64 FMA executed (incl. idle lanes)
  SQ_INSTS_VALU_MFMA_F64             {p['SQ_INSTS_VALU_MFMA_F64'] / steps:.0f} per MPC step (v_mfma_f64_16x16x4_f64), {p['SQ_VALU_MFMA_BUSY_CYCLES'] / p['SQ_INSTS_VALU_MFMA_F64']:.0f} busy cycles each
  MFMA utilisation                   {p['SQ_VALU_MFMA_BUSY_CYCLES'] / (p['GRBM_GUI_ACTIVE'] / 8 * 1024) * 100:.2f} % of all 1024 SIMDs
  GRBM_GUI_ACTIVE / 8 XCDs           {p['GRBM_GUI_ACTIVE'] / 8:.3g} cycles -> {clk:.2f} GHz effective clock
  wave lifetime                      VALU busy {p['SQ_ACTIVE_INST_VALU'] / p['SQ_WAVE_CYCLES'] * 100 * (4 if False else 1):.0f} % (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES), waiting on s_waitcnt {p['SQ_WAIT_ANY'] / p['SQ_WAVE_CYCLES'] * 100:.0f} %, issue-stalled {p['SQ_WAIT_INST_ANY'] / p['SQ_WAVE_CYCLES'] * 100:.0f} %
  SQ_WAVES                           {waves:.0f} ({waves / 256:.0f} per CU over the launch); LDS bank-conflict cycles {p['SQ_LDS_BANK_CONFLICT'] / steps:.0f} per step
"""


main = load(tags[0])
keep = {k: v for k, v in main.items() if k not in ("bench_line_under_rocprof", "kernel_stats")}
json.dump(keep, open(os.path.join(P, "r01_bench_fused_pmc_summary.json"), "w"), indent=1)
json.dump(main["bench_line_under_rocprof"], open(os.path.join(P, "r01_bench_fused_under_rocprof.json"), "w"))
st = glob.glob(os.path.join(REPO, "gpurun_out", tags[0], "trace", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(P, "r01_bench_fused_kernel_stats.csv"))
txt = """Round 1 -- rocprofv3 evidence (tools/collect_profile.sh: one --kernel-trace --stats run and four separate --pmc passes
per configuration, from /tmp with TMPDIR=/tmp, `python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline [...]` after `--`)
=====================================================================================================================
"""
txt += para(main, "Default bench: 256 egos, T = 20, fused closed loop (BASELINE.json configs[1])")
txt += """
Reading: one wave per CU, ~13k VALU wave-instructions per step at ~5.3 cycles each (tools/ubench/issue_cost.hip: every VALU
instruction of a lone wave issues in >= 5.3 cycles, a dependent FMA chain in 8.3, FMA->readlane->FMA in 26, a uniform
branch in ~40, a ds_write_b128 in 25): the path is bound by the issue latency of a single wavefront (SURVEY D6), not by
HBM and not by MFMA throughput.  The launch time is set by the slowest of the 256 egos (tools/wave_span.py: the mean
step takes 57.3k + 5.67k * n_iter cycles, the slowest ego averages 21 iterations against a fleet mean of 10.9).
"""
for t in tags[1:]:
    p = load(t)
    json.dump({k: v for k, v in p.items() if k not in ("bench_line_under_rocprof", "kernel_stats")},
              open(os.path.join(P, f"r01_{t}_pmc_summary.json"), "w"), indent=1)
    st = glob.glob(os.path.join(REPO, "gpurun_out", t, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(P, f"r01_{t}_kernel_stats.csv"))
    b = p["bench_line_under_rocprof"]
    how = "two wavefronts per ego" if b['config']['horizon'] > 31 else "one wavefront per ego, virtual speed rows, 4 egos per CU"
    txt += "\n" + para(p, f"{b['config']['egos_per_gpu']} egos, T = {b['config']['horizon']} ({how})")
txt += """
History of the default bench within round 1 (bench.py, 1 GPU, 256 egos, T = 20):
  v1 LDS-resident kernel, one launch per tick                      0.37 M steps/s   (0.67 ms per single-tick launch)   r01_v1_lds_*
  register-resident kernel, one launch per tick                    0.82 M
  + fused closed loop (50 ticks per launch)                        1.54 M           (26k VALU wave-instr per step)
  + steepest-edge entering rule (-35 % iterations)                 2.00 M
  + burst LDS reads behind sched barriers                          2.24 M           (18k VALU wave-instr per step)
  + packed 64-bit arg-max / arg-min, pair candidates, L prefetch   2.38 M
  + arithmetic pinned where written (no AGPR round trips)          2.60 M           (15k VALU wave-instr per step)
  + leaner MFMA operand generation, column-oriented factorisation
    with look-ahead, pipelined forward substitution                2.78 M           (13k VALU wave-instr per step)
  + two-wave kernel's constants read as LDS / global (not FLAT), single-pass publish   2.9 M   (12.8k VALU wave-instr per step)
Horizons 30 / 40 (4096 egos): LDS-resident kernel 0.84 / 0.18 M steps/s -> two-wave register kernel 3.2 / 1.4 M
-> T = 30 in ONE wave per ego (speed rows virtual, 40 KB of LDS: four egos per CU) 5.3 M.
Other shapes, same build (r01_other_configs.txt: egos, T, steps/s, ms per tick, mean iterations, fp64 roofline fraction).
Files: r01_bench_fused_kernel_stats.csv (rocprofv3 --stats), r01_bench_fused_pmc_summary.json (per-launch counter means; bench.py
reads FETCH_SIZE / WRITE_SIZE from it for `roofline.traffic`), r01_bench_fused_under_rocprof.json (bench line of the traced run),
r01_bench_default.json (plain `python bench.py`), r01_<tag>_* (other configurations), r01_ubench_issue_cost.txt (instruction
issue costs of a lone wave), r01_wave_span_T20.txt (per-ego step time vs iteration count), r01_reg_kernel_phase_stamps_T*.txt
(diagnostic -DJSIM_STAMPS builds: per-phase cycle shares; stamps perturb, read shares not times), r01_v1_lds_* (first kernel).
"""
open(os.path.join(P, "r01_SUMMARY.txt"), "w").write(txt)
print(txt)
