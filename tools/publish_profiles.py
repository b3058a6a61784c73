#!/usr/bin/env python3
"""Copy what tools/collect_profile.sh left under gpurun_out/r<R>_c{2,3,4,5}/ into profiles/ (tracked): the per-launch PMC /
kernel-trace summary (profiles/r0<R>_config{N}_pmc_summary.json -- bench.py reads `roofline.traffic` from the latest one),
rocprofv3's kernel-stats CSV of the traced run, and a readable digest (profiles/r0<R>_SUMMARY.txt).   usage: publish_profiles.py [round=3]"""
import json
import os
import shutil
import sys

R = int(sys.argv[1]) if len(sys.argv) > 1 else 3

os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lines = [f"Round {R} profile collection (tools/collect_profile.sh: one rocprofv3 --kernel-trace --stats run and four separate --pmc passes of",
         "`python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-respawn-start [--config N]` on one MI355X; routes planned on the GPU; launch order on).",
         "Per fused launch of the dominant kernel.  FETCH_SIZE / WRITE_SIZE in KiB as the counters report them; HBM bytes = (2 x FETCH + WRITE) KiB",
         "(gfx950 tallies 64 B per 128-B fetch request, MI355X_MICROARCH.md).  bench kernel_ms = HIP events inside bench.py in the traced run.", ""]
for c, t in ((2, f"r{R}_c2"), (3, f"r{R}_c3"), (4, f"r{R}_c4"), (5, f"r{R}_c5")):
    src = f"gpurun_out/{t}/summary.json"
    if not os.path.exists(src):
        continue
    d = json.load(open(src))
    json.dump(d, open(f"profiles/r{R:02d}_config{c}_pmc_summary.json", "w"), indent=1)
    shutil.copy(f"gpurun_out/{t}/trace/runc_kernel_stats.csv", f"profiles/r{R:02d}_config{c}_kernel_stats.csv")
    b = d["bench_line_under_rocprof"]
    r = b["roofline"]
    tpl = d["ticks_per_launch"]
    hbm = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
    egos = b["config"]["egos_per_gpu"]
    lines.append(f"config {c}: {b['config']['workload']}")
    lines.append(f"  kernel {d['kernel']}   (library at commit {d.get('commit', '?')})")
    cf = b["config"]
    if cf.get("straggler"):
        lines.append(f"  iterations per step (timed launches' own counters): mean {cf['mean_active_set_iters']}, slowest ego {cf['straggler']['max_ego_iters_per_tick']} "
                     f"= {cf['straggler']['slowest_over_mean']} x the mean")
    lines.append(f"  value under rocprof {b['value']:.0f} MPC steps/s; {tpl} ticks per launch; kernel-trace launch {d['fused_launch_ms_kernel_trace']:.3f} ms "
                 f"over {d['fused_launches']} launch(es); bench.py HIP events {r['kernel_ms']:.3f} ms")
    lines.append(f"  HBM traffic per launch {hbm / 1e6:.2f} MB (FETCH {d['FETCH_SIZE']:.0f} KiB, WRITE {d['WRITE_SIZE']:.0f} KiB) = {hbm / (egos * tpl):.0f} B per ego-tick; "
                 f"algorithmic {r['roofline_hbm']['algorithmic_bytes_per_launch'] / 1e6:.2f} MB" if "roofline_hbm" in r and "algorithmic_bytes_per_launch" in r["roofline_hbm"]
                 else f"  HBM traffic per launch {hbm / 1e6:.2f} MB (FETCH {d['FETCH_SIZE']:.0f} KiB, WRITE {d['WRITE_SIZE']:.0f} KiB) = {hbm / (egos * tpl):.0f} B per ego-tick")
    lines.append(f"  roofline against the fp64 peak: achieved {r['achieved']:.3f} {r['unit']} of {r['peak']} -> frac {r['frac']:.4f}")
    lines.append(f"  SQ: INSTS_VALU {d['SQ_INSTS_VALU']:.3e} (FMA_F64 {d['SQ_INSTS_VALU_FMA_F64']:.3e}, MFMA_F64 {d['SQ_INSTS_VALU_MFMA_F64']:.3e}), INSTS_LDS {d['SQ_INSTS_LDS']:.3e}, "
                 f"INSTS_SALU {d['SQ_INSTS_SALU']:.3e}, LDS_BANK_CONFLICT {d['SQ_LDS_BANK_CONFLICT']:.3e}")
    lines.append(f"      WAVE_CYCLES {d['SQ_WAVE_CYCLES']:.3e}, BUSY_CYCLES {d['SQ_BUSY_CYCLES']:.3e}, WAIT_ANY {d['SQ_WAIT_ANY']:.3e}, WAIT_INST_ANY {d['SQ_WAIT_INST_ANY']:.3e}, "
                 f"ACTIVE_INST_VALU {d['SQ_ACTIVE_INST_VALU']:.3e}, GRBM_GUI_ACTIVE {d['GRBM_GUI_ACTIVE']:.3e}")
    lines.append(f"      VALU instructions per wave-cycle {d['SQ_INSTS_VALU'] / d['SQ_WAVE_CYCLES']:.3f}; WAIT_ANY / WAVE_CYCLES {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.3f}")
    for k in d["kernel_stats"]:
        lines.append(f"    {k['Name'][:70]:70s} calls {k['Calls']:>4s} avg {float(k['AverageNs']) / 1e3:10.1f} us  {float(k['Percentage']):6.2f} %")
    lines.append("")
open(f"profiles/r{R:02d}_SUMMARY.txt", "w").write("\n".join(lines))
print("\n".join(lines))
