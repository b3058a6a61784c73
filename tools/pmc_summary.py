#!/usr/bin/env python3
"""Condense gpurun_out/TAG/{trace,fetch,write,sq1,sq2} (tools/collect_profile.sh) into
gpurun_out/TAG/summary.json: per-launch means of every counter over the FUSED launches of the dominant kernel (the ones
with the longest duration), their kernel-trace duration, and the bench line printed in the traced run."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
root = os.path.join(REPO, "gpurun_out", tag)


def rows(sub, pat):
    f = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    if not f:
        return []
    return list(csv.DictReader(open(f[0])))


tr = [r for r in rows("trace", "*kernel_trace.csv") if "mpc_step" in r["Kernel_Name"]]
dur = defaultdict(list)
for r in tr:
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
name = max(dur, key=lambda k: max(dur[k]))   # the kernel of the timed region: the one with the longest single launch
d = sorted(dur[name])
fused = [x for x in d if x > 0.5 * d[-1]]          # the multi-tick launches
single = [x for x in d if x <= 0.5 * d[-1]]
out = {"commit": os.environ.get("JSIM_COMMIT", "?"),   # the library's source state (the GPU box has no .git: handed in by the caller)
       "kernel": name, "dispatches": len(d), "fused_launches": len(fused),
       "fused_launch_ms_kernel_trace": sum(fused) / len(fused),
       "single_tick_launch_ms_kernel_trace": (sum(single) / len(single)) if single else None}
for sub in ("fetch", "write", "sq1", "sq2"):
    cr = [r for r in rows(sub, "*counter_collection.csv") if r["Kernel_Name"] == name]
    per = defaultdict(dict)
    for r in cr:
        per[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
        per[r["Dispatch_Id"]]["_dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not per:
        continue
    mx = max(v["_dur"] for v in per.values())
    sel = [v for v in per.values() if v["_dur"] > 0.5 * mx]
    for k in sel[0]:
        if k != "_dur":
            out[k] = sum(v[k] for v in sel) / len(sel)
for sub in ("trace",):
    log = os.path.join(root, sub + ".log")
    for line in open(log):
        if line.startswith('{"metric"'):
            out["bench_line_under_rocprof"] = json.loads(line)
            out["ticks_per_launch"] = out["bench_line_under_rocprof"]["roofline"]["ticks_per_launch"]
st = rows("trace", "*kernel_stats.csv")
out["kernel_stats"] = [r for r in st if any(k in r.get("Name", "") for k in ("mpc_step", "loop_", "tick_", "plan_astar", "launch_order", "obstacle_"))]
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("bench_line_under_rocprof", "kernel_stats")}, indent=1))
