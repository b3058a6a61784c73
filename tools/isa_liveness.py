#!/usr/bin/env python3
"""Diagnostic: VGPR/AGPR liveness of one kernel in hipcc's -S output (gfx950).  Prints the number of live vector
registers at every basic-block entry of the largest loops, to see what an unrolled kernel keeps alive where.
usage: isa_liveness.py file.s kernel_symbol_prefix"""
import re
import sys
from collections import defaultdict

src, sym = sys.argv[1], sys.argv[2]
L = open(src).read().split("\n")
a = [i for i, l in enumerate(L) if l.startswith(sym)][0]
b = [i for i, l in enumerate(L) if l.startswith(".Lfunc_end") and i > a][0]
lines = [l.split(";")[0].rstrip() for l in L[a + 1:b]]
lines = [l for l in lines if l.strip()]

def regs(tok):
    tok = tok.strip().lstrip("-").strip("|")
    m = re.match(r"^([va])\[(\d+):(\d+)\]$", tok)
    if m:
        return [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.match(r"^([va])(\d+)$", tok)
    if m:
        return [(m.group(1), int(m.group(2)))]
    return []

NODEF = ("ds_write", "global_store", "scratch_store", "flat_store", "buffer_store", "s_", "v_cmp", "v_readlane", "v_readfirstlane",
         "global_atomic", "ds_or", "ds_add", "v_cmpx")
blocks, cur, label_of = [], None, {}
for l in lines:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = {"label": m.group(1), "ins": [], "succ": []}
        label_of[m.group(1)] = len(blocks)
        blocks.append(cur)
        continue
    if cur is None:
        cur = {"label": "entry", "ins": [], "succ": []}
        blocks.append(cur)
    mm = re.match(r"^\s+([a-z_0-9]+)\s*(.*)$", l)
    if not mm:
        continue
    op, rest = mm.group(1), mm.group(2)
    args = [x.strip().split(" ")[0] for x in rest.split(",")] if rest else []
    cur["ins"].append((op, args))
    if op.startswith("s_cbranch") or op == "s_branch":
        cur["succ"].append(args[0])
        nb = {"label": None, "ins": [], "succ": [], "fall": True}
        if op == "s_branch":
            cur["nofall"] = True
        blocks.append(nb)
        cur = nb
for i, bl in enumerate(blocks):
    s = [label_of[t] for t in bl["succ"] if t in label_of]
    if not bl.get("nofall") and i + 1 < len(blocks):
        s.append(i + 1)
    bl["s"] = s
    use, dfn = set(), set()
    for op, args in bl["ins"]:
        if op == "s_endpgm":
            continue
        nodef = op.startswith(NODEF)
        d = [] if nodef or not args else regs(args[0])
        srcs = args if nodef else args[1:]
        if op.startswith(("v_fmac", "v_mac", "v_writelane", "v_mfma", "v_accvgpr_write")) and op != "v_accvgpr_write_b32":
            srcs = args
        if op.startswith("v_mfma"):
            srcs = args[1:]
        if "dpp" in op or op.startswith("v_cndmask") or op.startswith("v_writelane"):
            pass
        for t in srcs:
            for r in regs(t):
                if r not in dfn:
                    use.add(r)
        # partial-exec writes keep the old value alive in other lanes: treat defs under exec masks as killing anyway (approximation)
        for r in d:
            dfn.add(r)
    bl["use"], bl["def"] = use, dfn
live_in = [set() for _ in blocks]
changed = True
while changed:
    changed = False
    for i in range(len(blocks) - 1, -1, -1):
        out = set()
        for s in blocks[i]["s"]:
            out |= live_in[s]
        li = blocks[i]["use"] | (out - blocks[i]["def"])
        if li != live_in[i]:
            live_in[i] = li
            changed = True
# report: labels that are back-edge targets
pos = 0
starts = []
for bl in blocks:
    starts.append(pos)
    pos += len(bl["ins"])
for i, bl in enumerate(blocks):
    for t in bl["succ"]:
        if t in label_of and label_of[t] <= i and starts[i] - starts[label_of[t]] > 300:
            j = label_of[t]
            v = sum(1 for r in live_in[j] if r[0] == "v")
            ag = sum(1 for r in live_in[j] if r[0] == "a")
            print(f"loop {t}: {starts[i] - starts[j]} instrs, live-in at header: {v} VGPR + {ag} AGPR")
mx = max(range(len(blocks)), key=lambda i: len(live_in[i]))
print("max live-in over blocks:", len(live_in[mx]), "at instr", starts[mx])
