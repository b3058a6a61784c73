#!/usr/bin/env python3
"""Build-time guard: vector instructions that sit in a JOIN block in front of its EXEC restore (gfx950, hipcc -S output).

The pattern this finds is the cause of round 2's "multipliers attributed to the wrong rows" build of the T = 30 kernel
(compiler fact 6):

        s_and_saveexec_b64 s[4:5], s[6:7]      ; if (ok01) ...            divergent `if`
    ; %bb.705:                                 ;   then-block
        ...
    ; %bb.706:                                 ;   JOIN block
        v_accvgpr_write_b32 a46, v33           ; <- a live-range-split copy of a value that is live in ALL lanes ...
        s_or_b64 exec, exec, s[4:5]            ; <- ... placed BEFORE the exec restore: only the lanes of the `if` get it

LLVM's split / spill insertion skips the "block prologue" (exec restores, SGPR spills) when it places a VECTOR copy at the top
of a block, but a SCALAR copy that was placed there earlier ends the prologue scan (SIInstrInfo::isBasicBlockPrologue returns
false for COPY), so the vector copy lands in front of the restore and executes under the narrowed mask.  Lanes outside the
mask keep whatever the destination register held; the value is read later with all lanes active.  Whether a build has the
pattern depends on the allocator's split decisions, i.e. on everything -- which is why it came and went with unrelated pins.

usage: isa_exec_check.py file.s [kernel-name-substring]      exit status 1 if any finding
"""
import re
import sys

VEC = re.compile(r"^(v_|ds_|global_|buffer_|scratch_|flat_)")
HARMLESS = re.compile(r"^(v_readlane_b32|v_readfirstlane_b32|v_writelane_b32)")   # lane-indexed SGPR spill traffic: exec-independent
NARROW = re.compile(r"^(s_and_saveexec_b64|s_or_saveexec_b64|s_andn2_saveexec_b64)\b|^s_(xor|andn2|and|mov)_b64\s+exec,")
RESTORE = re.compile(r"^s_or_b64\s+exec,\s*exec,")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FALL = re.compile(r"^; %bb\.\d+:")


def kernels(path):
    """{kernel symbol: [(line number, code or block marker)]} of hipcc's -S output."""
    out, cur = {}, None
    for ln, raw in enumerate(open(path), 1):
        line = raw.rstrip("\n")
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        if LABEL.match(line) or FALL.match(line):
            cur.append((ln, "#" + line.split(":")[0].lstrip("; ")))
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        cur.append((ln, code))
        if code.split()[0] == "s_endpgm":
            cur = None
    return out


COPY = re.compile(r"^(v_mov_b32|v_mov_b64|v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32)")


def check_kernel(ins):
    """Which blocks hold vector instructions in front of their first exec restore, and which of those are the bug.

    Legitimate: the restore merged into the TAIL of a region body -- a block that directly follows the exec-narrowing instruction
    (+ its skip branch), the target of the `s_cbranch_execnz` that follows one, or a later block of a region that has uniform
    control flow inside (labelled, reached by scalar branches) and ends with real work.
    The bug (a JOIN block with a live-range-split / spill copy on the wrong side of its restore) shows as one of
      (a) an unlabelled block entered by pure fall-through (no branch in front of its `; %bb.N:` marker): it is a separate block only
          because it has a second predecessor whose branch was elided -- the `s_cbranch_execz` of a short then-block;
      (b) a labelled block that is the target of an `s_cbranch_execz`: the skip target IS the join block;
      (c) any other non-body block in which everything in front of the restore is a pure register copy."""
    body_labels, skip_labels = set(), set()
    for i, (ln, c) in enumerate(ins):
        if c.startswith("s_cbranch_execnz") and i > 0 and NARROW.match(ins[i - 1][1]):
            body_labels.add(c.split()[1])
        if c.startswith("s_cbranch_execz"):
            skip_labels.add(c.split()[1])
    findings = []
    n = len(ins)
    for i, (ln, c) in enumerate(ins):
        if not c.startswith("#"):
            continue
        name = c[1:]
        j = i - 1
        prev = ins[j][1] if j >= 0 else ""
        after_branch = prev.startswith("s_cbranch") or prev.startswith("s_branch")
        if j >= 0 and prev.startswith("s_cbranch_exec"):
            j -= 1
        is_body = (j >= 0 and bool(NARROW.match(ins[j][1]))) or name in body_labels
        pending, k = [], i + 1
        while k < n and not ins[k][1].startswith("#"):
            code = ins[k][1]
            op = code.split()[0]
            if RESTORE.match(code):
                if pending and not is_body:
                    labelled = name.startswith(".LBB")
                    kind = None
                    if not labelled and not after_branch and not (i > 0 and NARROW.match(prev)):
                        kind = "a"
                    elif labelled and name in skip_labels:
                        kind = "b"
                    elif all(COPY.match(cc.split()[0]) for _, cc in pending):
                        kind = "c"
                    if kind:
                        findings.append((name, ins[k][0], pending))
                break                              # only what precedes the FIRST restore of the block is in question
            if NARROW.match(code) or op.startswith("s_cbranch") or op == "s_branch":
                break
            if VEC.match(op) and not HARMLESS.match(op):
                pending.append((ins[k][0], code))
            k += 1
    return findings


def check(path, only=None):
    res = []
    for kname, ins in kernels(path).items():
        if only and only not in kname:
            continue
        for name, ln, pend in check_kernel(ins):
            res.append((kname, name, ln, pend))
    return res


def main():
    path = sys.argv[1]
    only = sys.argv[2] if len(sys.argv) > 2 else None
    f = check(path, only)
    for kernel, block, ln, ins in f:
        print(f"{kernel[:60]}  block {block}  exec restore at line {ln}: {len(ins)} vector instruction(s) in front of it in the same block")
        for l, c in ins[:6]:
            print(f"      {l}: {c}")
    print(f"{path}: {len(f)} finding(s)")
    return 1 if f else 0


if __name__ == "__main__":
    sys.exit(main())
