#!/usr/bin/env python
"""Per-loop instruction mix of one kernel in a hipcc -S listing: `isa_loops.py file.s <mangled-substring>`.

Finds the kernel, every backward branch (a loop) and prints, for each loop body, how many instructions of each class it
holds (VALU full-rate / transcendental / DPP / v_mov / accvgpr moves, LDS, global memory, waitcnt, barriers, scalar).
Used to see where a wave's issue slots go (the solver kernels are bound by per-wave instruction count).
"""
import collections
import re
import sys

TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")


def classify(op, line):
    if op.startswith("v_accvgpr") or (op.startswith("v_mov") and "a[" in line):
        return "accvgpr"
    if "dpp" in line or "row_" in line or "quad_perm" in line or op.startswith("v_permlane") or op.startswith("v_readlane") \
            or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "xlane"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_mov"):
        return "v_mov"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"):
        return "v_cmp/sel"
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "scratch" if op.startswith("scratch_") else "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end + 1]
    labels, instrs = {}, []
    for l in body:
        s = l.strip()
        m = re.match(r"^(\.LBB[0-9_]+):", s)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        if not s or s.startswith((";", ".", "_Z")):
            continue
        op = s.split()[0]
        instrs.append((op, s))
    print("kernel %s: %d instructions" % (lines[start].split(":")[0], len(instrs)))
    loops = []
    for i, (op, s) in enumerate(instrs):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    for a, b, tgt in sorted(loops):
        if b - a < 40:
            continue
        c = collections.Counter(classify(op, s) for op, s in instrs[a:b + 1])
        print("loop %-12s %5d instrs: " % (tgt, b - a + 1) + "  ".join("%s=%d" % kv for kv in c.most_common()))


if __name__ == "__main__":
    main()
