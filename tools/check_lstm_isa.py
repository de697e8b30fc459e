#!/usr/bin/env python
"""Build-time guard for the issue order of the BPTT kernel's tile loop (csrc/hode_lstm_kernels.hpp::lstm_bwd_kernel).

The kernel's time hangs on a schedule the SOURCE pins with sched_barrier / sched_group_barrier and a basic-block boundary
per unit tile (without the boundary the ten software-pipelined bodies of a step merge into one
3 000-instruction block and the kernel is 0.47 ms = 15 % slower at the bench shape).  A compiler update can undo that
without failing any numerics test, so this script reads the ISA of the built object and fails the build when the shape is
lost.  For lstm_bwd_kernel<NT, TPW, FLAT> it requires:

  * exactly TPW tile segments (delimited by branch instructions) that hold MFMAs, 4 * TPW * NT of them each;
  * inside a segment, between its first and its last MFMA: no VALU instruction at all (the element-wise burst of the next
    tile sits AHEAD of the MFMAs: fp32 MFMAs and VALU work share the issue port, tools/micro/mfma_valu_overlap.hip), at most
    MAX_GAP other instructions between two MFMAs (loads and their counted waits), no full-drain `s_waitcnt vmcnt(0)`;
  * no scratch (private segment 0).

    python tools/check_lstm_isa.py [--tpw 10] [--nt 3] [--verbose]      # exit code 0 = shape intact
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"
MAX_GAP = 3


def code_object(obj, out):
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=%s.fb" % out, obj, out + ".dummy"], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=%s.fb" % out,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], check=True)
    return out


def kernel_ops(co, symbol):
    txt = subprocess.run([LLVM + "llvm-objdump", "-d", "--disassemble-symbols=" + symbol, co], capture_output=True, text=True, check=True).stdout
    ops = []
    for line in txt.splitlines():
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*?)\s*//", line)
        if m:
            ops.append((m.group(1), m.group(2)))
    return ops


def check(tpw, nt, flat=True, verbose=False, tmp="/tmp/hode_isa", obj=None):
    obj = obj or os.path.join(ROOT, "hybrid-ode-neurips-2021_amd", "csrc", "build", "hode_lstm_tpw%d.o" % tpw)
    if not os.path.exists(obj):
        raise SystemExit("check_lstm_isa: %s not built (python build_hip.py)" % obj)
    os.makedirs(tmp, exist_ok=True)
    co = code_object(obj, os.path.join(tmp, "lstm_tpw%d.co" % tpw))
    sym = "_ZN4hode15lstm_bwd_kernelILi%dELi%dELb%dEEEvNS_11LstmBwdArgsE" % (nt, tpw, 1 if flat else 0)
    ops = kernel_ops(co, sym)
    if not ops:
        raise SystemExit("check_lstm_isa: kernel %s not found in %s" % (sym, obj))
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    scratch = vgpr = -1
    for blk in notes.split("- .agpr_count")[1:]:   # one metadata record per kernel
        if re.search(r"\.name:\s+%s\s" % re.escape(sym), blk):
            scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
            vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
    problems = []
    if scratch:
        problems.append("kernel uses %d B of scratch (register spills)" % scratch)
    is_branch = lambda o: o.startswith(("s_cbranch", "s_branch"))
    bounds = [-1] + [i for i, (o, _) in enumerate(ops) if is_branch(o)] + [len(ops)]
    segs = []
    for s, e in zip(bounds[:-1], bounds[1:]):
        m = [i for i in range(s + 1, e) if "mfma" in ops[i][0]]
        if m:
            segs.append((s + 1, e, m))
    want = 4 * tpw * nt
    if len(segs) != tpw:
        problems.append("%d basic blocks hold MFMAs, expected %d (one per unit tile): the tile bodies were merged or split" % (len(segs), tpw))
    for k, (s, e, m) in enumerate(segs):
        if len(m) != want:
            problems.append("tile block %d holds %d MFMAs, expected %d" % (k, len(m), want))
        inner = ops[m[0]:m[-1] + 1]
        valu = [o for o, _ in inner if o.startswith("v_") and "mfma" not in o]
        gap = max([b - a - 1 for a, b in zip(m[:-1], m[1:])] or [0])
        drains = [a for o, a in inner if o == "s_waitcnt" and re.search(r"vmcnt\(0\)", a)]
        if verbose:
            pre = ops[s:m[0]]
            print("tile block %2d: %3d MFMAs, max gap %d, VALU between MFMAs %d, full drains %d | ahead of the MFMAs: %d VALU, %d loads, %d LDS"
                  % (k, len(m), gap, len(valu), len(drains), sum(1 for o, _ in pre if o.startswith("v_")),
                     sum(1 for o, _ in pre if o.startswith(("buffer_", "global_"))), sum(1 for o, _ in pre if o.startswith("ds_"))))
        if valu:
            problems.append("tile block %d: %d VALU instructions between its MFMAs (%s ...)" % (k, len(valu), valu[0]))
        if gap > MAX_GAP:
            problems.append("tile block %d: %d instructions between two MFMAs (max %d)" % (k, gap, MAX_GAP))
        if drains:
            problems.append("tile block %d: s_waitcnt vmcnt(0) between its MFMAs" % k)
    if verbose or problems:
        print("lstm_bwd_kernel<%d, %d, %s>: %d instructions, %d VGPRs, scratch %d B, %d tile blocks" % (nt, tpw, str(flat).lower(), len(ops), vgpr, scratch, len(segs)))
    return problems


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--tpw", type=int, default=10)
    ap.add_argument("--nt", type=int, default=3)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--obj", help="object file to read instead of the product build's")
    a = ap.parse_args()
    bad = check(a.tpw, a.nt, verbose=a.verbose, obj=a.obj)
    for p in bad:
        print("ISA CHECK FAILED: " + p)
    sys.exit(1 if bad else 0)
