#!/usr/bin/env python
"""Instruction histogram of one kernel in a hipcc -S listing (development aid).

usage: python tools/asm_hist.py <listing.s> <substring of the mangled kernel name> [--loops]

Prints, for the whole kernel and (with --loops) for every backward-branch loop body, the number of instructions per
class: f64 VALU, f32 VALU, integer/move VALU, transcendental, LDS, VMEM, SALU, waitcnt, branches.  Issue-cycle estimate:
4 cycles per wave64 VALU instruction, 8 for float64 FMA/MUL/ADD on 64 lanes at half rate where applicable (the table
below is the estimate this tool uses, not a vendor figure)."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_"):
        if re.search(r"_f64|f64_", op):
            if re.search(r"rcp|rsq|sqrt|div_|trig|frexp|ldexp", op):
                return "valu_f64_special"
            return "valu_f64"
        if re.search(r"(sin|cos|exp|log|rcp|rsq|sqrt)_f32", op):
            return "valu_trans_f32"
        if re.search(r"_f32|f32_|_f16|bf16", op):
            return "valu_f32"
        if op.startswith("v_cmp") or op.startswith("v_cndmask"):
            return "valu_cmp_sel"
        if re.search(r"readlane|readfirstlane|writelane|permlane|mov_b32_dpp|_dpp", op):
            return "valu_xlane"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            return "valu_mov"
        return "valu_int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_sleep"):
        return "nop_barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    loops = "--loops" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l) and key in l:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    end = start
    while not lines[end].strip().startswith("s_endpgm"):
        end += 1
    insts, labels = [], {}
    for l in lines[start + 1:end + 1]:
        s = l.split(";")[0].strip()
        if not s or s.startswith("."):
            if re.match(r"^\.LBB\w+:", s):
                labels[s[:-1]] = len(insts)
            continue
        if re.match(r"^\.?\w+:$", s):
            labels[s[:-1]] = len(insts)
            continue
        insts.append(s)

    def hist(lo, hi, title):
        c = Counter(classify(x.split()[0]) for x in insts[lo:hi])
        tot = sum(c.values())
        print("%s: %d instructions" % (title, tot))
        for k, v in c.most_common():
            print("   %-18s %6d" % (k, v))
        ops = Counter(x.split()[0] for x in insts[lo:hi])
        print("   top ops: " + ", ".join("%s %d" % kv for kv in ops.most_common(24)))

    print(lines[start])
    hist(0, len(insts), "kernel")
    if loops:
        seen = []
        for i, x in enumerate(insts):
            op = x.split()[0]
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = x.split()[-1]
                if tgt in labels and labels[tgt] <= i:
                    seen.append((labels[tgt], i + 1, tgt))
        for lo, hi, tgt in sorted(seen, key=lambda t: t[0] - t[1]):
            hist(lo, hi, "loop %s [%d,%d)" % (tgt, lo, hi))


if __name__ == "__main__":
    main()
