#!/usr/bin/env python3
"""Instruction mix of every loop of one kernel in hipcc's gfx950 assembly (hipcc -S --cuda-device-only).

usage: isa_loops.py file.s <substring of the mangled kernel name> [--all]
A loop = a label that a later branch jumps back to; nested loops are reported separately (inner bodies are part of the
outer count).  Columns: instructions of each class between the label and the backward branch.
"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().endswith(("E", "E:")) or (l.startswith("_Z") and key in l and ":" in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels, insts = {}, []
    for l in body:
        t = l.strip()
        m = re.match(r"^(\.LBB[0-9_]+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        insts.append((op, t))
    total = {}
    for op, _ in insts:
        total[classify(op)] = total.get(classify(op), 0) + 1
    print("kernel: %d instructions %s" % (len(insts), total))
    for i, (op, t) in enumerate(insts):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                lo = labels[tgt]
                mix = {}
                for o, _ in insts[lo:i + 1]:
                    mix[classify(o)] = mix.get(classify(o), 0) + 1
                if mix.get("mfma", 0) or "--all" in sys.argv:
                    print("loop %s [%d..%d] %d insts: %s" % (tgt, lo, i, i + 1 - lo, dict(sorted(mix.items()))))


main()
