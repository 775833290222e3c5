#!/usr/bin/env python3
"""Instruction census of one kernel from hipcc -S output: per basic block, VALU instructions by issue class.
   python tools/isa_census.py file.s [min_instr]"""
import collections
import re
import sys

CLASS = [
    ("f64", re.compile(r"^v_(fma|mul|add|min|max|cmp\w*|cvt_f64|cvt_\w+_f64|ldexp|frexp\w*|rndne|trunc|floor|fract|rcp|rsq|sqrt|div\w*|trig_preop)_f64|^v_cvt_f64|^v_cvt_(i32|u32|f32)_f64|^v_cmp\w*_f64")),
    ("i64mul", re.compile(r"^v_mad_[ui]64_[ui]32|^v_mul_(hi|lo)_[ui]32")),
    ("trans32", re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32")),
    ("pk", re.compile(r"^v_pk_")),
    ("dpp/perm", re.compile(r"dpp|^v_permlane|^v_readlane|^v_readfirstlane|^v_writelane")),
    ("valu", re.compile(r"^v_")),
    ("lds", re.compile(r"^ds_")),
    ("vmem", re.compile(r"^(global|buffer|flat|scratch)_")),
    ("smem", re.compile(r"^s_(load|buffer_load)")),
    ("wait", re.compile(r"^s_waitcnt|^s_barrier|^s_nop|^s_sleep")),
    ("salu", re.compile(r"^s_")),
]


def classify(op):
    for name, rx in CLASS:
        if rx.search(op):
            return name
    return "other"


def main():
    path = sys.argv[1]
    thresh = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    blocks, cur, name = [], collections.Counter(), "entry"
    ops = collections.Counter()
    for line in open(path):
        line = line.split(";")[0].strip()
        if not line or line.startswith("."):
            if line.startswith(".LBB") and line.endswith(":"):
                blocks.append((name, cur)); cur, name = collections.Counter(), line[:-1]
            continue
        if line.endswith(":"):
            blocks.append((name, cur)); cur, name = collections.Counter(), line[:-1]
            continue
        op = line.split()[0]
        c = classify(op)
        cur[c] += 1
        if c in ("f64", "valu", "i64mul", "trans32"):
            ops[(name, op)] += 1
    blocks.append((name, cur))
    cols = [c for c, _ in CLASS]
    print("%-14s" % "block" + "".join("%9s" % c for c in cols) + "   VALU-total")
    for n, c in blocks:
        tot = sum(c.values())
        if tot < thresh:
            continue
        v = sum(c[k] for k in ("f64", "i64mul", "trans32", "pk", "dpp/perm", "valu"))
        print("%-14s" % n[-14:] + "".join("%9d" % c[k] for k in cols) + "   %d" % v)
    if len(sys.argv) > 3:
        want = sys.argv[3]
        print("\nops in", want)
        for (n, op), k in sorted(ops.items(), key=lambda kv: -kv[1]):
            if n == want:
                print("  %-24s %d" % (op, k))


main()
