#!/usr/bin/env python3
"""Instruction mix of one kernel from hipcc -S output, per basic block: how many wave-instructions of each issue class
(fp64 VALU, 32-bit VALU, transcendental, 32-bit integer multiply, LDS, scalar, vector memory) a block holds, and which blocks are
loop bodies (targets of a backward branch).  Used with the measured issue costs to price a kernel's VALU
time.   tools/isa_mix.py capi.s _ZN3ccv12k_rollout_r3ILi0ELi0E"""
import re, sys, collections

def classify(m):
    if m.startswith("v_"):
        if m.startswith(("v_mad_u64_u32", "v_mad_i64_i32")): return "mad64"
        if re.search(r"_(f64|i64|u64)\b|_f64_|f64$", m) and not m.startswith("v_cvt_f32_f64") or m in ("v_ldexp_f64", "v_fract_f64", "v_rndne_f64", "v_trig_preop_f64"):
            if re.match(r"v_(rcp|rsq|sqrt)_f64", m): return "trans64"
            if m.startswith("v_mad_u64_u32") or m.startswith("v_mad_i64_i32"): return "mad64"
            if m.startswith("v_lshl") or m.startswith("v_lshr") or m.startswith("v_ashr"): return "v32"
            return "f64"
        if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f(32|16)", m): return "trans32"
        if re.match(r"v_mul_(hi|lo)_(u|i)32", m): return "mul32"
        if m.startswith("v_mfma"): return "mfma"
        return "v32"
    if m.startswith("ds_"): return "lds"
    if m.startswith("s_waitcnt"): return "wait"
    if m.startswith("s_"): return "salu"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"

def main():
    path, prefix = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l and not l.startswith("\t"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur, order = collections.OrderedDict(), "entry", {}
    blocks[cur] = []
    for l in lines[start + 1:end + 1]:
        t = l.strip()
        m = re.match(r"^(\.LBB[0-9_]+):", t)
        if m:
            cur = m.group(1); blocks[cur] = []; continue
        if not t or t.startswith((";", ".", "//")): continue
        blocks[cur].append(t.split(";")[0].strip())
    names = list(blocks)
    pos = {n: i for i, n in enumerate(names)}
    loops = []           # (head, tail) by backward branch
    for n in names:
        for ins in blocks[n]:
            m = re.match(r"s_c?branch\S*\s+(\.LBB[0-9_]+)", ins)
            if m and pos.get(m.group(1), 1 << 30) <= pos[n]:
                loops.append((m.group(1), n))
    classes = ["f64", "v32", "mul32", "trans32", "trans64", "mad64", "lds", "vmem", "salu", "wait"]
    print("%-14s %6s " % ("block", "instr") + " ".join("%7s" % c for c in classes) + "  loops(head<-tail)")
    tot = collections.Counter()
    for n in names:
        c = collections.Counter(classify(i.split()[0]) for i in blocks[n])
        tot.update(c)
        inl = [h for h, t in loops if pos[h] <= pos[n] <= pos[t]]
        if sum(c.values()) >= int(sys.argv[3]) if len(sys.argv) > 3 else 1:
            print("%-14s %6d " % (n, sum(c.values())) + " ".join("%7d" % c[k] for k in classes) + "  " + ",".join(inl))
    print("%-14s %6d " % ("total", sum(tot.values())) + " ".join("%7d" % tot[k] for k in classes))
    print("loops:", loops)

main()
