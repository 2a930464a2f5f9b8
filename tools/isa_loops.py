"""Instruction mix of one kernel of an AMDGPU assembly file (hipcc -S --cuda-device-only), whole and per loop
(backward-branch ranges at least `min` instructions long).  usage: python tools/isa_loops.py file.s <kernel-name-substring> [min]"""
import collections
import re
import sys


def cat(x):
    if re.match(r"v_(fma|mul|add|fmac|max|min)_f64|v_(rcp|rsq|sqrt|div_scale|div_fmas|div_fixup|trig|fract|floor|ldexp|frexp)\w*f64", x):
        return "fp64 arith"
    if x.startswith("v_cmp") and "f64" in x:
        return "fp64 compare"
    if x.startswith("v_cndmask"):
        return "select (cndmask)"
    if "dpp" in x:
        return "DPP move"
    if x.startswith("v_accvgpr"):
        return "AGPR <-> VGPR move"
    if x.startswith("v_mov") or x.startswith("v_pk_mov"):
        return "VGPR move"
    if x.startswith("ds_"):
        return "LDS"
    if x.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "global memory"
    if x.startswith("s_waitcnt") or x.startswith("s_nop"):
        return "waitcnt / nop"
    if x.startswith("s_"):
        return "scalar"
    if x.startswith("v_cmp"):
        return "integer compare"
    if x.startswith("v_"):
        return "integer / bit VALU"
    return "other"


txt = open(sys.argv[1]).read()
want = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for m in re.finditer(r"\n(_Z\S+):[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    seq, labels = [], {}
    for l in body.split("\n"):
        if re.match(r"\.LBB\d+_\d+:", l):
            labels[l.split(":")[0]] = len(seq)
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            seq.append(l.strip())
    loops = set()
    for i, l in enumerate(seq):
        mm = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] <= i and i - labels[mm.group(1)] >= minlen:
            loops.add((labels[mm.group(1)], i))
    print(name[:80])
    for a, b in [(0, len(seq))] + sorted(loops):
        c = collections.Counter(cat(x.split()[0]) for x in seq[a:b + 1])
        tot = sum(c.values())
        print(f"[{a}, {b}] {tot} instructions: " + ", ".join(f"{k} {v} ({100 * v / tot:.0f} %)" for k, v in c.most_common()))
