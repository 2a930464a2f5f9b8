"""Static instruction mix per kernel of an AMDGPU assembly file (hipcc -S --cuda-device-only).
usage: python tools/isa_mix.py file.s [name-substring]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\n(_Z\S+):[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter()
    for x in ins:
        if x.startswith("global_load") or x.startswith("flat_load"):
            c["gload"] += 1
        elif x.startswith("global_store") or x.startswith("flat_store"):
            c["gstore"] += 1
        elif x.startswith("scratch_") or x.startswith("buffer_"):
            c["scratch_" + ("ld" if "load" in x else "st")] += 1
        elif x.startswith("ds_"):
            c["lds"] += 1
        elif "dpp" in x:
            c["dpp"] += 1
        elif x.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif x.startswith("s_nop"):
            c["nop"] += 1
        elif x.startswith("s_"):
            c["salu"] += 1
        elif re.match(r"v_(fma|mul|add|fmac)_f64", x):
            c["f64"] += 1
        elif x.startswith("v_cndmask"):
            c["cndmask"] += 1
        elif x.startswith("v_mov") or x.startswith("v_accvgpr"):
            c["mov"] += 1
        elif x.startswith("v_cmp"):
            c["cmp"] += 1
        elif x.startswith("v_"):
            c["valu_other"] += 1
        else:
            c["other"] += 1
    t = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)E", name)
    print((t.groups() if t else name[:60]), len(ins), dict(c))
