#!/usr/bin/env python3
"""instruction mix of EVERY loop (backward branch) of one kernel in a -save-temps gfx950 .s file, innermost first
usage: loops_all.py file.s '<demangled-substring>' [dump-loop-index]   (developer tool)"""
import re, subprocess, sys
from collections import Counter
txt = open(sys.argv[1]).read()
names = re.findall(r"^(_Z\S+):", txt, re.M)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
def key(ins):
    return ("mfma" if ins.startswith("v_mfma") else "accvgpr" if ins.startswith("v_accvgpr") else
            "trans" if re.match(r"v_(exp|rcp|rsq|log|sqrt)", ins) else "cvt" if ins.startswith("v_cvt") else
            "valu" if ins.startswith("v_") else "lds" if ins.startswith("ds_") else
            "vmem" if ins.startswith(("global_", "buffer_", "flat_")) else "scratch" if ins.startswith("scratch_") else
            "waitcnt" if ins.startswith("s_waitcnt") else "nop" if ins.startswith("s_nop") else
            "barrier" if ins.startswith("s_barrier") else "salu" if ins.startswith("s_") else "other")
for n, d in zip(names, dem):
    if sys.argv[2] not in d:
        continue
    i = txt.index("\n" + n + ":"); j = txt.index(".Lfunc_end", i)
    lines = txt[i:j].split("\n")
    lab = {}
    for k, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: lab[m.group(1)] = k
    loops = []
    for k, l in enumerate(lines):
        m = re.search(r"s_cbranch\S*\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < k:
            loops.append((k - lab[m.group(1)], lab[m.group(1)], k))
    loops.sort()
    print(d[:140])
    for idx, (_, a, b) in enumerate(loops):
        c = Counter()
        for l in lines[a:b]:
            if not l.startswith("\t") or l.strip().startswith((".", ";")): continue
            c[key(l.split()[0])] += 1
        print("  loop", idx, "lines", b - a, dict(sorted(c.items())))
        if len(sys.argv) > 3 and int(sys.argv[3]) == idx:
            print("\n".join(lines[a:b + 1]))
    break
