#!/usr/bin/env python3
"""instruction mix of the hottest loop (largest backward-branch span) of one kernel in a -save-temps gfx950 .s file
usage: loopstats.py file.s '<demangled-substring>' (developer tool)"""
import re, subprocess, sys
from collections import Counter
txt = open(sys.argv[1]).read()
names = re.findall(r"^(_Z\S+):", txt, re.M)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for n, d in zip(names, dem):
    if sys.argv[2] not in d:
        continue
    i = txt.index("\n" + n + ":"); j = txt.index(".Lfunc_end", i)
    lines = txt[i:j].split("\n")
    lab = {}
    for k, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: lab[m.group(1)] = k
    best = (0, 0, 0)
    for k, l in enumerate(lines):
        m = re.search(r"s_cbranch\S*\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in lab and lab[m.group(1)] < k and k - lab[m.group(1)] > best[0]:
            best = (k - lab[m.group(1)], lab[m.group(1)], k)
    _, a, b = best
    c = Counter()
    for l in lines[a:b]:
        if not l.startswith("\t") or l.strip().startswith((".", ";")): continue
        ins = l.split()[0]
        key = ("mfma" if ins.startswith("v_mfma") else "accvgpr" if ins.startswith("v_accvgpr") else
               "trans" if re.match(r"v_(exp|rcp|rsq|log|sqrt)", ins) else "cvt" if ins.startswith("v_cvt") else
               "valu" if ins.startswith("v_") else "lds" if ins.startswith("ds_") else
               "vmem" if ins.startswith(("global_", "buffer_", "flat_")) else "scratch" if ins.startswith("scratch_") else
               "waitcnt" if ins.startswith("s_waitcnt") else "nop" if ins.startswith("s_nop") else
               "barrier" if ins.startswith("s_barrier") else "salu" if ins.startswith("s_") else "other")
        c[key] += 1
    print(d[:120]); print("  loop lines", b - a, dict(c))
