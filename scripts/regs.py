#!/usr/bin/env python3
"""register / scratch usage per kernel instantiation from a hipcc -save-temps gfx950 .s file (developer tool)"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for m in re.finditer(r"\.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    ag, name, priv, sg, vg, sp = m.groups()
    rows.append((name, vg, ag, sp, priv))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (name, vg, ag, sp, priv), dem in zip(rows, names):
    if pat in dem:
        dem = dem.replace("void (anonymous namespace)::", "").replace("(sb_lstm_bwd_args)", "").replace("(sb_lstm_fwd_args)", "")
        print(f"vgpr {vg:>4} agpr {ag:>4} spill {sp:>4} scratch {priv:>5}  {dem[:110]}")
