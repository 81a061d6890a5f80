#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of scripts/gpu_pmc.sh into profiles/r01_final_pmc_traffic_<wl>.json:
HBM bytes per launch of the recurrent kernels = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE,
both counters in KB -> bytes x 1024.  usage: pmc_summary.py <workload> [gpurun_out dir] [out json]"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1] if len(sys.argv) > 1 else "small"
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", f"r01_final_pmc_traffic_{wl}.json")


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "lstm_" not in r["Kernel_Name"]:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = name.split("(")[0] + " grid=" + r["Grid_Size"]
        acc[name][0] += float(r["Counter_Value"])
        acc[name][1] += 1
    return acc


f = per_kernel(os.path.join(src, f"pmc_fetch_{wl}", "f_counter_collection.csv"), "FETCH_SIZE")
w = per_kernel(os.path.join(src, f"pmc_write_{wl}", "w_counter_collection.csv"), "WRITE_SIZE")
out = {"note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1 "
               f"--workload {wl}`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced "
               f"reads); KB -> bytes x1024; per-launch averages (scripts/pmc_summary.py)",
       "kernels": {}}
for k in sorted(f, key=lambda k: -f[k][0]):
    if k not in w:
        continue
    fb = 2.0 * 1024.0 * f[k][0] / f[k][1]
    wb = 1024.0 * w[k][0] / w[k][1]
    out["kernels"][k] = {"launches": f[k][1], "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
json.dump(out, open(dst, "w"), indent=1)
print(dst, len(out["kernels"]), "kernels")
