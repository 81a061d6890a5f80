#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of scripts/gpu_pmc.sh:
  * pmc_traffic_<wl>.json -- HBM bytes per launch of every kernel that takes >= 0.5 % of the traced time:
    2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md: wide coalesced reads are tallied at half) + WRITE_SIZE,
    both counters in KB -> bytes x 1024, with the kernel's average duration from the same pass;
  * pmc_sq_<wl>.json -- SQ activity per kernel: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the chip's
    1024 SIMDs) / (launch duration in the same pass x shader clock x 1024) -- the clock from GRBM_GUI_ACTIVE when the pass
    collected it and it is plausible, else 2.0 GHz (MI355X_MICROARCH.md: profiled passes run 1.89-1.95 GHz); round 4 divided by
    SQ_BUSY_CYCLES, a per-SE count, and got "fractions" of 3.5-10 -- and the wave-cycle split SQ_WAIT_ANY (parked on s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stalls),
    SQ_ACTIVE_INST_ANY (issuing), SQ_ACTIVE_INST_VALU over SQ_WAVE_CYCLES (quad-cycle units).
usage: pmc_summary.py <workload> [gpurun_out dir] [traffic json] [sq json]"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sound_bubble_amd.build import csrc_digest            # noqa: E402
PROVENANCE = {"csrc_sha16": csrc_digest(), "what": "sha256[:16] over sound_bubble_amd/csrc/*.{hip,h} + include/sound_bubble_hip.h of the "
              "tree the counters were taken on (sound_bubble_amd.build.csrc_digest); bench.py flags a summary whose digest differs "
              "from the running tree's as stale_profile", "tree": os.environ.get("SB_TREE_ID")}
wl = sys.argv[1] if len(sys.argv) > 1 else "small"
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", f"r02_pmc_traffic_{wl}.json")
dst_sq = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", f"r02_pmc_sq_{wl}.json")
csv.field_size_limit(1 << 30)


def short(name, grid):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name.split("(")[0][:160] + " grid=" + grid


def per_kernel(path):
    """-> {kernel: {counter: [sum, n]}}, {kernel: [sum_ns, n]}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    dur = collections.defaultdict(lambda: [0.0, 0])
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"], r["Grid_Size"])
        c = acc[k][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"])
        c[1] += 1
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            dur[k][1] += 1
    return acc, dur


def main():
    fpath = os.path.join(src, f"pmc_fetch_{wl}", "f_counter_collection.csv")
    wpath = os.path.join(src, f"pmc_write_{wl}", "w_counter_collection.csv")
    if os.path.exists(fpath) and os.path.exists(wpath):
        f, fd = per_kernel(fpath)
        w, _ = per_kernel(wpath)
        total = sum(v[0] for v in fd.values())
        out = {"note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `bench.py --steps 2 --warmup 1 "
                       f"--workload {wl} --no-exact`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies wide "
                       f"coalesced reads at half); KB -> bytes x1024; per-launch averages; kernels >= 0.5 % of the traced "
                       f"time (scripts/pmc_summary.py)", "kernels": {}}
        for k in sorted(f, key=lambda k: -fd[k][0]):
            if k not in w or fd[k][0] < 0.005 * total:
                continue
            fe, wr = f[k]["FETCH_SIZE"], w[k]["WRITE_SIZE"]
            fb, wb = 2.0 * 1024.0 * fe[0] / fe[1], 1024.0 * wr[0] / wr[1]
            avg_us = fd[k][0] / fd[k][1] / 1e3
            out["kernels"][k] = {"launches": fe[1], "avg_us_in_pmc_pass": avg_us, "fetch_bytes_corrected": fb,
                                 "write_bytes": wb, "hbm_bytes": fb + wb, "hbm_tbs": (fb + wb) / (avg_us * 1e-6) / 1e12,
                                 "share_of_traced_time": fd[k][0] / total}
        out["provenance"] = PROVENANCE
        json.dump(out, open(dst, "w"), indent=1)
        print(dst, len(out["kernels"]), "kernels")
    spath = os.path.join(src, f"pmc_sq_{wl}", "s_counter_collection.csv")
    if os.path.exists(spath):
        s, sd = per_kernel(spath)
        total = sum(v[0] for v in sd.values())
        out = {"note": f"rocprofv3 --pmc SQ_* pass of `bench.py --steps 2 --warmup 1 --workload {wl} --no-exact`; sums over "
                       f"all SEs per launch, averaged over launches.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                       f"(duration x shader clock x 1024 SIMDs); the wave-cycle split is in quad-cycles (MI355X_MICROARCH.md): WAIT_ANY = parked on "
                       f"s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing", "kernels": {}}
        for k in sorted(s, key=lambda k: -sd[k][0]):
            if sd[k][0] < 0.005 * total:
                continue
            g = lambda c: (s[k][c][0] / s[k][c][1]) if c in s[k] and s[k][c][1] else None
            wc, busy, mf = g("SQ_WAVE_CYCLES"), g("SQ_BUSY_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES")
            ent = {"launches": sd[k][1], "avg_us_in_pmc_pass": sd[k][0] / sd[k][1] / 1e3,
                   "share_of_traced_time": sd[k][0] / total}
            for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                      "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"):
                ent[c] = g(c)
            gui = g("GRBM_GUI_ACTIVE")
            clk = None
            if gui:                                     # summed over XCDs or not, depending on the profiler build: take what is sane
                for div in (1.0, 8.0):
                    c = gui / div / (ent["avg_us_in_pmc_pass"] * 1e-6)
                    if 1.2e9 <= c <= 2.6e9:
                        clk = c
                        break
            ent["GRBM_GUI_ACTIVE"] = gui
            ent["shader_clock_hz"] = clk or 2.0e9
            ent["shader_clock_source"] = "GRBM_GUI_ACTIVE / duration" if clk else "assumed 2.0 GHz"
            if mf is not None:
                ent["mfma_busy"] = mf / (ent["avg_us_in_pmc_pass"] * 1e-6 * ent["shader_clock_hz"] * 1024.0)
            if wc:
                for c, n in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"),
                             ("SQ_ACTIVE_INST_ANY", "active_inst_frac"), ("SQ_ACTIVE_INST_VALU", "active_valu_frac")):
                    if g(c) is not None:
                        ent[n] = g(c) / wc
            out["kernels"][k] = ent
        out["provenance"] = PROVENANCE
        json.dump(out, open(dst_sq, "w"), indent=1)
        print(dst_sq, len(out["kernels"]), "kernels")


main()
