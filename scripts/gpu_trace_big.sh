#!/bin/bash
# kernel trace (timeline) of the big train step in the current default mode -> per-kernel stats + the raw trace of one step
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or overlapped" 2>&1 | tail -6) > gpurun_out/t_wide.log 2>&1
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_big_wide" -o big -- python "$R/bench.py" --steps 3 --warmup 1 --workload big --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_big_wide.log" 2>&1
cd "$R"; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_big_wide/**/big_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last ~400 kernel records (the last timed step) in a compact form
t0 = int(rows[-400]["Start_Timestamp"])
with open("gpurun_out/big_wide_timeline_tail.txt", "w") as o:
    for r in rows[-400:]:
        o.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:9.1f} q{r.get('Queue_Id','?')} {r['Kernel_Name'][:110]}\n")
PY
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*.db" -delete
tail -4 gpurun_out/t_wide.log
