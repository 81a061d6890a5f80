"""parse a rocprofv3 kernel trace of exp_bwd_overlap.py: start / end of the recurrence and stream kernels, relative"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "lstm_bwd" in r["Kernel_Name"] or "stream_reduce" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-14:]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:10.1f}  q={r.get("Queue_Id", "?")} grid={r.get("Grid_Size", "?")} '
          f'{r["Kernel_Name"][26:90]}')
