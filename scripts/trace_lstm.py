"""print start / end (us, relative) of the recurrent kernels of the last train step in a rocprofv3 kernel trace"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sel = [r for r in rows if "lstm_" in r["Kernel_Name"]][-n:]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    nm = r["Kernel_Name"]
    nm = nm[nm.find("lstm_"):][:70]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:10.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:10.1f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f}  q={r.get("Queue_Id", "?")} {nm}')
