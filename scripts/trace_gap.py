"""print every kernel between the end of one cross-pass consumer and the start of the next producer in the last train step of a
rocprofv3 kernel trace (developer tool): what sits in the gap between two blocks' backward"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
bw = [i for i, r in enumerate(rows) if "lstm_bwd_rec_bf_kernel" in r["Kernel_Name"]]
# last step: take the last 18 backward launches (6 blocks x producer + 2 consumer launches); window = after block 2's consumers to block 4's producer
sel = bw[-18:]
i0, i1 = sel[3], sel[9] + 1
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    nm = nm[:90]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:8.1f}  q={r.get('Queue_Id','?')} {nm}")
