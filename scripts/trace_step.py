"""every kernel of the LAST train step in a rocprofv3 kernel trace, in start order, with the idle time in front of it on the
whole device (developer tool): where a step's wall time goes that no kernel covers"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a step starts at the features kernel of the front end
starts = [i for i, r in enumerate(rows) if "features_kernel" in r["Kernel_Name"]]
bw = [i for i, r in enumerate(rows) if "lstm_bwd_rec_bf_kernel" in r["Kernel_Name"]]
if bw:
    i0 = max(i for i in starts if i < bw[-1])                 # the last step that has a backward (a parity forward may follow)
    i1 = min([i for i in starts if i > bw[-1]] + [len(rows)])
    adam = [i for i in range(i0, i1) if "adam" in rows[i]["Kernel_Name"]]
    if adam: i1 = adam[-1] + 1
else:                                                         # forward-only run: the last complete forward
    i0, i1 = starts[-2], starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
busy_end = t0
idle = 0
tot = {}
for r in rows[i0:i1]:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - busy_end)
    idle += gap
    busy_end = max(busy_end, e)
    key = nm[:60]
    tot[key] = tot.get(key, 0) + (e - s)
    print(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} idle_before={gap/1e3:6.1f} q={r.get('Queue_Id','?')} {nm[:100]}")
print(f"step span {(busy_end-t0)/1e3:.1f} us, device idle inside it {idle/1e3:.1f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{v/1e3:9.1f} us  {k}")
