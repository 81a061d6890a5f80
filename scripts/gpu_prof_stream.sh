#!/bin/bash
# rocprofv3 kernel trace of the graphed streaming loop (small config): which launches make up one chunk
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stream" -o st -- python "$R/bench.py" --stream --workload small > "$R/gpurun_out/prof_stream.log" 2>&1
cd "$R"
f=$(find gpurun_out/prof_stream -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/stream_small_kernel_stats.csv
t=$(find gpurun_out/prof_stream -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-400:]
# one chunk = the span between two front-end STFT launches near the end of the run
names = [r["Kernel_Name"][:90] for r in tail]
t0 = [int(r["Start_Timestamp"]) for r in tail]; t1 = [int(r["End_Timestamp"]) for r in tail]
first = names[-1]
idx = [i for i, n in enumerate(names) if n == names[-1]]
a, b = idx[-2] + 1, idx[-1] + 1
print("launches per chunk:", b - a, " chunk span us:", (t1[b - 1] - t0[a]) / 1e3, " busy us:", sum(t1[i] - t0[i] for i in range(a, b)) / 1e3)
for i in range(a, b):
    print(f"{(t0[i] - t0[a]) / 1e3:8.1f} {(t1[i] - t0[i]) / 1e3:6.1f}  {names[i]}")
PY
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*.db" -delete
tail -3 gpurun_out/prof_stream.log
