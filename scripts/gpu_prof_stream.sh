#!/bin/bash
# rocprofv3 kernel trace of the graphed streaming loop (small + big config): per-kernel stats and the launches of one chunk
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for wl in small big; do
  cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stream_$wl" -o st -- python "$R/bench.py" --stream --workload $wl > "$R/gpurun_out/prof_stream_$wl.log" 2>&1
  cd "$R"
  cp "$(find gpurun_out/prof_stream_$wl -name '*kernel_stats.csv' | head -1)" gpurun_out/stream_${wl}_kernel_stats.csv
  python - "$(find gpurun_out/prof_stream_$wl -name '*kernel_trace.csv' | head -1)" $wl > gpurun_out/stream_${wl}_chunk_launches.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
t0 = [int(r["Start_Timestamp"]) for r in rows]; t1 = [int(r["End_Timestamp"]) for r in rows]
idx = [i for i, n in enumerate(names) if "features_kernel" in n]
a, b = idx[-3], idx[-2]
print(f"streaming chunk step, {sys.argv[2]} config, under rocprofv3 --kernel-trace (serialising: durations >= ~4.5 us per launch;")
print("the un-profiled period is the bench line's ms_per_step).  One chunk = the launches between two features_kernel launches:")
print("launches per chunk:", b - a, " period us:", (t0[b] - t0[a]) / 1e3, " busy us:", round(sum(t1[i] - t0[i] for i in range(a, b)) / 1e3, 1))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"at::native::", "", n); return n[:100]
for i in range(a, b):
    print(f"{(t0[i] - t0[a]) / 1e3:8.1f} {(t1[i] - t0[i]) / 1e3:6.1f}  {short(names[i])}")
PY
  head -4 gpurun_out/stream_${wl}_chunk_launches.txt
done
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*.db" -delete
