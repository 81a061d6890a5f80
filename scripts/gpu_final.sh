#!/bin/bash
# end of round: the default bench run (all lines) + rocprofv3 kernel-trace stats of the headline command
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(time (timeout 900 python bench.py 2>gpurun_out/bench.err | grep '^{' > gpurun_out/bench_lines.jsonl)) > gpurun_out/bench_time.log 2>&1
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_big_wide" -o k -- python "$R/bench.py" --steps 3 --warmup 1 --workload big --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_big_wide.log" 2>&1
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_small_wide" -o k -- python "$R/bench.py" --steps 5 --warmup 2 --workload small --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_small_wide.log" 2>&1
cd "$R"; cp "$(find gpurun_out/prof_big_wide -name '*kernel_stats.csv' | head -1)" gpurun_out/train_big_wide_kernel_stats.csv
cp "$(find gpurun_out/prof_small_wide -name '*kernel_stats.csv' | head -1)" gpurun_out/train_small_wide_kernel_stats.csv
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*.db" -delete
tail -2 gpurun_out/bench_time.log; head -5 gpurun_out/train_big_wide_kernel_stats.csv | cut -c1-160
