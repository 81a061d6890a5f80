#!/bin/bash
# kernel timeline of the big train step (last step): start / end of the recurrent kernels per hardware queue, and everything in one gap
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_trace
cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_trace" -o big -- python "$R/bench.py" --steps 3 --warmup 1 --workload big --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_trace.log" 2>&1
cd "$R"; python scripts/trace_lstm.py gpurun_out/prof_trace 40 > gpurun_out/r4_trace.txt 2>&1
python scripts/trace_gap.py gpurun_out/prof_trace > gpurun_out/r4_trace_gap.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*.db" -delete
head -30 gpurun_out/r4_trace.txt; echo ----; cat gpurun_out/r4_trace_gap.txt
