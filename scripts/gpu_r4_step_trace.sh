#!/bin/bash
# kernel timeline of the whole last big train step (every kernel, device idle time in front of each)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_trace
cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_trace" -o big -- python "$R/bench.py" --steps 3 --warmup 1 --workload big --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_trace.log" 2>&1
cd "$R"; python scripts/trace_step.py gpurun_out/prof_trace > gpurun_out/r4_step_trace.txt 2>&1
python scripts/trace_gap.py gpurun_out/prof_trace > gpurun_out/r4_trace_gap.txt 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete; find gpurun_out -name "*.db" -delete
tail -30 gpurun_out/r4_step_trace.txt; grep '^{' gpurun_out/prof_trace.log | cut -c1-200
