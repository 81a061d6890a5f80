#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof_trace_small
cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_trace_small" -o s -- python "$R/bench.py" --steps 3 --warmup 1 --workload small --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_trace_small.log" 2>&1
cd "$R"; python scripts/trace_step.py gpurun_out/prof_trace_small > gpurun_out/r4_step_trace_small.txt 2>&1
rm -rf gpurun_out/prof_trace_small
tail -30 gpurun_out/r4_step_trace_small.txt
