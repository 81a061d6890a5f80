#!/bin/bash
# kernel timeline of one inference forward (big, small)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for WL in big small; do
rm -rf gpurun_out/prof_trace_fwd
cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_trace_fwd" -o f -- python "$R/bench.py" --steps 4 --warmup 2 --workload $WL --forward-only --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_trace_fwd_$WL.log" 2>&1
cd "$R"; python scripts/trace_step.py gpurun_out/prof_trace_fwd > gpurun_out/r4_fwd_trace_$WL.txt 2>&1
done
rm -rf gpurun_out/prof_trace_fwd
tail -12 gpurun_out/r4_fwd_trace_big.txt
