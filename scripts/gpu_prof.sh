#!/bin/bash
# rocprofv3 kernel-trace summaries of the bench train step, small + big workloads (the same commands bench.py's
# roofline object describes: --workload <wl> prints that single train line)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_small" -o small -- python "$R/bench.py" --steps 5 --warmup 2 --workload small --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_small.log" 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_big" -o big -- python "$R/bench.py" --steps 3 --warmup 1 --workload big --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_big.log" 2>&1
cd "$R"; find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*.db" -delete
