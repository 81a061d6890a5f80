#!/bin/bash
# round-3 baseline: default + exact-mode bench lines and a kernel trace of the exact-mode big train step
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 600 python bench.py --workload big --no-cpu-baseline 2>&1 | grep '^{') > gpurun_out/base_big.jsonl 2> gpurun_out/base.err
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_exact_big" -o big -- python "$R/bench.py" --steps 3 --warmup 1 --workload big --no-cpu-baseline --no-exact --exact > "$R/gpurun_out/prof_exact_big.log" 2>&1
cd "$R"; find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*.db" -delete
ls gpurun_out/prof_exact_big
