#!/bin/bash
# one GPU-box session: tests, bench (small/big, train/forward), rocprof kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/tests.log 2>&1
(timeout 400 python bench.py --steps 10 --warmup 3 2>&1 | tail -2) > gpurun_out/bench_small.log 2>&1
(timeout 300 python bench.py --steps 5 --warmup 2 --workload big --no-cpu-baseline 2>&1 | tail -2) > gpurun_out/bench_big.log 2>&1
(timeout 200 python bench.py --steps 10 --warmup 3 --forward-only --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/fwd_small.log 2>&1
(timeout 200 python bench.py --steps 5 --warmup 2 --forward-only --workload big --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/fwd_big.log 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_small" -o small -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_small.log" 2>&1
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_big" -o big -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --workload big --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/prof_big.log" 2>&1
cd "$GRAFT_REPO_ROOT"; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
ls -la gpurun_out gpurun_out/prof_small 2>/dev/null | head -40
