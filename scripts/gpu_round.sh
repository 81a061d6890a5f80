#!/bin/bash
# one GPU-box session: tests, bench (small/big train + forward + streaming + extras), rocprof kernel trace, PMC traffic
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/tests.log 2>&1
(timeout 400 python bench.py 2>&1 | tail -1) > gpurun_out/bench_small.log 2>&1
(timeout 400 python bench.py --workload big 2>&1 | tail -1) > gpurun_out/bench_big.log 2>&1
(timeout 200 python bench.py --forward-only --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/fwd_small.log 2>&1
(timeout 200 python bench.py --forward-only --workload big --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/fwd_big.log 2>&1
(timeout 200 python bench.py --stream 2>&1 | tail -1) > gpurun_out/stream_small.log 2>&1
(timeout 200 python bench.py --stream --workload big 2>&1 | tail -1) > gpurun_out/stream_big.log 2>&1
(timeout 300 python bench.py --workload big-attn --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_big_attn.log 2>&1
(timeout 400 python bench.py --no-cpu-baseline --vendor-gpu-baseline 2>&1 | tail -1) > gpurun_out/vend_small.log 2>&1
(timeout 400 python bench.py --workload big --no-cpu-baseline --vendor-gpu-baseline 2>&1 | tail -1) > gpurun_out/vend_big.log 2>&1
(timeout 200 python scripts/bench_lstm.py 2>&1 | tail -8) > gpurun_out/lstm_micro.log 2>&1
bash scripts/gpu_prof.sh > /dev/null 2>&1
bash scripts/gpu_pmc.sh small > /dev/null 2>&1
bash scripts/gpu_pmc.sh big > /dev/null 2>&1
python scripts/pmc_summary.py small "$R/gpurun_out" "$R/gpurun_out/pmc_traffic_small.json" > /dev/null 2>&1
python scripts/pmc_summary.py big "$R/gpurun_out" "$R/gpurun_out/pmc_traffic_big.json" > /dev/null 2>&1
cd "$R"; find gpurun_out -name "*.db" -delete; find gpurun_out -name "*kernel_trace.csv" -size +30M -delete
find gpurun_out -name "*counter_collection.csv" -size +20M -delete; ls gpurun_out
