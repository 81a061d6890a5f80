#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or overlapped" 2>&1 | tail -12) > gpurun_out/t_wide.log 2>&1
(timeout 600 python bench.py --workload big --no-cpu-baseline --steps 6 --warmup 2 2>&1 | grep '^{') > gpurun_out/w1_big.jsonl 2> gpurun_out/w1.err
tail -6 gpurun_out/t_wide.log; tail -3 gpurun_out/w1.err
