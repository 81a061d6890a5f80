#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/exp_fwd_nogates.py > gpurun_out/r4b_nogates.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_distributed.py -m gpu -q --timeout 900 -k "rccl" -s 2>&1 | grep -v "^$" | tail -12) > gpurun_out/r4b_tests.log 2>&1
cat gpurun_out/r4b_nogates.log; tail -6 gpurun_out/r4b_tests.log
