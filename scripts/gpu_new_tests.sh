#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_torch_ops.py tests/test_gpu_switch_matrix.py -m gpu -q --durations=15 --timeout 900 2>&1 | grep -v "^$" | tail -400) > gpurun_out/new_tests.log 2>&1
tail -50 gpurun_out/new_tests.log
