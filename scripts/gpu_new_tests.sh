#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_distributed.py tests/test_eval_samples.py -m gpu -q -s -k "weight_forms or side_stream or fall_back or contending or other_radii" 2>&1 | grep -v "^$" | tail -60) > gpurun_out/tests_new.log 2>&1
tail -40 gpurun_out/tests_new.log
