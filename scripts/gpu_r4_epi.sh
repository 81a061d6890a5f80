#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -30) > gpurun_out/r4i_tests.log 2>&1
tail -4 gpurun_out/r4i_tests.log
bash scripts/gpu_ab_variants.sh "noepi main" 2 train
