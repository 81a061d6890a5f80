#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_losses.py tests/test_gpu_film_bank.py tests/test_gpu_records.py "tests/test_gpu_switch_matrix.py::test_every_documented_switch_is_in_the_matrix" -q --tb=short 2>&1 | grep -v "^$" | cut -c1-260 | head -150 ) > gpurun_out/r5i_tests.log 2>&1; tail -5 gpurun_out/r5i_tests.log
