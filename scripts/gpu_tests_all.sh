#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/tests_all.log 2>&1
tail -15 gpurun_out/tests_all.log
