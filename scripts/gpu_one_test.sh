#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest "$@" -m gpu -q --timeout 600 -x 2>&1 | grep -v "^$" | cut -c1-260 | tail -80) > gpurun_out/one_test.log 2>&1
cat gpurun_out/one_test.log
