#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1700 python -m pytest "$@" -m gpu -q --timeout 900 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-400 | tail -150) > gpurun_out/r4c_tests.log 2>&1
tail -150 gpurun_out/r4c_tests.log
