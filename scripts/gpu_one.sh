#!/bin/bash
# usage: gpu_one.sh <pytest args...>  -> gpurun_out/one.log
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest "$@" -m gpu -q --timeout 900 2>&1 | grep -v "^$" | grep -v "^E             \*" | tail -300) > gpurun_out/one.log 2>&1
tail -5 gpurun_out/one.log
