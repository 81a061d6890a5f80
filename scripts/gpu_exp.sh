#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python "$@" > gpurun_out/exp.log 2>&1; tail -40 gpurun_out/exp.log
