#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python scripts/dbg_overfit_nan.py 800 > gpurun_out/dbg_overfit_nan.log 2>&1; tail -30 gpurun_out/dbg_overfit_nan.log
