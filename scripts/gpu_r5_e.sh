#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 500 python scripts/stress_train_loop.py --epochs 600 > gpurun_out/stress_a.log 2>&1 ); tail -3 gpurun_out/stress_a.log
( timeout 500 python scripts/stress_train_loop.py --epochs 300 --sleep 100 > gpurun_out/stress_b.log 2>&1 ); tail -3 gpurun_out/stress_b.log
( timeout 400 python scripts/stress_train_loop.py --epochs 120 --loader > gpurun_out/stress_c.log 2>&1 ); tail -3 gpurun_out/stress_c.log
