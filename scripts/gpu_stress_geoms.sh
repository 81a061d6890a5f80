#!/bin/bash
# the harness epoch loop on cached synthetic batches at the BENCHED geometries: small B = 32 (time-segmented passes), big B = 16 (145-tile overlapped passes)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 500 python scripts/stress_train_loop.py --config bubble_small_synthetic.json --batch 32 --epochs 4000 > gpurun_out/stress_small32.log 2>&1 ); grep "TRIP\|SLOW\|give-ups" gpurun_out/stress_small32.log | head -5 | cut -c1-300; tail -1 gpurun_out/stress_small32.log | cut -c1-220
( timeout 560 python scripts/stress_train_loop.py --config bubble_big_synthetic.json --batch 16 --epochs 2500 > gpurun_out/stress_big16.log 2>&1 ); grep "TRIP\|SLOW\|give-ups" gpurun_out/stress_big16.log | head -5 | cut -c1-300; tail -1 gpurun_out/stress_big16.log | cut -c1-220
