#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_records.py -q 2>&1 | grep "^E" | cut -c1-300 | head -30 ) > gpurun_out/r5g_rec.log 2>&1; cat gpurun_out/r5g_rec.log | head -20
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "overlapped_forward or ragged or goldens" 2>&1 | tail -5 ) > gpurun_out/r5g_tests.log 2>&1; tail -3 gpurun_out/r5g_tests.log
run() { n=$1; ( timeout 600 python scripts/stress_train_loop.py --epochs 1700 > gpurun_out/stress_$n.log 2>&1 ); echo "== stress $n"; grep -c TRIP gpurun_out/stress_$n.log; grep "TRIP\|SLOW\|give-ups" gpurun_out/stress_$n.log | head -8 | cut -c1-500; tail -1 gpurun_out/stress_$n.log | cut -c1-300; }
run hold1
run hold2
run hold3
