#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_records.py -q 2>&1 | grep "^E" | cut -c1-300 | head -30 ) > gpurun_out/r5g_rec.log 2>&1; cat gpurun_out/r5g_rec.log | head -20
run() { n=$1; lib=$2; ( SB_LIB_PATH=$R/sound_bubble_amd/lib/exp/lib_$lib.so timeout 600 python scripts/stress_train_loop.py --epochs 1700 > gpurun_out/stress_$n.log 2>&1 ); echo "== stress $n"; grep -c TRIP gpurun_out/stress_$n.log; grep "TRIP\|SLOW" gpurun_out/stress_$n.log | head -6 | cut -c1-600; tail -2 gpurun_out/stress_$n.log | cut -c1-600; }
run long1 longwait
run long2 longwait
run long3 longwait
