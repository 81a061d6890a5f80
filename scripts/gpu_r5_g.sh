#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_gpu_records.py -q 2>&1 | grep -v "^$" | head -60 ) > gpurun_out/r5g_rec.log 2>&1; tail -3 gpurun_out/r5g_rec.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "recompute or two_product" 2>&1 | tail -15 ) > gpurun_out/r5g_tests.log 2>&1; tail -5 gpurun_out/r5g_tests.log
for i in 1 2 3; do
( SB_LIB_PATH=$R/sound_bubble_amd/lib/exp/lib_tripdbg.so timeout 420 python scripts/stress_train_loop.py --epochs 1500 > gpurun_out/stress_dbg$i.log 2>&1 ); echo "== stress dbg $i"; grep -c TRIP gpurun_out/stress_dbg$i.log; grep TRIP gpurun_out/stress_dbg$i.log | head -3 | cut -c1-900; tail -1 gpurun_out/stress_dbg$i.log
done
