#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "overlapped_forward or ragged or goldens or full_size or fullsize or side_stream or fall_back or hands_items" -s 2>&1 | grep -v "^$" | tail -8 | cut -c1-250 ) > gpurun_out/r5l_tests.log 2>&1; cat gpurun_out/r5l_tests.log
rm -f gpurun_out/r5l_ab.txt
for v in default nooverlap default; do
  if [ $v = nooverlap ]; then export SB_NO_FWD_OVERLAP=1; else unset SB_NO_FWD_OVERLAP; fi
  timeout 300 python bench.py --workload big --forward-only --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fwd $v', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a gpurun_out/r5l_ab.txt
done
unset SB_NO_FWD_OVERLAP
for i in 1 2; do timeout 300 python bench.py --workload big --steps 30 --no-cpu-baseline --no-parity --no-exact 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('train default', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a gpurun_out/r5l_ab.txt; done
for i in 1 2; do ( timeout 600 python scripts/stress_train_loop.py --epochs 1700 > gpurun_out/stress_tile$i.log 2>&1 ); grep "TRIP\|SLOW\|give-ups" gpurun_out/stress_tile$i.log | head -6 | cut -c1-300; tail -1 gpurun_out/stress_tile$i.log | cut -c1-200; done
