#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_records.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q 2>&1 | tail -15 ) > gpurun_out/r5f_tests.log 2>&1; tail -5 gpurun_out/r5f_tests.log
( timeout 420 python scripts/stress_train_loop.py --epochs 1200 > gpurun_out/stress_arena.log 2>&1 ); echo "== stress arena"; grep -c TRIP gpurun_out/stress_arena.log; tail -1 gpurun_out/stress_arena.log
for v in q24 recf32 q24 recf32; do
  if [ $v = recf32 ]; then export SB_LIB_PATH=$R/sound_bubble_amd/lib/exp/lib_recf32.so; else unset SB_LIB_PATH; fi
  ( timeout 300 python bench.py --workload big --steps 30 --no-cpu-baseline --no-parity --no-exact 2>gpurun_out/r5f_bench_$v.err | tail -1 >> gpurun_out/r5f_bench_$v.jsonl )
  python - <<PY
import json
l=open("gpurun_out/r5f_bench_$v.jsonl").read().strip().splitlines()[-1]
d=json.loads(l); print("$v", d["value"], d["ms_per_step"])
PY
done
