#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/sweep.txt
run() { timeout 300 python bench.py --workload big --no-cpu-baseline --no-exact --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$1', round(d['value'],1), round(d['ms_per_step'],3))" >> gpurun_out/sweep.txt; }
for r in 1 2; do
  run "default"
  for s in 16 24 48 64; do SB_BWD_CROSS_SLAB=$s run "bwd_cross_slab=$s"; done
  for s in 16 64; do SB_FWD_OVERLAP_SLAB=$s run "fwd_slab=$s"; done
done
cat gpurun_out/sweep.txt
