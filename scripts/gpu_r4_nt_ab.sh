#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/nt_ab.txt
for r in 1 2 3; do
for v in main nont; do
  if [ $v = main ]; then unset SB_LIB_VARIANT; else export SB_LIB_VARIANT=$v; fi
  for wl in small big; do
  timeout 300 python scripts/bench_variant.py --workload $wl --no-cpu-baseline --no-exact --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$v $wl', round(d['value'],1), round(d['ms_per_step'],3))" >> gpurun_out/nt_ab.txt
  done
done
done
cat gpurun_out/nt_ab.txt
