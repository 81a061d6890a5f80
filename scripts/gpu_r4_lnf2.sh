#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/exp_ln_film_step.txt
for r in 1 2 3; do
for v in main nt3 nt7 nt1; do
  if [ $v = main ]; then unset SB_LIB_VARIANT; else export SB_LIB_VARIANT=$v; fi
  timeout 300 python scripts/bench_variant.py --workload big --no-cpu-baseline --no-exact --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); ks=d['roofline']['kernels']; lf=[v['avg_launch_ms'] for k,v in ks.items() if 'ln_film' in k]; print('$v', round(d['value'],1), d['ms_per_step'], 'ln_film', lf)" >> gpurun_out/exp_ln_film_step.txt
done
done
cat gpurun_out/exp_ln_film_step.txt
