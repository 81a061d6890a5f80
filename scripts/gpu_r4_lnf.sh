#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/exp_ln_film.txt
for r in 1 2; do
for v in main nt1 nt3 c50 c13 nt1c50; do
  if [ $v = main ]; then unset SB_LIB_VARIANT; else export SB_LIB_VARIANT=$v; fi
  timeout 200 python scripts/exp_ln_film.py 2>&1 | grep variant >> gpurun_out/exp_ln_film.txt
done
done
cat gpurun_out/exp_ln_film.txt
