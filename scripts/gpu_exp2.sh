#!/bin/bash
# experiment: cross-pass consumer with / without its prologue loop (variant nopro: -DSB_EXP_CONS=1), producer || consumer and consumer alone
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" nopro; do
  echo "== variant '$v'"
  if [ -n "$v" ]; then export SB_LIB_PATH="$R/sound_bubble_amd/lib/exp/lib_$v.so"; else unset SB_LIB_PATH; fi
  timeout 600 python scripts/exp_cross_consume.py 2>&1 | grep "sync-between"
done
