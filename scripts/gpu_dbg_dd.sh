#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in look18 look24 look18 look24; do
echo "== $v"
SB_LIB_PATH="$R/sound_bubble_amd/lib/exp/lib_$v.so" timeout 600 python scripts/exp_cross_consume.py 2>&1 | grep "sync-between" | tail -3
done
echo "== main (32)"; timeout 600 python scripts/exp_cross_consume.py 2>&1 | grep "sync-between" | tail -3
