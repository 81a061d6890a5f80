#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" plainst nosig; do SB_LIB_VARIANT=$v timeout 300 python scripts/exp_cross_k1.py 2>&1 | grep "^variant"; done
