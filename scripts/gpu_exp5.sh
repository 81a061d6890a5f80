#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/exp_cross_k1.py 2>&1 | grep "^variant"
timeout 300 python scripts/exp_cross_consume.py 2>&1 | grep "sync-between"
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "cross_pass" 2>&1 | tail -3)
