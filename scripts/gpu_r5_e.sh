#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { n=$1; shift; ( env "$@" timeout 420 python scripts/stress_train_loop.py --epochs 1200 > gpurun_out/stress_$n.log 2>&1 ); echo "== $n"; grep -c TRIP gpurun_out/stress_$n.log; tail -1 gpurun_out/stress_$n.log; }
run base A=1
run pollrmw SB_LIB_PATH=$R/sound_bubble_amd/lib/exp/lib_pollrmw.so
run nodefer SB_NO_DEFERRED_REDUCE=1
run base2 A=1
