#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { # name, env...
  n=$1; shift
  rm -rf /tmp/run_$n
  ( env "$@" timeout 400 python -m sound_bubble_amd.train_cli --config experiments/overfit_test_samples.json --run_dir /tmp/run_$n --epochs 80 > gpurun_out/cli_$n.log 2>&1 ); echo "$n rc=$? nan=$(grep -c 'Average Loss: nan' gpurun_out/cli_$n.log) abort=$(grep -c 'launch aborted' gpurun_out/cli_$n.log) first_nan_epoch=$(grep -n 'Average Loss' gpurun_out/cli_$n.log | grep -m1 nan)"
}
run d1 A=1
run d2 A=1
run d3 A=1
run d4 A=1
run nodefer1 SB_NO_DEFERRED_REDUCE=1
run nodefer2 SB_NO_DEFERRED_REDUCE=1
run nocross1 SB_NO_BWD_CROSS_OVERLAP=1
run nocross2 SB_NO_BWD_CROSS_OVERLAP=1
