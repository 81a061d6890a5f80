#!/bin/bash
# the three probes of the overlapped forward's stalled producer, ~10 000 B = 9 steps each (time-boxed).
# usage: gpu_stall_probes.sh <epochs>
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
N=${1:-10000}
run() { name=$1; shift; echo "=== $name: $* ==="; env "$@" timeout 900 python scripts/stress_train_loop.py --epochs $N --batch 9 2>&1 | grep -v "^schedules this epoch\|Current checkpoint\|^val/" | tail -40; }
{
run baseline_kfd SB_DUMMY=0
run xcd_exact SB_FWD_GUARD_XCD_EXACT=1
run side_low SB_SIDE_STREAM_PRIORITY=low
} > gpurun_out/r06_stall_probes.txt 2>&1
tail -5 gpurun_out/r06_stall_probes.txt
