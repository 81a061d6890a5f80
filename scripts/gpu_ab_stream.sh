#!/bin/bash
# same-box A/B of the streaming chunk loop: SB_* switch given as $1 (e.g. SB_NO_VEC_LSTM) off / on
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
SW=${1:-SB_NO_INFER_WORKSPACE}
for i in 1 2 3; do
  for v in 0 1; do
    for wl in small big; do
      env $SW=$v python bench.py --stream --workload $wl 2>/dev/null | grep "^{" | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$SW=$v $wl', round(d['value'], 1), 'chunks/s', round(d['ms_per_step'], 4), 'ms')"
    done
  done
done 2>&1 | tee gpurun_out/ab_stream.log
