#!/bin/bash
# same-box A/B of the help timeout of the overlapped forward's consumer (default ~2 ms vs the first value ~6 ms)
R="$GRAFT_REPO_ROOT"; cd "$R"; export TMPDIR=/tmp
for v in default help6ms default help6ms; do
  if [ $v = default ]; then unset SB_LIB_PATH; else export SB_LIB_PATH=$R/sound_bubble_amd/lib/exp/lib_$v.so; fi
  timeout 300 python bench.py --workload big --forward-only --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fwd $v', round(d['value'],1), round(d['ms_per_step'],3), d.get('ms_per_step_median'), d['schedules']['per_rank'][0].get('fwd_giveups_total'))"
  timeout 300 python bench.py --workload big --steps 30 --no-cpu-baseline --no-parity --no-exact 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('train $v', round(d['value'],1), round(d['ms_per_step'],3), d.get('ms_per_step_median'), d['schedules']['per_rank'][0].get('fwd_giveups_total'))"
done
