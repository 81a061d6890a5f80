#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/fwd_ab.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -k "overlap or deterministic or full_size or forward or streaming" 2>&1 | tail -5) > gpurun_out/fwd_tests.log 2>&1
for i in 1 2 3; do
  SB_NO_DEFERRED_REDUCE=1 timeout 300 python bench.py --workload big --forward-only --steps 30 --warmup 5 --no-cpu-baseline --no-exact 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('memset per producer', round(d['value'],1), d['ms_per_step'])" >> gpurun_out/fwd_ab.txt
  timeout 300 python bench.py --workload big --forward-only --steps 30 --warmup 5 --no-cpu-baseline --no-exact 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('pooled flags', round(d['value'],1), d['ms_per_step'])" >> gpurun_out/fwd_ab.txt
done
tail -3 gpurun_out/fwd_tests.log; cat gpurun_out/fwd_ab.txt
