#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/defer_ab.txt
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "deferred or cross_pass or deterministic or full_size or train_step or adam or trajectory" 2>&1 | tail -30) > gpurun_out/defer_tests.log 2>&1
timeout 300 python scripts/find_fills.py > gpurun_out/find_fills.txt 2>&1
for i in 1 2; do
  SB_NO_DEFERRED_REDUCE=1 timeout 300 python bench.py --workload big --steps 20 --warmup 5 --no-cpu-baseline --no-exact 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('main-stream reductions', round(d['value'],1), d['ms_per_step'])" >> gpurun_out/defer_ab.txt
  timeout 300 python bench.py --workload big --steps 20 --warmup 5 --no-cpu-baseline --no-exact 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('deferred', round(d['value'],1), d['ms_per_step'])" >> gpurun_out/defer_ab.txt
done
rm -rf gpurun_out/prof_trace
cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_trace" -o big -- python "$R/bench.py" --steps 3 --warmup 1 --workload big --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_trace.log" 2>&1
cd "$R"; python scripts/trace_step.py gpurun_out/prof_trace > gpurun_out/r4_step_trace_defer.txt 2>&1
find gpurun_out -name "*.db" -delete
grep -v "^  \|^$\|Warning\|warn" gpurun_out/defer_tests.log | tail -8; cat gpurun_out/defer_ab.txt; head -40 gpurun_out/find_fills.txt
