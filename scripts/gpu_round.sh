#!/bin/bash
# one GPU-box session: tests, the default bench run (all lines), rocprof kernel trace
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q -s -x --timeout 900 2>&1 | grep -v "^$" | tail -60) > gpurun_out/tests.log 2>&1
(time (timeout 900 python bench.py 2>gpurun_out/bench.err | grep '^{' > gpurun_out/bench_lines.jsonl)) > gpurun_out/bench_time.log 2>&1
tail -3 gpurun_out/tests.log; cat gpurun_out/bench_time.log; tail -3 gpurun_out/bench.err
