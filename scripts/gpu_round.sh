#!/bin/bash
# one GPU-box session: tests, the default bench run (all lines), rocprof kernel trace, PMC passes
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/tests.log 2>&1
(timeout 600 python bench.py 2>&1 | grep '^{') > gpurun_out/bench_lines.jsonl 2> gpurun_out/bench.err
bash scripts/gpu_prof.sh > /dev/null 2>&1
bash scripts/gpu_pmc.sh small > /dev/null 2>&1
bash scripts/gpu_pmc.sh big > /dev/null 2>&1
ls gpurun_out
