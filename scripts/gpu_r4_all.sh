#!/bin/bash
# the whole -m gpu suite, as the driver runs it, + smoke
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 3000 python -m pytest tests -m gpu -q --timeout 1200 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -120) > gpurun_out/r4_all_tests.log 2>&1
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r4_smoke.log 2>&1
tail -15 gpurun_out/r4_all_tests.log; cat gpurun_out/r4_smoke.log
