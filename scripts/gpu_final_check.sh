R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v "^$" | tail -8) > gpurun_out/r05_gpu_tests_final.log 2>&1; tail -3 gpurun_out/r05_gpu_tests_final.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3) > gpurun_out/r05_smoke.log; cat gpurun_out/r05_smoke.log
