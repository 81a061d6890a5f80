#!/bin/bash
# end of round: the whole -m gpu suite, smoke(), and the default bench run of the final tree
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v "^$" | tail -8) > gpurun_out/r05_gpu_tests_final.log 2>&1; tail -3 gpurun_out/r05_gpu_tests_final.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3) > gpurun_out/r05_smoke.log; cat gpurun_out/r05_smoke.log
(time (timeout 1200 python bench.py 2>gpurun_out/r05c_bench.err | grep '^{' > gpurun_out/r05c_bench_lines.jsonl)) > gpurun_out/r05c_bench_time.log 2>&1; tail -4 gpurun_out/r05c_bench_time.log
python - <<'PY'
import json
ls=[json.loads(l) for l in open("gpurun_out/r05c_bench_lines.jsonl")]
for d in ls: print(d["config"]["workload"][:28], round(d["value"], 1), d.get("ms_per_step"))
print(ls[-1]["dtype"][:200])
PY
