#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest "tests/test_gpu_parity.py::test_overlapped_forward_hands_items_back_when_the_producer_stands_still" tests/test_gpu_records.py tests/test_cabi_cpu.py -q --tb=short -s 2>&1 | grep -v "^$" | cut -c1-300 | tail -12 ) > gpurun_out/r5k_tests.log 2>&1; cat gpurun_out/r5k_tests.log
(time (timeout 1200 python bench.py 2>gpurun_out/r05b_bench.err | grep '^{' > gpurun_out/r05b_bench_lines.jsonl)) > gpurun_out/r05b_bench_time.log 2>&1; tail -4 gpurun_out/r05b_bench_time.log
python - <<'PY'
import json
ls=[json.loads(l) for l in open("gpurun_out/r05b_bench_lines.jsonl")]
for d in ls: print(d["config"]["workload"][:28], round(d["value"], 1), d.get("ms_per_step"))
r=ls[-1]["roofline"]; print(r["kernel"][:60], "frac", r["frac"], "hbm_counter", r.get("frac_hbm_counter"), "traffic", r.get("traffic"), r.get("traffic_source","")[:40], "mfma", r.get("mfma_busy"), "pass", r["pass_level"]["frac"], r["pass_level"].get("frac_hbm_counter"), r["pass_level"].get("traffic_bytes_per_pass"))
PY
