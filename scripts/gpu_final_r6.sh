#!/bin/bash
# final pass of round 6 after the last kernel change: GPU suite, default bench run, counter passes (stamped with the tree's digest)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v "^$" | tail -6) > gpurun_out/r06_final_gpu_tests.log 2>&1; tail -2 gpurun_out/r06_final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/r06_final_gpu_tests.log
bash scripts/gpu_pmc.sh big _wide > gpurun_out/r06_pmc_big.log 2>&1; tail -1 gpurun_out/r06_pmc_big.log
bash scripts/gpu_pmc.sh small _wide > gpurun_out/r06_pmc_small.log 2>&1; tail -1 gpurun_out/r06_pmc_small.log
cp gpurun_out/pmc_traffic_big_wide.json profiles/r06_pmc_traffic_big_wide.json; cp gpurun_out/pmc_sq_big_wide.json profiles/r06_pmc_sq_big_wide.json
cp gpurun_out/pmc_traffic_small_wide.json profiles/r06_pmc_traffic_small_wide.json; cp gpurun_out/pmc_sq_small_wide.json profiles/r06_pmc_sq_small_wide.json
prof() { cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$1" -o k -- python "$R/bench.py" $2 --no-cpu-baseline --no-exact --no-parity > "$R/gpurun_out/prof_$1.log" 2>&1
  cd "$R"; f=$(find gpurun_out/prof_$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_train_$1_kernel_stats.csv; }
prof big_wide "--steps 3 --warmup 1 --workload big"
prof small_wide "--steps 5 --warmup 2 --workload small"
prof big_attn "--steps 3 --warmup 1 --workload big-attn"
( timeout 300 python bench.py --workload big-attn --steps 10 --no-cpu-baseline --no-exact --no-parity 2>/dev/null | grep '^{' > gpurun_out/r06_bench_big_attn.jsonl )
( timeout 300 python bench.py --workload big-attn --forward-only --steps 20 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/r06_bench_big_attn.jsonl )
(time (timeout 1200 python bench.py 2>gpurun_out/r06_final_bench.err | grep '^{' > gpurun_out/r06_final_bench_lines.jsonl)) > gpurun_out/r06_final_bench_time.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r06_final_bench_lines.jsonl"):
    d = json.loads(l); print(d["metric"][:40], d["config"]["workload"][:28], round(d["value"], 1), d.get("ms_per_step"), d.get("roofline", {}).get("stale_profile"))
PY
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*counter_collection.csv" -size +20M -delete; find gpurun_out -name "*.db" -delete
