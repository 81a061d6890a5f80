#!/bin/bash
# round 4 checkpoint: whole -m gpu suite + smoke, the default bench run (all lines), rocprofv3 kernel stats of the headline command
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 3000 python -m pytest tests -m gpu -q --timeout 1200 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -120) > gpurun_out/r4_all_tests.log 2>&1
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r4_smoke.log 2>&1
(time (timeout 900 python bench.py 2>gpurun_out/bench.err | grep '^{' > gpurun_out/bench_lines.jsonl)) > gpurun_out/bench_time.log 2>&1
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_big_wide" -o k -- python "$R/bench.py" --steps 3 --warmup 1 --workload big --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_big_wide.log" 2>&1
cd "$R"; cp "$(find gpurun_out/prof_big_wide -name '*kernel_stats.csv' | head -1)" gpurun_out/train_big_wide_kernel_stats.csv
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*.db" -delete
tail -8 gpurun_out/r4_all_tests.log; cat gpurun_out/r4_smoke.log; tail -3 gpurun_out/bench_time.log; tail -3 gpurun_out/bench.err
python - <<'PY'
import json
L = [json.loads(l) for l in open("gpurun_out/bench_lines.jsonl")]
d = L[-1]
print("HEADLINE", d["metric"], round(d["value"], 1), d.get("schedules"))
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:14]:
    print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in d.get("secondary", {}).items()})
PY
