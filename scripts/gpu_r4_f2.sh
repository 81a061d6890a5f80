#!/bin/bash
# round 4: hs-free inter-frame pass (h recomputed from the records) -- parity tests, bench lines, forward phase table of the instrumented build
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -k "cross_pass or wide_fused_bptt_single or full_size or deterministic" 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -40) > gpurun_out/r4g_tests.log 2>&1
tail -6 gpurun_out/r4g_tests.log
for wl in big; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>gpurun_out/r4g_$wl.err | grep '^{' > gpurun_out/r4g_$wl.jsonl
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4g_$wl.jsonl").read().strip().split("\n")[-1])
print("$wl train", round(d["value"], 1), "utt/s", round(d["ms_per_step"], 2), "ms", d["schedules"]["per_rank"])
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:9]:
    print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
done
SB_LIB_VARIANT=phase timeout 600 python scripts/phase_timing_train.py > gpurun_out/r4_phase_timing.txt 2>&1
grep -v "compact" gpurun_out/r4_phase_timing.txt | tail -60
