#!/bin/bash
# round 4: forward kernels with the wide two-pass LayerNorm on the loader waves + hs-free inter-frame pass: suite, bench lines, phase table
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 3000 python -m pytest tests -m gpu -q --timeout 1200 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -60) > gpurun_out/r4h_tests.log 2>&1
tail -8 gpurun_out/r4h_tests.log
for wl in big small; do
  timeout 300 python bench.py --workload $wl --forward-only --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/r4h_${wl}_fwd.jsonl
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>gpurun_out/r4h_$wl.err | grep '^{' > gpurun_out/r4h_$wl.jsonl
  python - <<PY
import json
f = json.loads(open("gpurun_out/r4h_${wl}_fwd.jsonl").read().strip().split("\n")[-1])
d = json.loads(open("gpurun_out/r4h_$wl.jsonl").read().strip().split("\n")[-1])
print("$wl fwd", round(f["value"], 1), " train", round(d["value"], 1), "utt/s", round(d["ms_per_step"], 2), "ms")
for k, x in sorted(f["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:4]:
    print(f"   fwd   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:8]:
    print(f"   train {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
done
SB_LIB_VARIANT=phase timeout 600 python scripts/phase_timing_train.py > gpurun_out/r4_phase_timing.txt 2>&1
grep -v "compact" gpurun_out/r4_phase_timing.txt | grep -B1 -A4 "inference\|train, wide   " | head -90
