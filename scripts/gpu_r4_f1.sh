#!/bin/bash
# round 4: two-stage loader of the forward kernels (LayerNorm on the waves without a y tile): full suite + forward / train A/B lines
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 3000 python -m pytest tests -m gpu -q --timeout 1200 -x 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -60) > gpurun_out/r4f_tests.log 2>&1
tail -5 gpurun_out/r4f_tests.log
for wl in big small; do
  timeout 300 python bench.py --workload $wl --forward-only --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/r4f_${wl}_fwd.jsonl
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>gpurun_out/r4f_$wl.err | grep '^{' > gpurun_out/r4f_$wl.jsonl
  python - <<PY
import json
f = json.loads(open("gpurun_out/r4f_${wl}_fwd.jsonl").read().strip().split("\n")[-1])
print("$wl fwd", round(f["value"], 1))
d = json.loads(open("gpurun_out/r4f_$wl.jsonl").read().strip().split("\n")[-1])
print("$wl train", round(d["value"], 1), "utt/s", round(d["ms_per_step"], 2), "ms")
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:9]:
    print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
done
