#!/bin/bash
# round 4, first call: the new launcher / RCCL tests + a same-tree baseline of the big train step
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_distributed.py -m gpu -q --timeout 900 -k "rccl or bench" -s 2>&1 | grep -v "^$" | tail -60) > gpurun_out/r4a_tests.log 2>&1
timeout 600 python bench.py --workload big --no-cpu-baseline --steps 10 --warmup 3 2>gpurun_out/r4a_big.err | grep '^{' > gpurun_out/r4a_big.jsonl
timeout 300 python bench.py --workload big --forward-only --steps 10 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/r4a_big_fwd.jsonl
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4a_big.jsonl").read())
print("big train", round(d["value"], 1), d["schedules"], d["rccl"])
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:10]:
    print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
f = json.loads(open("gpurun_out/r4a_big_fwd.jsonl").read())
print("big fwd", round(f["value"], 1))
PY
tail -15 gpurun_out/r4a_tests.log
