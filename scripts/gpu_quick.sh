#!/bin/bash
# developer run: parity tests + forward-only and training throughput of both workloads (no CPU legs, no sibling)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8) > gpurun_out/t_quick.log 2>&1
for wl in big small; do
  timeout 300 python bench.py --workload $wl --forward-only --steps 10 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/q_${wl}_fwd.jsonl
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-exact --steps 6 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/q_${wl}_train.jsonl
done
python - <<'PY'
import json
for wl in ("big", "small"):
    f = json.loads(open(f"gpurun_out/q_{wl}_fwd.jsonl").read())
    d = json.loads(open(f"gpurun_out/q_{wl}_train.jsonl").read())
    print(wl, "fwd", round(f["value"], 1), "train", round(d["value"], 1))
    for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:8]:
        print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
tail -5 gpurun_out/t_quick.log
