#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or fused or segmented or gradients" 2>&1 | tail -6) > gpurun_out/t_split.log 2>&1
for v in 0 1; do
  SB_NO_ROLE_SPLIT=$v timeout 600 python bench.py --workload small --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/ab_small_split$v.jsonl
done
python - <<'PY'
import json
for v in (0, 1):
    d = json.loads(open(f"gpurun_out/ab_small_split{v}.jsonl").read())
    print("NO_ROLE_SPLIT", v, round(d["value"], 1), "compact", round(d["compact_bptt"]["value"], 1))
    for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:3]:
        print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
    for k, x in sorted(d["compact_bptt"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:3]:
        print(f"   C {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
tail -4 gpurun_out/t_split.log
