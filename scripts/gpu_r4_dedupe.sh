#!/bin/bash
# round 4: one prologue per tile in the cross-pass consumer (claim / done / look-ahead): parity, then the experiment script and bench
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -k "cross_pass or full_size or deterministic" 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -30) > gpurun_out/r4j_tests.log 2>&1
tail -5 gpurun_out/r4j_tests.log
timeout 600 python scripts/exp_cross_consume.py 2>&1 | grep "sync-between"
for i in 1 2; do
timeout 600 python bench.py --workload big --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>gpurun_out/r4j_big.err | grep '^{' > gpurun_out/r4j_big.jsonl
python - <<PY
import json
d = json.loads(open("gpurun_out/r4j_big.jsonl").read().strip().split("\n")[-1])
print("big train", round(d["value"], 1), "utt/s", round(d["ms_per_step"], 2), "ms", d["schedules"]["per_rank"])
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:6]:
    print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
done
