#!/bin/bash
# round 4: wide gate recomputation for the inter-frame pass -- parity + same-box A/B of the big train step
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -k "gate_recompute or wide-recompute or (wide_overlapped_inter and 32)" 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -60) > gpurun_out/r4g_tests.log 2>&1
tail -8 gpurun_out/r4g_tests.log
for v in on off on off; do
  if [ $v = off ]; then export SB_NO_INTER_GATE_RECOMPUTE=1; else unset SB_NO_INTER_GATE_RECOMPUTE; fi
  timeout 600 python bench.py --workload big --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>gpurun_out/r4g_$v.err | grep '^{' > gpurun_out/r4g_$v.jsonl
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4g_$v.jsonl").read())
print("recompute $v: big train", round(d["value"], 1), "utt/s", round(d["ms_per_step"], 2), "ms", d["schedules"]["per_rank"])
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:7]:
    print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
done
