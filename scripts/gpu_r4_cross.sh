#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -x -k "cross_pass" 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-400 | tail -60) > gpurun_out/r4x_tests.log 2>&1
tail -40 gpurun_out/r4x_tests.log
if grep -q "passed" gpurun_out/r4x_tests.log && ! grep -q "failed" gpurun_out/r4x_tests.log; then
for v in on off on off; do
  if [ $v = off ]; then export SB_NO_BWD_CROSS_OVERLAP=1; else unset SB_NO_BWD_CROSS_OVERLAP; fi
  timeout 600 python bench.py --workload big --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>gpurun_out/r4x_$v.err | grep '^{' > gpurun_out/r4x_$v.jsonl
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4x_$v.jsonl").read())
print("cross overlap $v: big train", round(d["value"], 1), "utt/s", round(d["ms_per_step"], 2), "ms", d["schedules"]["per_rank"])
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:7]:
    print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
  tail -3 gpurun_out/r4x_$v.err
done
fi
