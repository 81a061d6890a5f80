#!/bin/bash
# experiment: non-temporal hint on the blocked record stores / loads (variant library built with -DSB_EXP_NT_RECORDS)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 900 -k "gate_recompute" 2>&1 | tail -3) > gpurun_out/r4n_tests.log 2>&1
tail -3 gpurun_out/r4n_tests.log
for v in base nt base nt; do
  if [ $v = nt ]; then export SB_LIB_VARIANT=ntside; else unset SB_LIB_VARIANT; fi
  timeout 600 python scripts/bench_variant.py --workload big --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>gpurun_out/r4n_$v.err | grep '^{' > gpurun_out/r4n_$v.jsonl
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4n_$v.jsonl").read())
print("$v: big train", round(d["value"], 1), "utt/s", round(d["ms_per_step"], 2), "ms")
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:7]:
    print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
done
