#!/bin/bash
# round 4: whole suite, then same-box A/B of the forward kernels: two-stage loader (new) against the one-stage one (lib/exp/lib_oldfwd.so)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 3000 python -m pytest tests -m gpu -q --timeout 1200 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -60) > gpurun_out/r4f_tests.log 2>&1
tail -12 gpurun_out/r4f_tests.log
for v in new old new old; do
  if [ $v = old ]; then export SB_LIB_VARIANT=oldfwd; else unset SB_LIB_VARIANT; fi
  for wl in big small; do
    timeout 300 python scripts/bench_variant.py --workload $wl --forward-only --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/r4f_${v}_${wl}_fwd.jsonl
    timeout 600 python scripts/bench_variant.py --workload $wl --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>gpurun_out/r4f_${v}_$wl.err | grep '^{' > gpurun_out/r4f_${v}_$wl.jsonl
    python - <<PY
import json
f = json.loads(open("gpurun_out/r4f_${v}_${wl}_fwd.jsonl").read().strip().split("\n")[-1])
d = json.loads(open("gpurun_out/r4f_${v}_$wl.jsonl").read().strip().split("\n")[-1])
print("$v $wl fwd", round(f["value"], 1), " train", round(d["value"], 1), "utt/s", round(d["ms_per_step"], 2), "ms")
for k, x in sorted(f["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:4]:
    print(f"   fwd   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:8]:
    if "fwd" in k: print(f"   train {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
  done
done
