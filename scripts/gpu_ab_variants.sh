#!/bin/bash
# same-box A/B of library variants (lib/exp/lib_<name>.so; "main" = the tree's library): usage gpu_ab_variants.sh "<v1> <v2> ..." <rounds> [train]
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
VARS="$1"; ROUNDS=${2:-2}; TRAIN=$3
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    if [ $v = main ]; then unset SB_LIB_VARIANT; else export SB_LIB_VARIANT=$v; fi
    for wl in big small; do
      timeout 300 python scripts/bench_variant.py --workload $wl --forward-only --steps 30 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/ab_${v}_${wl}_fwd.jsonl
      if [ -n "$TRAIN" ]; then timeout 600 python scripts/bench_variant.py --workload $wl --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/ab_${v}_$wl.jsonl; fi
      python - <<PY
import json, os
f = json.loads(open("gpurun_out/ab_${v}_${wl}_fwd.jsonl").read().strip().split("\n")[-1])
line = "$v $wl fwd %.1f" % f["value"]
ks = sorted(f["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:2]
line += "  [" + ", ".join("%.3f" % x["avg_launch_ms"] for _, x in ks) + "]"
if "$TRAIN":
    d = json.loads(open("gpurun_out/ab_${v}_$wl.jsonl").read().strip().split("\n")[-1])
    line += "   train %.1f" % d["value"]
    ks = sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"])[:5]
    line += "  [" + ", ".join("%.3f" % x["avg_launch_ms"] for _, x in ks) + "]"
print(line)
PY
    done
  done
done
