#!/bin/bash
# same-box A/B of library variants on the headline train step (lib/exp/lib_<name>.so via SB_LIB_VARIANT; "main" = the tree's library)
# usage: gpu_ab_train.sh "<v1> <v2> ..." <rounds> [workload]
R="${GRAFT_REPO_ROOT:-.}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
VARS="$1"; ROUNDS=${2:-2}; WL=${3:-big}
for r in $(seq 1 $ROUNDS); do
  for v in $VARS; do
    if [ $v = main ]; then unset SB_LIB_VARIANT; else export SB_LIB_VARIANT=$v; fi
    timeout 600 python bench.py --workload $WL --no-cpu-baseline --no-exact --no-parity --steps 20 --warmup 5 2>/dev/null | grep '^{' | tail -1 > gpurun_out/ab_${v}_${WL}_$r.jsonl
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${v}_${WL}_$r.jsonl").read())
ks = sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1].get("share_of_step", 0))[:5]
print("$v $WL round $r: %.1f utt/s  %.3f ms (median %.3f)  [" % (d["value"], d["ms_per_step"], d.get("ms_per_step_median", 0)) + ", ".join("%s %.3f" % (k[:28], x["avg_launch_ms"]) for k, x in ks) + "]")
PY
  done
done
