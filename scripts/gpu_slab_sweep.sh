#!/bin/bash
# same-box sweep of the overlapped schedules' slab lengths in the wide mode (big config) + the operator-boundary overhead
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --workload big --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/sweep_$label.jsonl
  python - "$label" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/sweep_{sys.argv[1]}.jsonl").read())
ks = d["roofline"]["kernels"]
pick = lambda s: next((f"{v['avg_launch_ms']:.3f}" for k, v in ks.items() if s in k), "-")
print(f"{sys.argv[1]:>14}: {d['value']:7.1f} utt/s  {d['ms_per_step']:.2f} ms   bwd pair {pick('inter overlapped')}  producer {pick('[producer]')}  consumer {pick('[consumer')}  intra bwd {pick('intra-frame fused BPTT')}")
PY
}
{
run base0 SB_X=0
run bslab16 SB_BWD_OVERLAP_SLAB=16
run bslab24 SB_BWD_OVERLAP_SLAB=24
run bslab48 SB_BWD_OVERLAP_SLAB=48
run bslab64 SB_BWD_OVERLAP_SLAB=64
run fslab16 SB_FWD_OVERLAP_SLAB=16
run fslab48 SB_FWD_OVERLAP_SLAB=48
run nobwdovl SB_NO_BWD_OVERLAP=1
run base1 SB_X=0
timeout 300 python scripts/exp_op_overhead.py big 10
timeout 300 python scripts/exp_op_overhead.py small 10
} > gpurun_out/sweep.log 2>&1
cat gpurun_out/sweep.log
