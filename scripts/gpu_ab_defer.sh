#!/bin/bash
# developer run, one box: parity tests, then the train steps of both workloads (wide + compact sibling) with the library as
# built and with a -DSB_NO_DEFER rebuild (record stores issued at the end of their own step)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 600 2>&1 | tail -4) > gpurun_out/t_quick.log 2>&1
run() { for wl in big small; do timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/ab_${wl}_$1.jsonl; done; }
run defer
cp -r sound_bubble_amd/lib /tmp/lib_keep
SB_EXTRA_HIPCC_FLAGS=-DSB_NO_DEFER timeout 900 python -m sound_bubble_amd.build --force > gpurun_out/ab_build.log 2>&1
run nodefer
rm -rf sound_bubble_amd/lib; cp -r /tmp/lib_keep sound_bubble_amd/lib
run defer2
python - <<'PY'
import json
for wl in ("big", "small"):
    for v in ("defer", "nodefer", "defer2"):
        d = json.loads(open(f"gpurun_out/ab_{wl}_{v}.jsonl").read())
        print(wl, v, "wide", round(d["value"], 1), "compact", round(d["compact_bptt"]["value"], 1))
        for k, x in sorted(d["roofline"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"]):
            if "fwd" in k: print(f"      W {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
        for k, x in sorted(d["compact_bptt"]["kernels"].items(), key=lambda kv: -kv[1]["share_of_step"]):
            if "fwd" in k: print(f"      C {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
tail -3 gpurun_out/t_quick.log
