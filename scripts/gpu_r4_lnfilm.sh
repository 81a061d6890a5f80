#!/bin/bash
# round 4: fused LayerNorm + FiLM backward: new tests, the model-level parity tests, A/B of the big train step by switch
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_multireso.py -m gpu -q --timeout 900 -k "ln_film or golden or full_size or hs_free or cross_pass or deterministic or double_accumulated or trajectory or adam" 2>&1 | grep -v "^$" | grep -v "^E               \*" | cut -c1-300 | tail -30) > gpurun_out/r4k_tests.log 2>&1
tail -6 gpurun_out/r4k_tests.log
for v in on off on off; do
  if [ $v = off ]; then export SB_NO_LN_FILM_FUSION=1; else unset SB_NO_LN_FILM_FUSION; fi
  timeout 600 python bench.py --workload big --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>gpurun_out/r4k_$v.err | grep '^{' > gpurun_out/r4k_$v.jsonl
  python - <<PY
import json
d = json.loads(open("gpurun_out/r4k_$v.jsonl").read().strip().split("\n")[-1])
print("ln+film fusion $v: big train", round(d["value"], 1), "utt/s", round(d["ms_per_step"], 2), "ms")
for k, x in d["roofline"]["kernels"].items():
    if "ln_film" in k or "ln_bwd" in k or "film" in k: print(f"   {x['share_of_step']*100:5.1f}% {x['launches_per_step']:4.1f} x {x['avg_launch_ms']:.3f} ms {k}")
PY
done
