#!/bin/bash
# round 5, first trip: the -m gpu suite on the round-start tree (+ the bucket-mutation test), the overfit run on the nine
# bundled scenes (train_cli, reference JSON surface) with its report, and the day's baseline bench line
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out/overfit; export TMPDIR=/tmp
(time timeout 900 python -m pytest tests -m gpu -x -q) > gpurun_out/r05_start_gpu_tests.log 2>&1; tail -3 gpurun_out/r05_start_gpu_tests.log
(time timeout 1200 python -m sound_bubble_amd.train_cli --config experiments/overfit_test_samples.json --run_dir gpurun_out/overfit) > gpurun_out/overfit/train.log 2>&1
tail -5 gpurun_out/overfit/train.log
timeout 300 python scripts/overfit_report.py gpurun_out/overfit gpurun_out/overfit/report.json 2>&1 | tail -12
timeout 600 python bench.py --workload big --no-cpu-baseline --no-exact --steps 20 --warmup 5 2>gpurun_out/bench_base.err | grep '^{' > gpurun_out/r05_start_bench_big.jsonl
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_start_bench_big.jsonl").read().splitlines()[-1]); print("big train", d["value"], d["ms_per_step"])
PY
