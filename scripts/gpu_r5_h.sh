#!/bin/bash
# round 5, job A: the overfit training run (checkpoint + curve), the whole -m gpu suite, one quick bench line
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/overfit5
( time timeout 900 python -m sound_bubble_amd.train_cli --config experiments/overfit_test_samples.json --run_dir gpurun_out/overfit5 --epochs 300 ) > gpurun_out/overfit5_train.log 2>&1
grep -c "Average Loss: nan" gpurun_out/overfit5_train.log; grep "val/si_sdr_i:" gpurun_out/overfit5_train.log | tail -3; tail -4 gpurun_out/overfit5_train.log
( timeout 300 python scripts/overfit_report.py gpurun_out/overfit5 gpurun_out/overfit5/report.json > gpurun_out/overfit5_report.log 2>&1 ); tail -3 gpurun_out/overfit5_report.log | cut -c1-300
( timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_trained.py::test_trained_checkpoint_matches_the_imported_reference_at_positive_si_sdr 2>&1 | grep -v "^$" | tail -40 ) > gpurun_out/r5h_tests.log 2>&1; tail -6 gpurun_out/r5h_tests.log
( timeout 300 python bench.py --workload big --steps 20 --no-cpu-baseline --no-exact 2>gpurun_out/r5h_bench.err | tail -1 > gpurun_out/r5h_bench.jsonl ); python -c "
import json; d=json.loads(open('gpurun_out/r5h_bench.jsonl').read()); print('big', d['value'], d['ms_per_step'], d.get('schedules',{}).get('per_rank'))"
