#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_losses.py tests/test_gpu_film_bank.py "tests/test_gpu_parity.py::test_overlapped_forward_hands_items_back_when_the_producer_stands_still" -q --tb=short -s 2>&1 | grep -v "^$" | cut -c1-300 | tail -30 ) > gpurun_out/r5j_tests.log 2>&1; tail -12 gpurun_out/r5j_tests.log
( timeout 300 python bench.py --workload big --steps 20 --no-cpu-baseline --no-exact 2>gpurun_out/r5j_bench.err | tail -1 > gpurun_out/r5j_bench.jsonl ); python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j_bench.jsonl').read()); r=d['roofline']
print('big', round(d['value'],1), d['ms_per_step'], 'traffic', r.get('traffic'), r.get('traffic_source','')[:60], r.get('traffic_kernel'), 'frac', r['frac'], 'hbm_counter', r.get('frac_hbm_counter'), 'pass', r.get('pass_level',{}).get('frac'), r.get('pass_level',{}).get('frac_hbm_counter'))
for k,v in r.get('kernels',{}).items(): print('  ', k[:60], v.get('frac_hbm_counter'), v.get('mfma_busy'), v.get('traffic_kernel'))
PY
bash scripts/gpu_pmc.sh big-attn _wide > gpurun_out/r05_pmc_big_attn.log 2>&1; tail -2 gpurun_out/r05_pmc_big_attn.log
for i in 2 3; do ( timeout 600 python scripts/stress_train_loop.py --epochs 1700 > gpurun_out/stress_back$i.log 2>&1 ); grep "TRIP\|SLOW\|give-ups" gpurun_out/stress_back$i.log | head -6 | cut -c1-300; tail -1 gpurun_out/stress_back$i.log | cut -c1-200; done
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*counter_collection.csv" -size +20M -delete; find gpurun_out -name "*.db" -delete
