#!/bin/bash
# round-4 evidence: rocprofv3 kernel-trace stats (wide = default mode: big, small; compact: big) and PMC passes (separate
# passes, never combined with sys/hip/hsa trace domains) of the default-mode train step
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
prof() { # name, bench args
  cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$1" -o k -- python "$R/bench.py" $2 --no-cpu-baseline --no-exact > "$R/gpurun_out/prof_$1.log" 2>&1
}
prof big_wide "--steps 3 --warmup 1 --workload big"
prof small_wide "--steps 5 --warmup 2 --workload small"
pmc() { # workload
  WL=$1; ARGS="--steps 2 --warmup 1 --workload $WL --no-cpu-baseline --no-exact"
  cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch_$WL" -o f -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_fetch_$WL.log" 2>&1
  cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write_$WL" -o w -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_write_$WL.log" 2>&1
  cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d "$R/gpurun_out/pmc_sq_$WL" -o s -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_sq_$WL.log" 2>&1
  cd "$R"; python scripts/pmc_summary.py $WL "$R/gpurun_out" "$R/gpurun_out/pmc_traffic_${WL}_wide.json" "$R/gpurun_out/pmc_sq_${WL}_wide.json" > "$R/gpurun_out/pmc_summary_$WL.log" 2>&1
}
pmc big
pmc small
# the overlapped inter-frame backward pair cannot be seen by a serialising profiler: its two kernels in plain order
# (SB_BWD_PAIR_SERIAL=1) -- the pair's HBM traffic is the sum of the two launches
ARGS="--steps 2 --warmup 1 --workload big --no-cpu-baseline --no-exact"
cd /tmp && SB_BWD_PAIR_SERIAL=1 timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch_bigpair" -o f -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_fetch_bigpair.log" 2>&1
cd /tmp && SB_BWD_PAIR_SERIAL=1 timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write_bigpair" -o w -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_write_bigpair.log" 2>&1
cd "$R"; python scripts/pmc_summary.py bigpair "$R/gpurun_out" "$R/gpurun_out/pmc_traffic_big_wide_pair.json" "$R/gpurun_out/pmc_sq_bigpair_unused.json" > "$R/gpurun_out/pmc_summary_bigpair.log" 2>&1
cd "$R"; find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*counter_collection.csv" -size +20M -delete; find gpurun_out -name "*.db" -delete
ls gpurun_out | head -50
# the vendor yardstick on this tree (MIOpen / rocBLAS through torch.nn on the same workloads)
cd "$R"; timeout 900 python bench.py --workload big --vendor-gpu-baseline --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/vendor_gpu_baseline_lines.jsonl
timeout 900 python bench.py --workload small --vendor-gpu-baseline --no-cpu-baseline --no-exact --steps 10 --warmup 3 2>/dev/null | grep '^{' >> gpurun_out/vendor_gpu_baseline_lines.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/vendor_gpu_baseline_lines.jsonl"):
    d = json.loads(l); v = d.get("vendor_gpu_baseline") or {}
    print(d["config"]["workload"][:12], "ours", round(d["value"], 1), "vendor", v.get("value"), "x", v.get("ours_over_vendor"))
PY
