#!/bin/bash
# PMC passes for the bench train step of one workload (separate passes, as MI355X_MICROARCH.md prescribes; never
# combined with sys/hip/hsa trace domains): HBM traffic (FETCH_SIZE, WRITE_SIZE) and SQ activity (MFMA-busy, wait /
# issue split).  usage: gpu_pmc.sh <small|big>
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp
WL=${1:-small}
ARGS="--steps 2 --warmup 1 --workload $WL --no-cpu-baseline --no-exact"
cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch_$WL" -o f -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_fetch_$WL.log" 2>&1
cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write_$WL" -o w -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_write_$WL.log" 2>&1
cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d "$R/gpurun_out/pmc_sq_$WL" -o s -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_sq_$WL.log" 2>&1
cd "$R"
python scripts/pmc_summary.py $WL "$R/gpurun_out" "$R/gpurun_out/pmc_traffic_$WL.json" "$R/gpurun_out/pmc_sq_$WL.json" > "$R/gpurun_out/pmc_summary_$WL.log" 2>&1
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*counter_collection.csv" -size +20M -delete
find gpurun_out -name "*.db" -delete
