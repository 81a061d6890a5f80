#!/bin/bash
# PMC passes for the bench train step of one workload (separate passes, as MI355X_MICROARCH.md prescribes; never
# combined with sys/hip/hsa trace domains): HBM traffic (FETCH_SIZE, WRITE_SIZE) and SQ activity (MFMA-busy, wait /
# issue split).  SB_OVERLAP_FORCE=1: rocprofv3 --pmc serialises every dispatch, the side-stream probe fails and the
# library would fall back to its plain-order kernels -- forced, the counters see the SHIPPED producer / consumer
# instantiations (serialised: producer, then the consumer's two launches).  --no-parity: no B = 1 scene in the averages.
# usage: gpu_pmc.sh <small|big> [out-suffix]
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp
WL=${1:-small}; SUF=${2:-_wide}
ARGS="--steps 2 --warmup 1 --workload $WL --no-cpu-baseline --no-exact --no-parity"
export SB_OVERLAP_FORCE=1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch_$WL" -o f -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_fetch_$WL.log" 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write_$WL" -o w -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_write_$WL.log" 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d "$R/gpurun_out/pmc_sq_$WL" -o s -- python "$R/bench.py" $ARGS > "$R/gpurun_out/pmc_sq_$WL.log" 2>&1
cd "$R"
python scripts/pmc_summary.py $WL "$R/gpurun_out" "$R/gpurun_out/pmc_traffic_$WL$SUF.json" "$R/gpurun_out/pmc_sq_$WL$SUF.json" > "$R/gpurun_out/pmc_summary_$WL.log" 2>&1
grep -h "schedules" gpurun_out/pmc_fetch_$WL.log | head -1 | python -c "import sys,json; [print('schedules under the counter pass:', json.loads(l)['schedules']['per_rank']) for l in sys.stdin if l.startswith('{')]"
find gpurun_out -name "*kernel_trace.csv" -size +30M -delete; find gpurun_out -name "*counter_collection.csv" -size +20M -delete
find gpurun_out -name "*.db" -delete
