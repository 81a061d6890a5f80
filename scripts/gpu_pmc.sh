#!/bin/bash
# HBM traffic counters (separate passes, as MI355X_MICROARCH.md prescribes) for the bench train step
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp
WL=${1:-small}
cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/pmc_fetch_$WL" -o f -- python "$R/bench.py" --steps 2 --warmup 1 --workload $WL --no-cpu-baseline > "$R/gpurun_out/pmc_fetch_$WL.log" 2>&1
cd /tmp && timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/pmc_write_$WL" -o w -- python "$R/bench.py" --steps 2 --warmup 1 --workload $WL --no-cpu-baseline > "$R/gpurun_out/pmc_write_$WL.log" 2>&1
cd "$R"; ls -la gpurun_out/pmc_fetch_$WL gpurun_out/pmc_write_$WL; find gpurun_out -name "*kernel_trace.csv" -size +30M -delete
