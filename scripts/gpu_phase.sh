#!/bin/bash
# developer run: per-phase cycle counters of the recurrent kernels (builds a -DSB_PHASE_TIMING variant of the library on the box)
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out sound_bubble_amd/lib/exp; export TMPDIR=/tmp
cp -r sound_bubble_amd/lib /tmp/lib_keep
SB_EXTRA_HIPCC_FLAGS=-DSB_PHASE_TIMING timeout 1500 python -m sound_bubble_amd.build --force > gpurun_out/phase_build.log 2>&1
timeout 600 python scripts/phase_timing_train.py > gpurun_out/phase_timing.txt 2>&1
rm -rf sound_bubble_amd/lib; cp -r /tmp/lib_keep sound_bubble_amd/lib
cat gpurun_out/phase_timing.txt | tail -80
