#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { # name, env..., args
  n=$1; shift
  ( env "$@" timeout 400 python scripts/dbg_overfit_cli.py 70 $EXTRA > gpurun_out/dbg_cli_$n.log 2>&1 ); echo "$n rc=$?"; grep -v "^NOT USING\|amdgpu.ids" gpurun_out/dbg_cli_$n.log | tail -2 | cut -c1-400
}
run default A=1
run default2 A=1
EXTRA=--no-reprobe run noreprobe A=1
run nocross SB_NO_BWD_CROSS_OVERLAP=1
run nodefer SB_NO_DEFERRED_REDUCE=1
