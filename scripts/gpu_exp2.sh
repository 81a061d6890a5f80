#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" nopro; do
  echo "== variant '$v'"
  SB_LIB_VARIANT=$v timeout 600 python - <<'PY' 2>&1 | grep "sync-between"
import os, sys, runpy
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from sound_bubble_amd import _lib as _L
if os.environ.get("SB_LIB_VARIANT"):
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "exp", f"lib_{os.environ['SB_LIB_VARIANT']}.so")
runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "scripts", "exp_cross_consume.py"), run_name="__main__")
PY
done
