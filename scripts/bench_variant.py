"""Developer utility: bench.py against an experiment build of the library (SB_LIB_VARIANT=<name> -> lib/exp/lib_<name>.so)."""
import os, sys, runpy
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from sound_bubble_amd import _lib as _L
if os.environ.get("SB_LIB_VARIANT"):
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "exp", f"lib_{os.environ['SB_LIB_VARIANT']}.so")
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "bench.py"), run_name="__main__")
