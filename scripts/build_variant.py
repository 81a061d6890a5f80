#!/usr/bin/env python3
"""Developer tool: build an A/B variant of the library under sound_bubble_amd/lib/exp/lib_<name>.so with extra hipcc flags
(e.g. `python scripts/build_variant.py nt -DSB_EXP_NT_RECORDS`); run it with `SB_LIB_VARIANT=<name> python scripts/bench_variant.py ...`."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from sound_bubble_amd import build as B

name, flags = sys.argv[1], sys.argv[2:]
exp = os.path.join(B.LIBDIR, "exp")
os.makedirs(exp, exist_ok=True)
B.build()
only = [s for s in B.SOURCES if s.startswith(os.environ.get("SB_VARIANT_TUS", "sb_lstm"))]      # the translation units the experiment flags touch


def cc(src):
    obj = os.path.join(exp, src.replace(".hip", f"_{name}.o"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *B.PER_FILE_FLAGS.get(src, []), *flags, "-O3", "-std=c++17",
                           "-fPIC", "-Wno-unused-value", "-c", os.path.join(B.CSRC, src), "-o", obj])
    return obj


with ThreadPoolExecutor(len(only)) as ex:
    objs = dict(zip(only, ex.map(cc, only)))
allobjs = [objs.get(s, os.path.join(B.LIBDIR, s.replace(".hip", ".o"))) for s in B.SOURCES]
for src, stem, xflags in B.EXTRA_UNITS:                  # second builds of a translation unit (their own macro sets) go in with the variant's flags too
    obj = os.path.join(exp, f"{stem}_{name}.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *xflags, *flags, "-O3", "-std=c++17", "-fPIC",
                           "-Wno-unused-value", "-c", os.path.join(B.CSRC, src), "-o", obj])
    allobjs.append(obj)
out = os.path.join(exp, f"lib_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + allobjs)
print("built", out)
