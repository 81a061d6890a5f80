"""Build libsoundbubble_hip.so (gfx950) in-tree with hipcc.  No torch dependency."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsoundbubble_hip.so")
SOURCES = ["sb_lstm.hip", "sb_lstm_gen.hip", "sb_lstm_vec.hip", "sb_lstm_bf_fwd.hip", "sb_lstm_bf_bwd.hip", "sb_lstm_stream.hip", "sb_attention.hip",
           "sb_linear.hip", "sb_elementwise.hip", "sb_mrstft.hip"]
HEADERS = [os.path.join(CSRC, "sb_common.h"), os.path.join(CSRC, "sb_lstm_bf_common.h"),
           os.path.join(HERE, "..", "include", "sound_bubble_hip.h")]


def csrc_digest():
    """sha256 (first 16 hex digits) over the kernel sources and the C-ABI header, names and bytes in sorted order: what a
    committed counter profile is stamped with (scripts/pmc_summary.py) and what bench.py compares it against (`stale_profile`)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) + [os.path.normpath(HEADERS[-1])]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# The recurrent kernels keep few accumulators: MFMA results straight into VGPRs saves the v_accvgpr_read copies (and
# their hazard nops) in front of every cell update (+2-3 % on the inference forward, training neutral; same-box A/B).
# The streaming / GEMM kernels hold ~100 accumulator registers and are better off with AGPRs (hipcc's default).
# -ffp-contract=off: every fused multiply-add in the recurrent kernels is written out (__builtin_fmaf), so all template
# variants of a kernel (e.g. the time-segmented and the plain schedule) round identically -- bit-exact outputs.
_BF_FLAGS = ["-mllvm", "-amdgpu-mfma-vgpr-form", "-ffp-contract=off"]
PER_FILE_FLAGS = {"sb_lstm_bf_fwd.hip": _BF_FLAGS, "sb_lstm_bf_bwd.hip": _BF_FLAGS}
# translation units built a second time with another macro set: (source, object stem, extra flags).  The opt-in two-product
# forward (sb_lstm_fwd_args.products == 2) is the forward file again with -DSB_FWD_2P: its inference kernels only, launcher
# sb_launch_lstm_fwd_bf_2p.
EXTRA_UNITS = [("sb_lstm_bf_fwd.hip", "sb_lstm_bf_fwd_2p", _BF_FLAGS + ["-DSB_FWD_2P"])]


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", s, "-o", o]
            cmd[2:2] = os.environ.get("SB_EXTRA_HIPCC_FLAGS", "").split()     # e.g. -DSB_PHASE_TIMING (dev tool)
            cmd[2:2] = PER_FILE_FLAGS.get(src, [])
            jobs.append(cmd)
    for src, stem, flags in EXTRA_UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, stem + ".o")
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-c", s, "-o", o]
            cmd[2:2] = os.environ.get("SB_EXTRA_HIPCC_FLAGS", "").split()
            cmd[2:2] = flags
            jobs.append(cmd)
    if jobs:                                   # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(run, jobs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
