"""Per-phase s_memtime ticks of the forward recurrent kernels inside real training steps of the model (wide and compact BPTT
   state), next to the inference forward.  GPU box; needs a -DSB_PHASE_TIMING build (scripts/gpu_phase.sh)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sound_bubble_amd import _lib as L, ops
# (SB_LIB_VARIANT=phase: an instrumented build made beforehand by `scripts/build_variant.py phase -DSB_PHASE_TIMING` -- _lib.py
#  resolves the variant's path itself)
lib = ctypes.CDLL(L.LIB_PATH)
KINDS = ["plain", "fused Linear", "summed input + fused Linear (inter-frame producer)", "ordered consumer (intra-frame)",
         "bidirectional partial-Linear (intra-frame, first block)"]
def table(tag):
    buf = (ctypes.c_float * (5 * 16 * 8))()
    assert lib.sb_debug_phase_fwd(buf) == 0
    t = torch.tensor(list(buf)).view(5, 16, 8)[:, :, :5]
    for k, name in enumerate(KINDS):
        if float(t[k].abs().sum()) == 0: continue
        rows = t[k][:4]        # the four waves of tile 0
        print(f"{tag:28s} {name}")
        for w in range(4):
            r = rows[w].tolist()
            print(f"    wave {w}: A(W_hh h + LN) {r[0]:6.0f}  B(W_ih u + cell) {r[1]:6.0f}  C(store_h) {r[2]:5.0f}  D(records, y) {r[3]:5.0f}  E(barrier) {r[4]:5.0f}   sum {sum(r):6.0f}")
import argparse
import sound_bubble_amd as sb
args = argparse.Namespace(batch=0, steps=2, warmup=1, bptt=None)
dev = torch.device("cuda:0")
for wl in ("big", "small"):
    for mode, fwd in (("wide", True), ("wide", False), ("compact", False)):
        bench.run_workload(torch, None, sb, ops, wl, args, dev, 1, 0, forward_only=fwd, mode=mode, steps=2, warmup=1, profile=False)
        torch.cuda.synchronize()
        table(f"{wl} / {'inference' if fwd else 'train, ' + mode}")
        if not fwd:
            b = (ctypes.c_float * 8)()
            if lib.sb_debug_phase_bwd_split(b) == 0 and sum(b[:4]) > 0:
                print(f"{wl} / train, {mode}: chunk role of the last role-split backward launch, ticks per period: work before hand-over "
                      f"{b[0]:.0f}, wait {b[1]:.0f}, work after {b[2]:.0f}, wait at second barrier {b[3]:.0f}  (sum {sum(b[:4]):.0f}); "
                      f"work after = stage issue {b[4]:.0f} + flush {b[5]:.0f} + dW chunk {b[6]:.0f} + wait for the staged rows {b[7]:.0f}")
            r = (ctypes.c_float * 24)()
            if hasattr(lib, "sb_debug_phase_bwd_rec") and lib.sb_debug_phase_bwd_rec(r) == 0 and sum(r) > 0:
                for w in range(4):
                    v = r[6 * w:6 * w + 6]
                    print(f"    recurrence role, wave {w}, ticks per step: wait for the records {v[0]:.0f}  next records' loads {v[1]:.0f}  "
                          f"gates + dgates {v[2]:.0f}  LDS dgates + MFMA {v[3]:.0f}  partial sums out {v[4]:.0f}  barrier + reduction {v[5]:.0f}   "
                          f"sum {sum(v):.0f}")
