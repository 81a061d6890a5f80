#!/usr/bin/env python3
"""Experiment: per-item cost of the time-segmented schedule of the inter-frame recurrences (GPU box only).
B=28 (254 tiles) with the test hook "254,k": every workgroup walks its own tile's k segments -> pure prologue /
hand-off cost per item.  B=32 (290 tiles) with "256,k": the real cross-workgroup schedule."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_bubble_amd import ops  # noqa: E402

H = 64
T, F = 625, 145


def timed(fn, iters=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def run(B, C, hook):
    ops.SCHED_OVERRIDE = tuple(int(v) for v in hook.split(",")) if hook else None     # (workers, segments)
    dev = "cuda"
    geom = ops.Geom.inter(B, T, F)
    torch.manual_seed(0)
    x = torch.randn(geom.P, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    d = tuple(t.to(dev) for t in (torch.randn(256, C) * 0.1, torch.randn(256, H) * 0.1, torch.zeros(256), torch.zeros(256)))
    lw, lb, y = torch.randn(C, H, device=dev) * 0.1, torch.zeros(C, device=dev), torch.empty(geom.P, C, device=dev)
    h0, c0 = torch.zeros(geom.nseq, H, device=dev), torch.zeros(geom.nseq, H, device=dev)
    out = {}

    def fwd():
        out["r"] = ops.lstm_fwd(x, g, b, [d], geom, h0=h0, c0=c0, save=True, want_state=True, lin=(lw, lb, y), want_hs=True)

    t_f = timed(fwd)
    gates = out["r"][2]
    dy = torch.randn(geom.P, C, device=dev)
    gm = ops.absmax(dy)
    real_absmax = ops.absmax
    ops.absmax = lambda t: gm          # keep the absmax launch out of the timing
    t_b = timed(lambda: ops.lstm_bwd_rec([d[1]], gates, None, geom, dy=dy, w_lin=lw))
    ops.absmax = real_absmax
    print(f"B={B:3d} tiles={geom.nseq // 16:4d} hook={hook or '-':8s}  fwd(lin,save) {t_f:8.1f} us   bwd_rec(fused) {t_b:8.1f} us", flush=True)


if __name__ == "__main__":
    C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    for hook in (None, "254,2", "254,4", "254,7", "254,14"):
        run(28, C, hook)
    ops.TIME_SEGMENTS = False
    run(32, C, None)
    ops.TIME_SEGMENTS = True
    for hook in (None, "256,3", "256,5", "256,7", "256,8", "256,14", "256,15"):
        run(32, C, hook)
