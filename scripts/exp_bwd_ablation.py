#!/usr/bin/env python3
"""Experiment: marginal cost of the memory streams of the backward recurrence (variant libraries built with
-DSB_EXP_SKIP=256: no dgates store) against the current library.  GPU box only.
(Constant records instead of the loads are NOT a valid ablation: the compiler hoists the cell arithmetic with them.)"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_bubble_amd import _lib as L  # noqa: E402

mask = sys.argv[1]
if mask != "cur":
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "exp", f"lib_{mask}.so")
from sound_bubble_amd import ops  # noqa: E402

H, T, F = 64, 625, 145


def timed(fn, iters=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(C, geom, ndir, lin):
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.randn(geom.P, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dirs = [tuple(t.to(dev) for t in (torch.randn(256, C) * 0.1, torch.randn(256, H) * 0.1, torch.zeros(256), torch.zeros(256)))
            for _ in range(ndir)]
    lw, lb, y = torch.randn(C, H, device=dev) * 0.1, torch.zeros(C, device=dev), torch.empty(geom.P, C, device=dev)
    hs, _, gates, u = ops.lstm_fwd(x, g, b, dirs, geom, save=True, lin=(lw, lb, y) if lin else None)
    dy = torch.randn(geom.P, C, device=dev)
    dhs = torch.randn(geom.P, ndir * H, device=dev)
    gm = ops.absmax(dy)
    real = ops.absmax
    ops.absmax = lambda t: gm
    whh = [d[1] for d in dirs]
    if lin:
        t = timed(lambda: ops.lstm_bwd_rec(whh, gates, None, geom, dy=dy, w_lin=lw))
    else:
        t = timed(lambda: ops.lstm_bwd_rec(whh, gates, dhs, geom))
    ops.absmax = real
    return t


r = [case(16, ops.Geom.inter(32, T, F), 1, True), case(32, ops.Geom.inter(16, T, F), 1, True),
     case(16, ops.Geom.intra(32 * T, F // 5), 2, False), case(32, ops.Geom.intra(16 * T, F), 2, False)]
print(f"lib={mask:4s}  bwd_rec: small inter {r[0]:7.1f}  big inter {r[1]:7.1f}  small intra {r[2]:7.1f}  big intra {r[3]:7.1f} us", flush=True)
