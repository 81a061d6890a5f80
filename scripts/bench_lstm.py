#!/usr/bin/env python3
"""Micro-benchmark of the recurrent LSTM kernels at the BASELINE geometries (GPU box only)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_bubble_amd import _lib as _L  # noqa: E402
if os.environ.get("SB_LIB_VARIANT"):      # experiment builds under lib/exp/
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "exp", f"lib_{os.environ['SB_LIB_VARIANT']}.so")
from sound_bubble_amd import ops  # noqa: E402

H = 64


def run(name, C, geom, ndir, save, iters=5):
    dev = "cuda"
    torch.manual_seed(0)
    x = torch.randn(geom.P, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dirs = [tuple(t.to(dev) for t in (torch.randn(256, C) * 0.1, torch.randn(256, H) * 0.1, torch.zeros(256), torch.zeros(256)))
            for _ in range(ndir)]
    for _ in range(2):
        hs, _, gates, u = ops.lstm_fwd(x, g, b, dirs, geom, save=save)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hs, _, gates, u = ops.lstm_fwd(x, g, b, dirs, geom, save=save)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * 256 * (C + H) * geom.P * ndir
    out = f"{name:28s} fwd save={int(save)} {ms*1e3:8.1f} us  {fl/ms/1e9:6.1f} TF/s"
    if save:
        dhs = torch.randn_like(hs)
        for _ in range(2):
            dg = ops.lstm_bwd_rec([d[1] for d in dirs], gates, dhs, geom)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            dg = ops.lstm_bwd_rec([d[1] for d in dirs], gates, dhs, geom)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / iters
        fl2 = 2.0 * 256 * H * geom.P * ndir
        out += f" | bwd_rec {ms2*1e3:8.1f} us  {fl2/ms2/1e9:6.1f} TF/s"
    print(out, flush=True)


if __name__ == "__main__":
    T, F = 625, 145
    for save in (False, True):
        run("big inter  B=16 C=32", 32, ops.Geom.inter(16, T, F), 1, save)
        run("big intra  B=16 C=32", 32, ops.Geom.intra(16 * T, F), 2, save)
        run("small inter B=32 C=16", 16, ops.Geom.inter(32, T, F), 1, save)
        run("small intra B=32 C=16", 16, ops.Geom.intra(32 * T, F // 5), 2, save)
