"""Experiment (round 4): how much wall time the inter-frame fused backward (145 tiles, one workgroup per CU, 111 CUs idle) and
the intra-frame bidirectional fused backward (persistent, all CUs) take when they run SIDE BY SIDE on two streams, against one
after the other -- the potential of overlapping the backward across the two passes of a block.  Timing only: both kernels get
independent inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sound_bubble_amd import ops

H, T, F, B, C = 64, 625, 145, 16, 32
dev = "cuda"
torch.manual_seed(0)
ops.BPTT = "wide"


def mk(geom, ndir, lin_dim):
    x = torch.randn(geom.P, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    dirs = [tuple(t.to(dev) for t in (torch.randn(256, C) * 0.1, torch.randn(256, H) * 0.1, torch.zeros(256), torch.zeros(256)))
            for _ in range(ndir)]
    lin_w, lin_b = torch.randn(C, lin_dim, device=dev) * 0.1, torch.zeros(C, device=dev)
    y = torch.empty(geom.P, ndir, C, device=dev) if ndir == 2 else torch.empty(geom.P, C, device=dev)
    hs, _, gates, u = ops.lstm_fwd(x, g, b, dirs, geom, save=True, lin=(lin_w, lin_b, y), want_hs=(ndir == 1))
    dy = torch.randn(geom.P, C, device=dev) * 0.01
    return dict(x=x, g=g, dirs=dirs, lin_w=lin_w, hs=hs, gates=gates, u=u, dy=dy)


gi, ga = ops.Geom.inter(B, T, F), ops.Geom.intra(B * T, F)
I, A = mk(gi, 1, H), mk(ga, 2, 2 * H)


def tg(n):
    return [torch.zeros(256, C, device=dev), torch.zeros(256, H, device=dev), torch.zeros(256, device=dev), torch.zeros(256, device=dev)]


def k1():
    ops.absmax_hints_clear()
    return ops.lstm_bwd_fused(I["dirs"][0][1], I["gates"], gi, I["dy"], I["lin_w"], I["u"], I["hs"], I["dirs"][0][0], tg(1),
                              lin_targets=(torch.zeros(C, H, device=dev), torch.zeros(C, device=dev)))


def k3():
    ops.absmax_hints_clear()
    return ops.lstm_bwd_fused_bi([d[1] for d in A["dirs"]], A["gates"], ga, A["u"], A["hs"], [d[0] for d in A["dirs"]], [tg(1), tg(1)],
                                 dy=A["dy"], w_lin=A["lin_w"], lin_targets=(torch.zeros(C, 2 * H, device=dev), torch.zeros(C, device=dev)))


def wall(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


side = torch.cuda.Stream()


def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    k1()
    with torch.cuda.stream(side):
        k3()
    main.wait_stream(side)


def serial():
    k1()
    k3()


print(f"K1 inter fused (145 tiles)        {wall(k1):.3f} ms")
print(f"K3 intra fused bidirectional      {wall(k3):.3f} ms")
print(f"one after the other               {wall(serial):.3f} ms")
print(f"side by side on two streams       {wall(both):.3f} ms")
print(f"side by side on two streams       {wall(both):.3f} ms")
print(f"one after the other               {wall(serial):.3f} ms")
