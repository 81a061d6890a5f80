"""Experiment: the cross-pass consumer (bidirectional backward with the per-tile LayerNorm-backward prologue, drawn items) run
ALONE after its producer has finished, against the plain bidirectional backward -- what the prologue / draws / per-tile scale cost
per item, without any contention."""
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
assert ops.overlap_available()


def tg():
    return [torch.zeros(256, C, device=dev), torch.zeros(256, H, device=dev), torch.zeros(256, device=dev), torch.zeros(256, device=dev)]


def ev():
    return torch.cuda.Event(enable_timing=True)


def run(mode, sync_between):
    slab = ops.BWD_CROSS_SLAB
    flags = torch.empty((B * F + 15) // 16 + 4 + 16 + 3 * ((B * T + 15) // 16) + 24, device=dev, dtype=torch.int32)
    ops.absmax_hints_clear()
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    if mode == "cross":
        du, ok, keep = ops.lstm_bwd_fused(I["dirs"][0][1], I["gates"], gi, I["dy"], I["lin_w"], I["u"], I["hs"], I["dirs"][0][0], tg(),
                                          lin_targets=(torch.zeros(C, H, device=dev), torch.zeros(C, device=dev)), produce=(flags, slab))
        assert ok
    else:
        du = ops.lstm_bwd_fused(I["dirs"][0][1], I["gates"], gi, I["dy"], I["lin_w"], I["u"], I["hs"], I["dirs"][0][0], tg(),
                                lin_targets=(torch.zeros(C, H, device=dev), torch.zeros(C, device=dev)))
    e1.record()
    if sync_between:
        torch.cuda.synchronize()
        e1.record()
    dx = torch.empty(ga.P, C, device=dev)
    if mode == "cross":
        order, need = ops._cross_order(B, T, F, slab, torch.device(dev, 0))
        pend = ops.CrossBwd(flags, slab, (gi.nseq + 15) // 16, order, need, du, I["x"], I["dy"], I["g"],
                            torch.zeros(C, device=dev), torch.zeros(C, device=dev), dx, keep)
        r = ops.lstm_bwd_fused_bi([d[1] for d in A["dirs"]], A["gates"], ga, A["u"], None, [d[0] for d in A["dirs"]], [tg(), tg()],
                                  dy=dx, w_lin=A["lin_w"], lin_targets=(torch.zeros(C, 2 * H, device=dev), torch.zeros(C, device=dev)),
                                  consume=pend)
        assert r is not None
    else:
        ops.ln_bwd(du.view(-1, 1, C), I["x"], I["g"], res=I["dy"], out=dx)
        ops.lstm_bwd_fused_bi([d[1] for d in A["dirs"]], A["gates"], ga, A["u"], None, [d[0] for d in A["dirs"]], [tg(), tg()],
                              dy=dx, w_lin=A["lin_w"], lin_targets=(torch.zeros(C, 2 * H, device=dev), torch.zeros(C, device=dev)))
    e2.record()
    torch.cuda.synchronize()
    ops.check_sched_status()
    if mode == 'cross' and os.environ.get('SB_DBG'):
        nt = (B * T + 15) // 16
        d = flags[(B * F + 15) // 16 + 4 + 16 + 3 * nt:].tolist()
        print('   dbg: wg per (xcd,dir)', d[:16], ' lookahead', d[16], 'own', d[17], 'recomputed', d[18], 'waited', d[19], 'stolen items', d[20], ' us: rows %.1f / tile, done-wait %.1f / wait, slab-wait %.1f / own' % (d[21] * 64 / 100.0 / max(1, d[16] + d[17] + d[18]), d[22] * 64 / 100.0 / max(1, d[19]), d[23] * 64 / 100.0 / max(1, d[17] + d[18])))
    return e0.elapsed_time(e1), e1.elapsed_time(e2), e0.elapsed_time(e2)


for mode, sb in (("plain", False), ("cross", True), ("cross", False), ("plain", False), ("cross", True), ("cross", False)):
    run(mode, sb)
    r = [run(mode, sb) for _ in range(4)]
    a = [sum(x[i] for x in r) / len(r) for i in range(3)]
    print(f"{mode:6s} sync-between={sb!s:5s}  inter {a[0]:.3f} ms   intra(+ln) {a[1]:.3f} ms   total {a[2]:.3f} ms")
