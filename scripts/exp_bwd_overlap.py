"""Experiment: overlapped inter-frame backward (recurrence || stream kernel on the idle CUs) at the BASELINE big geometry:
time of the pair of plain launches vs sb_lstm_bwd_inter_overlapped over the slab length."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sound_bubble_amd import _lib as _L
if os.environ.get("SB_LIB_VARIANT"):
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "exp", f"lib_{os.environ['SB_LIB_VARIANT']}.so")
from sound_bubble_amd import ops

PRE = os.environ.get("PRE", "")
if "streams" in PRE:                      # torch's stream pool (32 streams) before the library's side stream exists
    _s = [torch.cuda.Stream() for _ in range(4)]
    with torch.cuda.stream(_s[0]):
        torch.zeros(8, device="cuda")
    torch.cuda.synchronize()
if "graph" in PRE:
    _g = torch.cuda.CUDAGraph()
    _x = torch.zeros(1024, device="cuda")
    with torch.cuda.graph(_g):
        _x += 1
    if "noreplay" not in PRE:
        _g.replay()
    torch.cuda.synchronize()
    if "del" in PRE:
        del _g
        import gc
        gc.collect()
        torch.cuda.synchronize()
if "capture" in PRE:                      # raw stream capture without instantiating a graph exec
    _cs = torch.cuda.Stream()
    _x = torch.zeros(1024, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(_cs):
        _x += 1
    torch.cuda.synchronize()
if "mem" in PRE:                          # a fragmented caching-allocator pool
    _t = [torch.empty(int(n), device="cuda") for n in (3e8, 1e8, 2e8, 5e7, 1.5e8, 7e7)]
    del _t
B_, T_, F_, C_ = 16, 625, 145, 32
geom = ops.Geom.inter(B_, T_, F_)
torch.manual_seed(0)
x = torch.randn(geom.P, C_, device="cuda")
g, b = torch.rand(C_, device="cuda") + 0.5, torch.randn(C_, device="cuda") * 0.1
wi, wh = torch.randn(256, C_, device="cuda") * 0.1, torch.randn(256, 64, device="cuda") * 0.1
dirs = [(wi, wh, torch.randn(256, device="cuda") * 0.1, torch.randn(256, device="cuda") * 0.1)]
lin_w, lin_b = torch.randn(C_, 64, device="cuda") * 0.2, torch.randn(C_, device="cuda") * 0.1
y = torch.empty(geom.P, C_, device="cuda")
hs, _, gates, u = ops.lstm_fwd(x, g, b, dirs, geom, save=True, lin=(lin_w, lin_b, y))
dy = torch.randn(geom.P, C_, device="cuda") * 0.01
tg = [torch.zeros(256, C_, device="cuda"), torch.zeros(256, 64, device="cuda"), torch.zeros(256, device="cuda"),
      torch.zeros(256, device="cuda")]
lin = (torch.zeros(C_, 64, device="cuda"), torch.zeros(C_, device="cuda"))
ln = (torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda"))


def plain():
    dg = ops.lstm_bwd_rec([wh], gates, None, geom, dy=dy, w_lin=lin_w)
    return ops.lstm_bwd_stream(dg, u, hs, [wi], F_, T_ * F_, F_, targets=[tg], ln=(x, g, dy, ln[0], ln[1]), lin_targets=lin)[1]


def over():
    return ops.lstm_bwd_inter_overlapped(wh, gates, geom, dy, lin_w, u, hs, wi, tg, lin, (x, g, ln[0], ln[1]))


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"plain pair: {timeit(plain):.0f} us")
for slab in [int(v) for v in os.environ.get("SLABS", "16,32,64").split(",")]:
    ops.BWD_OVERLAP_SLAB = slab
    print(f"overlapped slab={slab}: {timeit(over):.0f} us", flush=True)
ops.check_sched_status()
