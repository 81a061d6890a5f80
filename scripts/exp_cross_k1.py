"""Experiment: the cross-pass PRODUCER (inter-frame fused role-split backward with write-through du stores and slab signals) alone,
against the plain fused launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sound_bubble_amd import _lib as _L
if os.environ.get("SB_LIB_VARIANT"):
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "exp", f"lib_{os.environ['SB_LIB_VARIANT']}.so")
from sound_bubble_amd import ops
H, T, F, B, C = 64, 625, 145, 16, 32
dev = "cuda"
torch.manual_seed(0)
ops.BPTT = "wide"
geom = ops.Geom.inter(B, T, F)
x = torch.randn(geom.P, C, device=dev)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
dirs = [tuple(t.to(dev) for t in (torch.randn(256, C) * 0.1, torch.randn(256, H) * 0.1, torch.zeros(256), torch.zeros(256)))]
lin_w, lin_b = torch.randn(C, H, device=dev) * 0.1, torch.zeros(C, device=dev)
y = torch.empty(geom.P, C, device=dev)
hs, _, gates, u = ops.lstm_fwd(x, g, b, dirs, geom, save=True, lin=(lin_w, lin_b, y))
dy = torch.randn(geom.P, C, device=dev) * 0.01
assert ops.overlap_available()


def tg():
    return [torch.zeros(256, C, device=dev), torch.zeros(256, H, device=dev), torch.zeros(256, device=dev), torch.zeros(256, device=dev)]


def run(prod):
    slab = 32
    flags = torch.empty((B * F + 15) // 16 + 4 + 16 + 3 * ((B * T + 15) // 16) + 24, device=dev, dtype=torch.int32)
    ops.absmax_hints_clear()
    ops.PROFILE = {}
    for _ in range(6):
        ops.lstm_bwd_fused(dirs[0][1], gates, geom, dy, lin_w, u, hs, dirs[0][0], tg(),
                           lin_targets=(torch.zeros(C, H, device=dev), torch.zeros(C, device=dev)), produce=(flags, slab) if prod else None)
    torch.cuda.synchronize()
    (k, evs), = ops.PROFILE.items()
    ops.PROFILE = None
    return sum(e[0].elapsed_time(e[1]) for e in evs[2:]) / (len(evs) - 2)


for prod in (False, True, False, True):
    print(f"variant {os.environ.get('SB_LIB_VARIANT', '-'):8s} producer={prod!s:5s} {run(prod):.3f} ms")
