"""Per-phase cycles of the single-direction forward step with the fused Linear, pipelined (default) vs one-barrier
   (SB_STEP_PLAIN=1).  GPU box; needs a -DSB_PHASE_TIMING build (see phase_timing.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_bubble_amd import ops
H = 64
ops.PHASE_TIMING_BUF = torch.zeros(4096, device="cuda")
def run(name, C, geom, plain):
    ops.STEP_PLAIN = plain
    torch.manual_seed(0)
    x = torch.randn(geom.P, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dirs = [tuple(t.cuda() for t in (torch.randn(256, C) * 0.1, torch.randn(256, H) * 0.1, torch.zeros(256), torch.zeros(256)))]
    lw, lb, y = torch.randn(C, H, device="cuda") * 0.1, torch.zeros(C, device="cuda"), torch.empty(geom.P, C, device="cuda")
    for _ in range(3):
        ops.lstm_fwd(x, g, b, dirs, geom, save=False, lin=(lw, lb, y), want_hs=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.lstm_fwd(x, g, b, dirs, geom, save=False, lin=(lw, lb, y), want_hs=False); e1.record(); torch.cuda.synchronize()
    d = ops.PHASE_TIMING_BUF[:128].view(16, 8)[:, :5].cpu()
    print(name, "plain" if plain else "pipe ", "us/step %.3f" % (e0.elapsed_time(e1) * 1e3 / geom.nsteps), "phases:",
          d[:4].numpy().round(0).tolist(), "sum", float(d[0].sum()), flush=True)
T, F = 625, 145
for plain in (True, False):
    run("big inter", 32, ops.Geom.inter(16, T, F), plain)
    run("small inter", 16, ops.Geom.inter(32, T, F), plain)
