"""Per-phase s_memtime cycles of the recurrent kernels (GPU box; needs a -DSB_PHASE_TIMING build:
   SB_EXTRA_HIPCC_FLAGS=-DSB_PHASE_TIMING python -m sound_bubble_amd.build --force)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sound_bubble_amd import _lib as _L
if os.environ.get("SB_LIB_VARIANT"):
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "exp", f"lib_{os.environ['SB_LIB_VARIANT']}.so")
from sound_bubble_amd import ops
H = 64
ops.PHASE_TIMING_BUF = torch.zeros(4096, device="cuda")
def run(name, C, geom, ndir):
    torch.manual_seed(0)
    x = torch.randn(geom.P, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dirs = [tuple(t.cuda() for t in (torch.randn(256, C) * 0.1, torch.randn(256, H) * 0.1, torch.zeros(256), torch.zeros(256))) for _ in range(ndir)]
    for _ in range(3):
        ops.lstm_fwd(x, g, b, dirs, geom, save=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.lstm_fwd(x, g, b, dirs, geom, save=False); e1.record(); torch.cuda.synchronize()
    d = ops.PHASE_TIMING_BUF[:128].view(16, 8)[:, :5].cpu()
    print(name, "us/step %.3f" % (e0.elapsed_time(e1) * 1e3 / geom.nsteps), "cycles A(mfma issue) B(xpart+cell) C(ln+store_h) D(stores+load) E(barrier):",
          d[:4].numpy().round(0).tolist(), "sum", float(d[0].sum()), flush=True)
T, F = 625, 145
if True:
    run("big inter", 32, ops.Geom.inter(16, T, F), 1)
    run("big intra", 32, ops.Geom.intra(16 * T, F), 2)
    run("small inter", 16, ops.Geom.inter(32, T, F), 1)

def runb(name, C, geom, ndir):
    torch.manual_seed(0)
    x = torch.randn(geom.P, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    dirs = [tuple(t.cuda() for t in (torch.randn(256, C) * 0.1, torch.randn(256, H) * 0.1, torch.zeros(256), torch.zeros(256))) for _ in range(ndir)]
    hs, _, gates, u = ops.lstm_fwd(x, g, b, dirs, geom, save=True)
    dhs = torch.randn_like(hs)
    for _ in range(2):
        ops.lstm_bwd_rec([d[1] for d in dirs], gates, dhs, geom)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.lstm_bwd_rec([d[1] for d in dirs], gates, dhs, geom); e1.record(); torch.cuda.synchronize()
    d = dhs.view(-1)[:128].view(16, 8)[:, :5].cpu()
    print("BWD", name, "us/step %.3f" % (e0.elapsed_time(e1) * 1e3 / geom.nsteps), "A(wait record + prefetch) B(cell backward + split) C(dgates store + 48 mfma issue) D(partials -> LDS) E(barrier + reduce):",
          d[:2].numpy().round(0).tolist(), "sum", float(d[0].sum()), flush=True)
if True:
    runb("big inter", 32, ops.Geom.inter(16, T, F), 1)
    runb("big intra", 32, ops.Geom.intra(16 * T, F), 2)
    runb("small inter", 16, ops.Geom.inter(32, T, F), 1)

