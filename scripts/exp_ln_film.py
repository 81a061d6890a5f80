"""Developer micro-benchmark: ln_film_bwd_kernel (+ its two reductions) alone at the headline geometry on cold caches, per library
variant (SB_LIB_VARIANT -> lib/exp/lib_<name>.so, scripts/build_variant.py).  Round 4 used it to price non-temporal loads (adopted:
285 -> 257 us here, 0.228 -> 0.209 ms inside the train step), a non-temporal store and other time chunks (nothing more):
profiles/r04_exp_ln_film.txt"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sound_bubble_amd import _lib as _L
if os.environ.get("SB_LIB_VARIANT"):
    _L.LIB_PATH = os.path.join(os.path.dirname(_L.LIB_PATH), "exp", f"lib_{os.environ['SB_LIB_VARIANT']}.so")
import torch
from sound_bubble_amd import ops
B, T, F, C = 16, 626, 145, 32
P = B * T * F
torch.manual_seed(0)
du = torch.randn(P, 2, C, device="cuda"); x = torch.randn(P, C, device="cuda"); res = torch.randn(P, C, device="cuda")
fx = torch.randn(P, C, device="cuda"); fw = torch.randn(B, F, C, device="cuda"); g = torch.randn(C, device="cuda")
dw = torch.zeros(B, F, C, device="cuda"); db = torch.zeros_like(dw); dg = torch.zeros(C, device="cuda"); dbt = torch.zeros(C, device="cuda")
junk = torch.empty(512 * 1024 * 1024 // 4, device="cuda")
ts = []
for it in range(12):
    junk.normal_()                                   # flush L2 / MALL
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ops.absmax_hints_clear()
    e0.record()
    out = ops.ln_film_bwd(du, x, g, res, fx, fw, dw, db, dg, dbt, (B, T, F, C))
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = sorted(ts[2:])
print(f"variant {os.environ.get('SB_LIB_VARIANT', 'main'):8s} ln_film_bwd + its two reductions: median {ts[len(ts)//2]*1e3:.1f} us  min {ts[0]*1e3:.1f} us  "
      f"checksum {float(out.double().abs().sum()):.6e}")
