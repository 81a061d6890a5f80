#!/usr/bin/env python3
"""Experiment behind DESIGN.md's "recompute the gates in the backward instead of storing them" entry.

  build (CPU container):  python scripts/exp_gate_recompute.py build
      -> sound_bubble_amd/lib/exp/lib_{fwdnogates,rc1,rc3}.so  (sb_lstm_bf.hip rebuilt with -DSB_EXP_SKIP=1: the forward
         recurrence without its gate-record stores;  -DSB_EXP_RECOMPUTE=1: the backward recurrence with the recompute's
         instruction mix added per step;  =3: the same and without the gate-record loads)
  run (GPU box):          python scripts/exp_gate_recompute.py
      -> per variant: the intra-frame forward (training) and the fused bidirectional backward at both BASELINE geometries.
The variants compute garbage gradients by construction; only their timing is meaningful.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"fwdnogates": "-DSB_EXP_SKIP=1", "rc1": "-DSB_EXP_RECOMPUTE=1", "rc3": "-DSB_EXP_RECOMPUTE=3"}
if os.environ.get("SB_EXP_VARIANTS"):       # e.g. SB_EXP_VARIANTS="noagpr=-DSB_AGPR_OPERANDS=0": other A/B builds of sb_lstm_bf_{fwd,bwd}.hip
    VARIANTS = dict(v.split("=", 1) for v in os.environ["SB_EXP_VARIANTS"].split(";"))


def build():
    from sound_bubble_amd import build as B
    exp = os.path.join(B.LIBDIR, "exp")
    os.makedirs(exp, exist_ok=True)
    BF = ("sb_lstm_bf_fwd.hip", "sb_lstm_bf_bwd.hip")          # the two translation units of the recurrent kernels
    procs = []
    for name, flag in VARIANTS.items():
        for src in BF:
            obj = os.path.join(exp, src.replace(".hip", f"_{name}.o"))
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *B.PER_FILE_FLAGS[src], flag, "-O3", "-std=c++17",
                   "-fPIC", "-Wno-unused-value", "-c", os.path.join(B.CSRC, src), "-o", obj]
            procs.append((name, src, obj, subprocess.Popen(cmd)))
    B.build()
    for name, src, obj, p in procs:
        assert p.wait() == 0, (name, src)
    for name in VARIANTS:
        objs = [os.path.join(exp, s.replace(".hip", f"_{name}.o")) if s in BF else os.path.join(B.LIBDIR, s.replace(".hip", ".o"))
                for s in B.SOURCES]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                               os.path.join(exp, f"lib_{name}.so")] + objs)
        print("built", name)


def measure(variant):
    import torch
    from sound_bubble_amd import _lib as L
    if variant != "cur":
        L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "exp", f"lib_{variant}.so")
    from sound_bubble_amd import ops
    H, T, F = 64, 625, 145

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

    out = []
    for name, C, geom, fuse_lin in (("small intra (B=32, conv-LSTM, 29 steps)", 16, ops.Geom.intra(32 * T, F // 5), False),
                                    ("big intra (B=16, 145 steps)", 32, ops.Geom.intra(16 * T, F), True)):
        dev = "cuda"
        torch.manual_seed(0)
        x = torch.randn(geom.P, C, device=dev)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        dirs = [tuple(t.to(dev) for t in (torch.randn(256, C) * 0.1, torch.randn(256, H) * 0.1, torch.zeros(256),
                                          torch.zeros(256))) for _ in range(2)]
        t_f = timed(lambda: ops.lstm_fwd(x, g, b, dirs, geom, save=True))
        hs, _, gates, u = ops.lstm_fwd(x, g, b, dirs, geom, save=True)
        lin_w = torch.randn(C, 2 * H, device=dev) * 0.1
        dy = torch.randn(geom.P, C, device=dev) * 0.01
        dhs = torch.randn(geom.P, 2 * H, device=dev) * 0.01
        gm = ops.absmax(dy if fuse_lin else dhs)
        tg = [[torch.zeros(256, C, device=dev), torch.zeros(256, H, device=dev), torch.zeros(256, device=dev),
               torch.zeros(256, device=dev)] for _ in range(2)]
        ltg = [torch.zeros(C, 2 * H, device=dev), torch.zeros(C, device=dev)] if fuse_lin else None
        kw = dict(dy=dy, w_lin=lin_w, lin_targets=ltg) if fuse_lin else dict(dhs=dhs)
        t_b = timed(lambda: ops.lstm_bwd_fused_bi([dirs[0][1], dirs[1][1]], gates, geom, u, hs, [dirs[0][0], dirs[1][0]], tg,
                                                  gmax=gm, **kw))
        out.append(f"{name}: fwd(train) {t_f:7.1f} us  fused-bwd {t_b:7.1f} us")
    print(f"lib={variant:10s} " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 1:
        measure(sys.argv[1])
    else:
        for v in ["cur"] + list(VARIANTS) + ["cur"]:
            subprocess.call([sys.executable, os.path.abspath(__file__), v])
