"""Experiment (round 4): what the big training FORWARD gains when the inter-frame pass stops storing its gate records
(c_prev + u / hs pairs only).  Forward only, under autograd recording; the backward is not run (it would need the records).
usage: python scripts/exp_fwd_nogates.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sound_bubble_amd as sb
from sound_bubble_amd import ops
import bench

cls, params, B, negw, clip, lr = bench.WORKLOADS["big"]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = getattr(sb, cls)(**params).to(dev).train()
inputs, target = bench.synth_batch(torch, B, 1234, dev, True)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = model(inputs)["output"]
        del out
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for mode in ("records", "no-inter-gates", "records", "no-inter-gates"):
    ops.EXP_NO_INTER_GATES = mode == "no-inter-gates"
    run(3)
    ops.PROFILE = {}
    run(3)
    prof, ops.PROFILE = ops.PROFILE, None
    ms = run(10)
    print(f"{mode:16s} forward (training mode) {ms:7.3f} ms")
    for k, evs in prof.items():
        print(f"      {sum(e[0].elapsed_time(e[1]) for e in evs) / len(evs):7.3f} ms x {len(evs) // 3}  {k}")
