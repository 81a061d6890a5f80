#!/usr/bin/env python3
"""Which lines of the package launch the ATen / runtime glue kernels of a train step (copies, fills, elementwise, reductions)?
torch.profiler with Python stacks over two big train steps; prints the glue launches grouped by the innermost sound_bubble_amd
frame.  usage: find_glue.py [big|small]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                      # noqa: E402
import sound_bubble_amd as sb                                     # noqa: E402
from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step     # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "big"
cls, params, B, negw, clip, lr = bench.WORKLOADS[wl]
dev = torch.device("cuda")
torch.manual_seed(0)
model = getattr(sb, cls)(**params).to(dev).train()
bucket = FlatBucket(model)
optim = FusedAdam(bucket, lr=lr)
inputs, target = bench.synth_batch(torch, B, 1234, dev, cls != "NetOptim")
for _ in range(3):
    train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip)
    torch.cuda.synchronize()
names = ("copy", "fill", "Memcpy", "Memset", "cat", "sum", "mean", "zero", "add", "mul", "stack", "contiguous", "clone", "to")
print(f"ATen / runtime ops per step ({wl}; 2 steps profiled), by name:")
byname = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") or "Memcpy" in ev.name or "Memset" in ev.name or "hipMemcpy" in ev.name or "hipMemset" in ev.name:
        byname[ev.name] += 1
for n, c in byname.most_common(40):
    print(f"  {c / 2:7.1f}  {n}")
print("by (op, stack):")
rows = []
for ka in prof.key_averages(group_by_stack_n=12):
    if not any(t in ka.key for t in names):
        continue
    st = [f for f in (ka.stack or []) if "sound_bubble_amd" in f or "bench.py" in f]
    rows.append((ka.count, ka.key, st[0].split("sound_bubble_amd/")[-1][:100] if st else "(no package frame)"))
for c, k, f in sorted(rows, reverse=True)[:60]:
    print(f"  {c / 2:7.1f}  {k:30s} {f}")
