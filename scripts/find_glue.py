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
# Python-level census: which package lines call the tensor methods / factories that launch glue kernels (the profiler's stacks
# are empty on this ROCm build)
import traceback
calls = collections.Counter()


def _frame():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "sound_bubble_amd" in f.filename or f.filename.endswith("bench.py"):
            return f"{os.path.basename(f.filename)}:{f.lineno} {f.line.strip()[:90]}"
    return "(outside the package)"


def _wrap_method(name):
    orig = getattr(torch.Tensor, name)

    def w(self, *a, **k):
        if self.is_cuda:
            calls[(name, _frame())] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, w)
    return orig


def _wrap_factory(name):
    orig = getattr(torch, name)

    def w(*a, **k):
        dev = k.get("device")
        if dev is not None and "cuda" in str(dev):
            calls[(name, _frame())] += 1
        return orig(*a, **k)
    setattr(torch, name, w)
    return orig


saved = {n: _wrap_method(n) for n in ("copy_", "fill_", "zero_", "clone", "contiguous", "sum", "mean", "to", "float", "mul", "add")}
savedf = {n: _wrap_factory(n) for n in ("zeros", "zeros_like", "ones", "ones_like", "full", "cat", "stack", "tensor")}
for _ in range(2):
    train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip)
torch.cuda.synchronize()
for n, o in saved.items():
    setattr(torch.Tensor, n, o)
for n, o in savedf.items():
    setattr(torch, n, o)
print(f"Python-level calls per step ({wl}; .contiguous() / .to() on an already fitting tensor launch nothing):")
for (name, frame), c in sorted(calls.items(), key=lambda kv: -kv[1]):
    print(f"  {c / 2:6.1f}  {name:12s} {frame}")
sys.exit(0)
