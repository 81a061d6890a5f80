#!/usr/bin/env python3
"""Experiment: the whole train step (forward + loss + backward + clip + Adam) captured in ONE hipGraph and replayed,
against eager launches (GPU box only).  usage: exp_graph_step.py [small|big]"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sound_bubble_amd as sb  # noqa: E402
from sound_bubble_amd import ops  # noqa: E402
from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "big"
cls, params, B, negw, clip, lr = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = getattr(sb, cls)(**params).to(dev).train()
bucket = FlatBucket(model)
optim = FusedAdam(bucket, lr=lr)
inputs, target = bench.synth_batch(torch, B, 1234, dev, cls != "NetOptim")


def step():
    return train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip)


def timed(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_eager = timed(step)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
optim.step_count += 0
with torch.cuda.graph(g):
    loss = step()
t_graph = timed(g.replay)
print(f"{wl}: eager {t_eager:.3f} ms/step ({B / t_eager * 1e3:.1f} utt/s)   hipGraph replay {t_graph:.3f} ms/step "
      f"({B / t_graph * 1e3:.1f} utt/s)   loss {float(loss):.4f}", flush=True)
