#!/usr/bin/env python3
"""Forward and train-step time of the reference constructors' OWN widths -- Net(L=4): n_fft 280 (F 141), 2 microphones, D 64,
H 128, six conv-LSTM blocks -- on the generic-shape kernels (csrc/sb_lstm_gen.hip, sb_wgrad's generic form).  Not a BASELINE
config (no shipped experiment JSON uses these widths): a secondary figure for DESIGN.md.
usage: bench_default_ctor.py [--batch B] [--steps K] [--family dis_embd3|optim] [--seconds S]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sound_bubble_amd as sb                                             # noqa: E402
from sound_bubble_amd import ops                                          # noqa: E402
from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--family", default="dis_embd3")
ap.add_argument("--seconds", type=float, default=5.0)
args = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
cls = sb.NetDisEmbd3 if args.family == "dis_embd3" else sb.NetOptim
m = cls(L=4).to(dev)
N = int(24000 * args.seconds)
g = torch.Generator().manual_seed(1234)
mix = (0.1 * torch.randn(args.batch, 2, N, generator=g)).clamp(-1, 1).to(dev)
tgt = (0.05 * torch.randn(args.batch, 1, N, generator=g)).to(dev)
inputs = {"mixture": mix}
if args.family == "dis_embd3":
    inputs["dis_embed"] = torch.eye(3)[torch.arange(args.batch) % 3].to(dev)
out = {"model": f"{cls.__name__}(L=4): D=64 H=128 B=6 conv_lstm n_fft=280 num_ch=2", "params": sum(p.numel() for p in m.parameters()),
       "batch": args.batch, "seconds": args.seconds}
m.eval()
with torch.no_grad():
    for _ in range(2):
        m(inputs)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(args.steps):
        m(inputs)
    torch.cuda.synchronize()
    out["forward_ms"] = 1e3 * (time.time() - t0) / args.steps
    out["forward_utt_s"] = args.batch / (out["forward_ms"] * 1e-3)
# streaming: one 8 ms chunk at a time through the captured hipGraph (B = 1): the reference's real-time use
from sound_bubble_amd.streaming import StreamingSeparator                 # noqa: E402
with torch.no_grad():
    de = torch.eye(3)[:1].to(dev) if args.family == "dis_embd3" else None
    sep = StreamingSeparator(m, batch_size=1, dis_embed=de)
    chunk = (0.1 * torch.randn(1, 2, m.stft_chunk_size + m.stft_pad_size, generator=g)).to(dev)
    for _ in range(20):
        sep.feed(chunk)
    torch.cuda.synchronize()
    lat = []
    for _ in range(200):
        t0 = time.perf_counter()
        sep.feed(chunk)
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - t0))
    lat.sort()
    out["stream_chunk_ms_p50"], out["stream_chunk_ms_p99"] = lat[100], lat[197]
    out["stream_chunk_samples"] = int(m.stft_chunk_size)
    out["stream_realtime_factor"] = (m.stft_chunk_size / 24000.0 * 1e3) / lat[100]
    del sep
m.train()
bucket = FlatBucket(m)
optim = FusedAdam(bucket, lr=1e-3)
for _ in range(2):
    loss = train_step(m, bucket, optim, inputs, tgt, 100.0, grad_clip=1.0)
torch.cuda.synchronize()
ops.PROFILE = {}
t0 = time.time()
for _ in range(args.steps):
    loss = train_step(m, bucket, optim, inputs, tgt, 100.0, grad_clip=1.0)
torch.cuda.synchronize()
out["train_ms"] = 1e3 * (time.time() - t0) / args.steps
out["train_utt_s"] = args.batch / (out["train_ms"] * 1e-3)
out["loss"] = float(loss)
out["peak_mem_gb"] = torch.cuda.max_memory_allocated() / 2 ** 30
prof = {}
for label, recs in ops.PROFILE.items():
    ms = sum(r[0].elapsed_time(r[1]) for r in recs)
    prof[label] = {"launches_per_step": len(recs) / args.steps, "ms_per_step": ms / args.steps}
out["recurrent_kernels"] = prof
print(json.dumps(out))
