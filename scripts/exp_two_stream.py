"""Experiment: does running two half-batch train steps on two HIP streams beat one full-batch step?
(latency-bound inter-frame recurrences of one half overlapping the HBM-bound intra kernels of the other)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sound_bubble_amd as sb
from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step

wl = sys.argv[1] if len(sys.argv) > 1 else "small"
cls, params, B, negw, clip, lr = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)


def make(Bh, seed):
    torch.manual_seed(0)
    model = getattr(sb, cls)(**params).to(dev).train()
    bucket = FlatBucket(model)
    optim = FusedAdam(bucket, lr=lr)
    inputs, target = bench.synth_batch(torch, Bh, seed, dev, cls != "NetOptim")
    return model, bucket, optim, inputs, target


def run(reps, K=8):
    for r in reps[:1]:
        pass
    streams = [torch.cuda.Stream() for _ in reps]
    for _ in range(2):
        for r, s in zip(reps, streams):
            with torch.cuda.stream(s):
                train_step(r[0], r[1], r[2], r[3], r[4], negw, grad_clip=clip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for r, s in zip(reps, streams):
            with torch.cuda.stream(s):
                train_step(r[0], r[1], r[2], r[3], r[4], negw, grad_clip=clip)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = sum(r[3]["mixture"].shape[0] if isinstance(r[3], dict) else r[3].shape[0] for r in reps)
    return (t2 - t0) / K * 1e3, (t1 - t0) / K * 1e3, n


for split in (1, 2, 4):
    reps = [make(B // split, 100 + i) for i in range(split)]
    ms, cpu_ms, n = run(reps)
    print(f"{wl} split={split} batch/stream={B // split}: {ms:.2f} ms/step  (host launch {cpu_ms:.2f} ms)  {n / ms * 1e3:.1f} utt/s", flush=True)
    del reps
    torch.cuda.empty_cache()
