"""Developer utility: host-side aten copy / fill / zero ops of one big-config train step, grouped by shape (torch.profiler)."""
import sys, os, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import sound_bubble_amd as sb
from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step
cls, params, B, negw, clip, lr = bench.WORKLOADS["big"]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = getattr(sb, cls)(**params).to(dev).train()
bucket = FlatBucket(model); optim = FusedAdam(bucket, lr=lr)
inputs, target = bench.synth_batch(torch, B, 1, dev, True)
for _ in range(2):
    train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::zeros", "aten::clone", "aten::contiguous"):
        st = [s for s in (ev.stack or []) if "sound_bubble_amd" in s or "bench.py" in s]
        cnt[(ev.name, str(ev.input_shapes)[:60], st[0][-70:] if st else "?")] += 1
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(v, k)
