"""Developer utility: which Python line launches small fill kernels inside the backward pass of a big-config train step"""
import sys, os, collections, traceback
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
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("zero", "fill", "full", "ones", "copy", "clone", "add", "mul", "cat")):
            st = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack() if "sound_bubble_amd" in f.filename or "bench.py" in f.filename]
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            cnt[(name, str(shp), " < ".join(st[-3:]))] += 1
        return func(*args, **(kwargs or {}))
with M():
    train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip)
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(v, k)
