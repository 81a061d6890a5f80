"""Train step through the module (`model(inputs)` + SnrlpLossFn) and through the dispatcher operators
(sound_bubble::separate + ::snrlp_loss, sound_bubble_amd/torch_ops.py): same kernels -- what does the operator boundary cost?
usage: python scripts/exp_op_overhead.py [big|small] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                  # noqa: E402
import bench                                  # noqa: E402
import sound_bubble_amd as sb                 # noqa: E402
from sound_bubble_amd import ops, torch_ops as T                          # noqa: E402
from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step, allreduce_grads     # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "big"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cls, params, B, negw, clip, lr = bench.WORKLOADS[wl]
torch.manual_seed(0)
model = getattr(sb, cls)(**params).cuda().train()
bucket = FlatBucket(model)
optim = FusedAdam(bucket, lr=lr)
inputs, target = bench.synth_batch(torch, B, 1234, "cuda", cls != "NetOptim")
w = T.separate_module(model)


def op_step():
    bucket.zero_grad()
    est = w(inputs)["output"]
    loss = torch.ops.sound_bubble.snrlp_loss(est, target, negw)[0]
    ops.absmax_hints_clear()
    loss.backward()
    ops.absmax_hints_clear()
    optim.step(grad_clip=clip, world_size=allreduce_grads(bucket))
    return loss.detach()


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        l = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, float(l)


for rnd in range(int(os.environ.get('ROUNDS', '2'))):
    a = timed(lambda: train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip))
    b = timed(op_step)
    print(f"[{wl}] round {rnd}: module {a[0]:.3f} ms/step (loss {a[1]:.4f})   operators {b[0]:.3f} ms/step (loss {b[1]:.4f})   "
          f"ratio {b[0] / a[0]:.4f}   reserved {torch.cuda.memory_reserved() >> 20} MiB  mallocs {torch.cuda.memory_stats()['num_device_alloc']}"
          f"  pending {len(T._PENDING)}  overlap_ok {ops.overlap_available()}")
ops.check_sched_status()
