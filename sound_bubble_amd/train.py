"""Training-step plumbing for the hot path: flat parameter/gradient bucket, fused clip + Adam (HIP),
one RCCL all-reduce per step.

Replaces, for the data-parallel path, nn.DataParallel's per-step broadcast/scatter/gather/reduce
(src/hl_modules/distance_based_hl_module.py:34-35) and PLModule.reset_grad/backprop (:430-441):
one process per GPU, identical replicas, ONE all-reduce over a single flat fp32 gradient buffer
(2.0 MB big / 0.9 MB small), then clip_grad_norm_ semantics and Adam on every rank.
"""
import torch

from . import ops


class FlatBucket:
    """All parameters of `module` re-pointed into one contiguous fp32 buffer; grads likewise."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        # 16-byte align every parameter so the HIP kernels can use 128-bit loads on weight rows
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, offs):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view(p.shape)
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.offsets = offs
        self.n_params = n

    def zero_grad(self):
        self.grad.zero_()


class FusedAdam:
    """torch.optim.Adam(lr, betas, eps) semantics (no weight decay / amsgrad) over a FlatBucket, with
    clip_grad_norm_(max_norm=grad_clip) folded in.  `lr` may be changed between steps (schedulers)."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay or amsgrad:
            raise NotImplementedError("weight_decay / amsgrad are not used by any shipped config")
        self.bucket, self.lr, self.betas, self.eps = bucket, lr, betas, eps
        self.m = torch.zeros_like(bucket.flat)
        self.v = torch.zeros_like(bucket.flat)
        self.step_count = 0
        self.sumsq = torch.zeros(1, device=bucket.flat.device, dtype=torch.float32)
        self.param_groups = [{"lr": lr}]        # scheduler-facing view (get_current_lr, hl_module:158-160)

    def step(self, grad_clip=None, world_size=1):
        """grads already SUMMED over ranks; they are scaled by 1/world_size here."""
        b = self.bucket
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        clip = float(grad_clip) if grad_clip else 0.0
        if clip > 0:
            self.sumsq.zero_()
            ops.sumsq(b.grad, self.sumsq)
        ops.adam_step(b.flat, b.grad, self.m, self.v, lr, self.betas[0], self.betas[1], self.eps, self.step_count,
                      gscale=1.0 / world_size, clip=clip, sumsq_buf=self.sumsq)

    def state_dict(self):
        return {"m": self.m, "v": self.v, "step": self.step_count, "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"])
        self.v.copy_(sd["v"])
        self.step_count = int(sd["step"])
        self.param_groups[0]["lr"] = sd["lr"]


def allreduce_grads(bucket):
    """One RCCL all-reduce (sum) of the whole gradient bucket (no-op without torch.distributed)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(bucket.grad, op=dist.ReduceOp.SUM)
        return dist.get_world_size()
    return 1


def train_step(model, bucket, optim, inputs, target, neg_weight, grad_clip=None):
    """zero_grad -> forward -> SNRLP loss.mean() -> backward -> all-reduce -> clip -> Adam
    (tain_val.py:66-80 + hl_module:303-321,430-441).  Returns the (device) loss scalar."""
    from .functional import SnrlpLossFn
    from . import ops
    bucket.zero_grad()
    est = model(inputs)["output"]
    loss, _ = SnrlpLossFn.apply(est, target, neg_weight)
    ops.absmax_hints_clear()
    loss.backward()
    ops.absmax_hints_clear()
    world = allreduce_grads(bucket)
    optim.step(grad_clip=grad_clip, world_size=world)
    return loss.detach()
