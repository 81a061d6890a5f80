"""Training-step plumbing for the hot path: flat parameter/gradient bucket, fused clip + Adam (HIP),
one RCCL all-reduce per step.

Replaces, for the data-parallel path, nn.DataParallel's per-step broadcast/scatter/gather/reduce
(src/hl_modules/distance_based_hl_module.py:34-35) and PLModule.reset_grad/backprop (:430-441):
one process per GPU, identical replicas, ONE all-reduce over a single flat fp32 gradient buffer
(2.0 MB big / 0.9 MB small), then clip_grad_norm_ semantics and Adam on every rank.
"""
import torch

from . import ops
from .forms import bump_weight_epoch


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
            p._sb_flat_grad = True          # functional._GradTargets: the HIP reductions may accumulate into .grad directly
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
        # optimiser steps the kernel turned into no-ops because the watchdog word of the guarded schedules was set (a bounded
        # wait of that step's launches gave up: garbage gradients).  Read with the word itself (ops.check_sched_status*).
        self.skipped = torch.zeros(1, device=bucket.flat.device, dtype=torch.int32)
        self.param_groups = [{"lr": lr}]        # scheduler-facing view (get_current_lr, hl_module:158-160)

    def step(self, grad_clip=None, world_size=1):
        """grads already SUMMED over ranks; they are scaled by 1/world_size here."""
        b = self.bucket
        self.step_count += 1
        lr = self.param_groups[0]["lr"]
        clip = float(grad_clip) if grad_clip else 0.0
        if clip > 0:
            ops.sumsq(b.grad, self.sumsq, accumulate=False)
        ops.adam_step(b.flat, b.grad, self.m, self.v, lr, self.betas[0], self.betas[1], self.eps, self.step_count,
                      gscale=1.0 / world_size, clip=clip, sumsq_buf=self.sumsq, skipped=self.skipped)
        bump_weight_epoch()       # the kernel wrote the parameters behind torch's version counters: weight forms are stale

    def state_dict(self):
        """torch.optim.Adam.state_dict() layout (what the reference's dump_state stores under 'optimizer',
        hl_module:141-156): per-parameter {step, exp_avg, exp_avg_sq} indexed in model.parameters() order, sliced out of
        the flat moment buffers, plus one param_group -- so the reference's load_state can resume from our last.pt."""
        b = self.bucket
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(b.params, b.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.m[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.v[o:o + n].view(p.shape).clone()}
        group = {"lr": self.param_groups[0]["lr"], "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": False, "params": list(range(len(b.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's layout (a reference last.pt / best.pt) and this class's round-1 private layout
        {m, v, step, lr}."""
        b = self.bucket
        if "m" in sd and "param_groups" not in sd:                  # round-1 layout
            self.m.copy_(sd["m"])
            self.v.copy_(sd["v"])
            self.step_count = int(sd["step"])
            self.param_groups[0]["lr"] = sd["lr"]
            return
        groups = sd["param_groups"]
        order = [i for g in groups for i in g["params"]]
        if len(order) != len(b.params):
            raise ValueError(f"optimizer state covers {len(order)} parameters, the model has {len(b.params)}")
        g0 = groups[0]
        if g0.get("weight_decay", 0) or g0.get("amsgrad", False) or g0.get("maximize", False):
            raise NotImplementedError("weight_decay / amsgrad / maximize optimizer states are not supported")
        self.m.zero_()
        self.v.zero_()
        steps = set()
        for slot, (p, o) in zip(order, zip(b.params, b.offsets)):
            st = sd["state"].get(slot)
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {slot}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(p.shape)}")
            n = p.numel()
            self.m[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter Adam step counts differ ({sorted(steps)}): one shared step is supported")
        self.step_count = steps.pop() if steps else 0
        self.param_groups[0]["lr"] = g0["lr"]
        self.betas, self.eps = tuple(g0.get("betas", self.betas)), g0.get("eps", self.eps)


def _via_host(t):
    """gloo (CPU tests; several ranks on one GPU in the -m gpu tests) moves device tensors through the host"""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend() == "gloo"


def _broadcast(t, src):
    import torch.distributed as dist
    if _via_host(t):
        h = t.cpu()
        dist.broadcast(h, src)
        t.copy_(h)
    else:
        dist.broadcast(t, src)


def broadcast_replica(bucket, optim=None, src=0):
    """Make every rank an identical replica of rank `src` (start of training / after a resume): parameters, and the Adam
    moments + step when an optimizer is given.  nn.DataParallel re-broadcasts the parameters every step
    (hl_module:34-35); with one process per GPU once is enough, because every rank applies the same update."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    _broadcast(bucket.flat, src)
    bump_weight_epoch()
    if optim is not None:
        _broadcast(optim.m, src)
        _broadcast(optim.v, src)
        meta = torch.tensor([float(optim.step_count), float(optim.param_groups[0]["lr"])], device=bucket.flat.device,
                            dtype=torch.float64)
        _broadcast(meta, src)
        optim.step_count, optim.param_groups[0]["lr"] = int(meta[0].item()), float(meta[1].item())


FORCE_ALLREDUCE = False     # run the collective in a world of one as well (bench.py --init-dist, the RCCL smoke test)


def allreduce_grads(bucket, force=False):
    """One RCCL all-reduce (sum) of the whole gradient bucket (no-op without torch.distributed).  In a process group of ONE
    rank the collective is skipped unless `force` / FORCE_ALLREDUCE asks for it (RCCL then runs its kernel on the bucket --
    a sum over one rank -- in stream order between the backward and the optimiser: what tests/test_gpu_distributed.py checks)."""
    import torch.distributed as dist
    from . import ops
    ops.deferred_join()                      # reductions still on the library's side stream (normally joined at the end of backward)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force or FORCE_ALLREDUCE):
        if _via_host(bucket.grad):
            h = bucket.grad.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            bucket.grad.copy_(h)
        else:
            dist.all_reduce(bucket.grad, op=dist.ReduceOp.SUM)      # ONE RCCL all-reduce of the whole bucket
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
    ops.deferred_join()                      # (the engine's final callback has done it: idempotent)
    ops.absmax_hints_clear()
    world = allreduce_grads(bucket)
    optim.step(grad_clip=grad_clip, world_size=world)
    return loss.detach()
