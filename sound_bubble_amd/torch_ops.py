"""The hot path as dispatcher-registered PyTorch operators (`torch.library.custom_op` + `register_autograd`; BASELINE
north_star: "through PyTorch-ROCm custom ops").

Three operators in the `sound_bubble` namespace, at the seam the reference itself has (the `Net` plug-in class selected by
dotted path in the experiment JSON, `hl_module:22-113`, and the loss classes next to it), each with a fake (meta)
implementation, so FakeTensor propagation, `torch.compile(fullgraph=True)` and `torch.export` trace through the
separator as ONE opaque node per call:

  sound_bubble::separate(mixture, dis_embed?, params[], state[], model, pad, train, grad_bucket?) -> Tensor[]
        [output, *next_state (streaming.flatten_state order), handle] -- `Net.forward` (net.py:70-93 of both reference
        families).  backward: sound_bubble::separate_backward(handle, d_output, model) -> Tensor[] /
        sound_bubble::separate_backward_bucket(handle, d_output, model, grad_bucket(mutated)) -> Tensor (one gradient per
        entry of params[]; an EMPTY tensor where the HIP reductions have already accumulated into the parameter's flat
        gradient buffer -- train.FlatBucket -- or the parameter takes none).
  sound_bubble::snrlp_loss(est, gt, neg_weight) -> (loss, loss_vec, d_est)          -- SNRLP.py:17-42
  sound_bubble::multireso_fuse_loss(est, gt, cfg) -> (loss, d_est)                  -- MultiResoLoss.py:6-31

Why one operator for the separator and not one per stage: the stage functions (functional.py) hand each other kernel-private
state that is not a tensor in the dispatcher's sense -- BPTT records in tile-blocked layouts, the overlapped schedules'
slab flags (`ops.FwdOverlap`), the FiLM bank every block's backward adds into, gradient targets inside the flat bucket --
and run ~20 times per step; cutting them into dispatcher ops would turn every such hand-over into an op output and
serialise the overlapped schedules at op boundaries.  The operator below keeps the stage graph INSIDE: its CUDA
implementation runs `Net.forward` with autograd recording re-enabled (the dispatcher calls it below the Autograd key),
parks the graph root under the integer `handle` it returns, and the backward operator replays that graph
(`torch.autograd.grad`).  Same kernels, same launches, same numbers as `model(inputs)` -- tests/test_gpu_torch_ops.py
holds the two bit-identical and runs `torch.library.opcheck` on all three.

`separate_module(model)` wraps a Net so that `wrapped(inputs, input_state, pad)` goes through the operator.
"""
import json
import weakref
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import functional as Fn
from .streaming import flatten_state, unflatten_state_buffers

_MODELS = weakref.WeakValueDictionary()     # model id -> Net (the operator's `model` argument)
_PENDING = {}                               # handle -> (graph root, parameter list)
_NEXT = [1]
_AK = None
MAX_PENDING = 8
_MR_CACHE = {}


class _record_autograd:
    """custom-op implementations run with the Autograd dispatch keys excluded (the dispatcher is below them); the stage
    graph inside needs them back for its torch glue ops (pad, views, the FiLM-plane matmuls)."""

    def __enter__(self):
        global _AK
        K = torch._C.DispatchKey
        if _AK is None:
            _AK = [K.AutogradOther, K.AutogradCPU, K.AutogradCUDA, K.AutogradNestedTensor, K.AutogradFunctionality,
                   K.ADInplaceOrView]
        ex = torch._C._dispatch_tls_local_exclude_set()
        self.prev = [k for k in _AK if ex.has(k)]
        for k in _AK:
            torch._C._dispatch_tls_set_dispatch_key_excluded(k, False)
        return self

    def __exit__(self, *a):
        for k in self.prev:
            torch._C._dispatch_tls_set_dispatch_key_excluded(k, True)


def register_model(model) -> int:
    """-> the integer the operators know `model` by (weakly held: the id dies with the module)"""
    mid = model.__dict__.get("_sb_op_id")
    if mid is None or _MODELS.get(mid) is not model:
        mid = _NEXT[0]
        _NEXT[0] += 1
        object.__setattr__(model, "_sb_op_id", mid)
        _MODELS[mid] = model
    return mid


def _model(mid):
    m = _MODELS.get(mid)
    if m is None:
        raise RuntimeError(f"sound_bubble::separate: no live model registered under id {mid} (torch_ops.register_model)")
    return m


def _state_names(m, batch):
    return list(flatten_state(m._make_buffers(batch, lambda *s: tuple(s))).keys())


def _out_len(m, n, pad):
    """samples out for n samples in (net.py:8-18,70-93: pad to whole chunks + look-ahead, crop the chunk padding)"""
    look = m.stft_pad_size if m.lookahead else 0
    if pad:
        return n
    if (n - look) % m.stft_chunk_size or n <= look:
        raise RuntimeError(f"pad=False needs {look} + k * {m.stft_chunk_size} samples, got {n}")
    return n - look


@torch.library.custom_op("sound_bubble::separate", mutates_args=(), device_types="cuda")
def separate(mixture: Tensor, dis_embed: Optional[Tensor], params: List[Tensor], state: List[Tensor], model: int,
             pad: bool, train: bool, grad_bucket: Optional[Tensor] = None) -> List[Tensor]:
    # grad_bucket (flat_grad_bucket(model) or None): not touched here -- carried to separate_backward, which adds the flat-
    # bucket parameters' gradients into it and declares so; an operator INPUT so that a tracing compiler sees it as a graph input
    m = _model(model)
    own = list(m.parameters())
    if len(params) != len(own) or any(a.shape != b.shape for a, b in zip(params, own)):
        raise RuntimeError("sound_bubble::separate: params[] must match list(model.parameters()) of the registered model")
    B = mixture.shape[0]
    names = _state_names(m, B)
    st = None
    if state:
        if len(state) != len(names):
            raise RuntimeError(f"sound_bubble::separate: {len(names)} state buffers expected, got {len(state)}")
        st = unflatten_state_buffers(names, state)
    inputs = {"mixture": mixture}
    if dis_embed is not None:
        inputs["dis_embed"] = dis_embed
    # the usual call hands the module's own parameters; any other tensors of the same shapes (a functional caller, opcheck's
    # cloned arguments) are swapped in for the call, torch.func.functional_call style -- the result is a function of params[]
    live = all(a is b or (a.data_ptr() == b.data_ptr() and a.requires_grad == b.requires_grad) for a, b in zip(params, own))
    if not live:
        own = list(params)
    record = train and any(p.requires_grad for p in own)
    with _record_autograd(), torch.set_grad_enabled(record):
        if live:
            res = m(inputs, st, pad=pad)
        else:
            res = torch.func.functional_call(m, {n: p for (n, _), p in zip(m.named_parameters(), params)}, (inputs, st),
                                             {"pad": pad})
    out = res["output"]
    nxt = flatten_state(res["next_state"])
    assert list(nxt.keys()) == names
    handle = torch.zeros(1, dtype=torch.int64)
    if record and out.requires_grad:
        h = _NEXT[0]
        _NEXT[0] += 1
        handle[0] = h
        # a forward whose backward never runs (an exception, a dropped loss) keeps its autograd graph -- BPTT records and
        # all -- parked here: at most MAX_PENDING of them, and the next one RAISES instead of silently dropping the oldest
        # (whose backward would then fail far from the cause); drop_pending() releases them deliberately
        if len(_PENDING) >= MAX_PENDING:
            raise RuntimeError(f"sound_bubble::separate: {len(_PENDING)} recorded forwards are waiting for their backward "
                               f"(MAX_PENDING = {MAX_PENDING}): run or drop them (torch_ops.drop_pending()), or call with "
                               f"train=False / under torch.no_grad() when no backward will follow")
        _PENDING[h] = (out, own)
    # outputs may not alias inputs: buffers the forward passed through untouched are copied
    ins = {t.data_ptr() for t in state}
    # (the cropped output of a padded call is a view: operators return dense tensors)
    return [out.detach().contiguous()] + [(v.detach().clone() if v.data_ptr() in ins else v.detach()) for v in nxt.values()] + [handle]


@separate.register_fake
def _(mixture, dis_embed, params, state, model, pad, train, grad_bucket=None):
    m = _model(model)
    B = mixture.shape[0]
    shapes = flatten_state(m._make_buffers(B, lambda *s: tuple(s)))
    out = mixture.new_empty((B, m.num_src, _out_len(m, mixture.shape[-1], pad)), dtype=torch.float32)
    return [out] + [mixture.new_empty(s, dtype=torch.float32) for s in shapes.values()] + \
        [torch.empty(1, dtype=torch.int64, device="cpu")]


def drop_pending():
    """release every recorded forward that is still waiting for its backward (their graphs and BPTT records); -> how many"""
    n = len(_PENDING)
    _PENDING.clear()
    return n


def flat_grad_bucket(model):
    """the ONE flat gradient buffer train.FlatBucket pointed the parameters' .grad views into (None without a bucket):
    what separate_backward declares as mutated"""
    base = None
    for p in model.parameters():
        g = Fn.direct_grad_target(p)
        if g is None:
            continue
        b = g._base if g._base is not None else g
        if base is None:
            base = b
        elif b.untyped_storage().data_ptr() != base.untyped_storage().data_ptr():
            raise RuntimeError("sound_bubble::separate_backward: the parameters' gradient views live in more than one flat "
                               "buffer; build ONE train.FlatBucket per model")
    return base


def _run_backward(handle, d_output, model, grad_bucket):
    ent = _PENDING.pop(int(handle), None)
    if ent is None:
        raise RuntimeError("sound_bubble::separate_backward: this forward recorded no graph (train=False / no parameter "
                           "requires grad), its backward has already run, or it was dropped (torch_ops.drop_pending)")
    have = flat_grad_bucket(_model(model))
    if (have is None) != (grad_bucket is None) or (have is not None and (grad_bucket.numel() != have.numel() or
                                                                         grad_bucket.dtype != have.dtype or not grad_bucket.is_contiguous())):
        _PENDING[int(handle)] = ent
        raise RuntimeError("sound_bubble::separate_backward: grad_bucket must be (a tensor standing in for) "
                           "torch_ops.flat_grad_bucket(model) -- the buffer the flat-bucket parameters' gradients are added "
                           "into (separate_backward_bucket); the plain separate_backward is for models WITHOUT a train.FlatBucket")
    # The gradients go into the tensor that was PASSED: under a functionalising compiler that is a copy of the bucket (the
    # declared mutation is replayed onto the real one afterwards), so the parameters' .grad views are re-pointed into it for
    # the duration of the call -- same offsets; the identity when the real bucket was passed.
    repoint = []
    if have is not None and grad_bucket.data_ptr() != have.data_ptr():
        flat = grad_bucket.view(-1)
        for p in _model(model).parameters():
            g = Fn.direct_grad_target(p)
            if g is not None:
                off = g.storage_offset() - have.storage_offset()
                repoint.append((p, g))
                p.grad = flat[off:off + g.numel()].view(g.shape)
    try:
        return _backward_into(ent, d_output)
    finally:
        for p, g in repoint:
            p.grad = g


def _backward_into(ent, d_output):
    root, own = ent
    need = [p for p in own if p.requires_grad]
    with _record_autograd():
        got = torch.autograd.grad(root, need, d_output.contiguous(), allow_unused=True)
    from . import ops
    ops.deferred_join()                      # (idempotent: the engine's final callback has normally done it)
    it = iter(got)
    outs = []
    seen = {d_output.untyped_storage().data_ptr()}
    for p in own:
        g = next(it) if p.requires_grad else None
        if _grad_in_place(p):
            # the HIP reductions have added into the parameter's flat gradient buffer already (functional._GradTargets);
            # whatever came back through torch glue ops joins it there
            if g is not None:
                p.grad.add_(g)
            outs.append(p.new_empty(0))
        else:
            if g is None:
                g = torch.zeros_like(p)
            else:
                # operator outputs may not share storage (the FiLM planes' gradients are slices of one bank, d_output may come
                # back as somebody's gradient unchanged)
                g = g.contiguous()
                key = g.untyped_storage().data_ptr()
                if key in seen or g._base is not None:
                    g = g.clone()
                seen.add(g.untyped_storage().data_ptr())
            outs.append(g)
    return outs


# Two backward operators, because a functionalising compiler only takes a MUTATING custom operator whose results are plain
# tensors (torch/_higher_order_ops/auto_functionalize.py: no Tensor[] returns):
#   separate_backward(handle, d_output, model) -> Tensor[]: one gradient per parameter, nothing mutated -- for a model whose
#       parameters own no flat gradient bucket;
#   separate_backward_bucket(handle, d_output, model, grad_bucket) -> Tensor: for a model under train.FlatBucket.  The HIP
#       weight-gradient reductions ADD into the per-parameter views of grad_bucket (= flat_grad_bucket(model)) while the
#       backward runs: declared (mutates_args), so the write is visible to the compiler, which may neither reorder the call
#       against readers of the bucket (the all-reduce, the optimiser) nor dedupe it.  Returns the number of parameters served.
@torch.library.custom_op("sound_bubble::separate_backward", mutates_args=(), device_types="cuda")
def separate_backward(handle: Tensor, d_output: Tensor, model: int) -> List[Tensor]:
    return _run_backward(handle, d_output, model, None)


@torch.library.custom_op("sound_bubble::separate_backward_bucket", mutates_args=("grad_bucket",), device_types="cuda")
def separate_backward_bucket(handle: Tensor, d_output: Tensor, model: int, grad_bucket: Tensor) -> Tensor:
    m = _model(model)
    stray = [n for n, p in m.named_parameters() if p.requires_grad and Fn.direct_grad_target(p) is None]
    if stray:
        raise RuntimeError(f"sound_bubble::separate_backward_bucket: parameters outside the flat gradient bucket require "
                           f"gradients ({stray[:3]} ...): put every trainable parameter into the train.FlatBucket")
    outs = _run_backward(handle, d_output, model, grad_bucket)
    return torch.full((1,), len(outs), dtype=torch.int64)


@separate_backward_bucket.register_fake
def _(handle, d_output, model, grad_bucket):
    return torch.empty(1, dtype=torch.int64, device="cpu")


def _grad_in_place(p):
    """the rule both the implementation and its fake follow: no gradient tensor is returned for a frozen parameter or one
    whose gradient lives in train.FlatBucket's flat buffer (accumulated there, as .backward() would)"""
    return (not p.requires_grad) or Fn.direct_grad_target(p) is not None


@separate_backward.register_fake
def _(handle, d_output, model):
    return [d_output.new_empty((0,) if _grad_in_place(p) else tuple(p.shape)) for p in _model(model).parameters()]


def _separate_setup(ctx, inputs, output):
    ctx.model = inputs[4]
    ctx.handle = output[-1]
    ctx.n_state = len(inputs[3])
    ctx.n_params = len(inputs[2])
    # The flat bucket is NOT a saved tensor: it is zeroed (FlatBucket.zero_grad) and added into (every backward of the model)
    # in place between this forward and its backward as a matter of course, and autograd's saved-tensor version check would
    # turn both into "modified by an inplace operation" errors.  Its VALUE at forward time is irrelevant -- the backward
    # only adds into whatever it holds when it runs -- so it rides along as a plain attribute.
    ctx.bucket = inputs[7] if len(inputs) > 7 else None
    ctx.set_materialize_grads(False)


def _separate_bwd(ctx, grads):
    # (a trailing argument left at its default -- grad_bucket = None -- is not among the inputs autograd shows here)
    n_in = len(ctx.needs_input_grad)
    d_out = grads[0]
    if d_out is None:
        return (None, None, [None] * ctx.n_params, [None] * ctx.n_state, None, None, None, None)[:n_in]
    bucket = ctx.bucket
    if bucket is not None:        # every gradient lands in the flat bucket (declared mutated): nothing comes back through autograd
        torch.ops.sound_bubble.separate_backward_bucket(ctx.handle, d_out, ctx.model, bucket)
        return (None, None, [None] * ctx.n_params, [None] * ctx.n_state, None, None, None, None)[:n_in]
    gs = torch.ops.sound_bubble.separate_backward(ctx.handle, d_out, ctx.model)
    # the state entries are carried values (net.py:88-93: detached between chunks in the reference's streaming use): no
    # gradient flows into them, and none into the waveform
    return (None, None, [g if g.numel() else None for g in gs], [None] * ctx.n_state, None, None, None, None)[:n_in]


separate.register_autograd(_separate_bwd, setup_context=_separate_setup)


# ---- losses ----
@torch.library.custom_op("sound_bubble::snrlp_loss", mutates_args=(), device_types="cuda")
def snrlp_loss(est: Tensor, gt: Tensor, neg_weight: float, mode: int = 0) -> Tuple[Tensor, Tensor, Tensor]:
    """(batch-mean loss, per-utterance loss vector, d loss / d est) -- src/losses/SNRLP.py:17-42; the fused kernel forms the
    gradient in the same two passes (csrc/sb_elementwise.hip); mode = ops.SNR_LOSS_MODES[snr_loss_name] (0: 'snr')"""
    with _record_autograd(), torch.enable_grad():
        e = est.detach().requires_grad_(True)
        loss, lv = Fn.SnrlpLossFn.apply(e, gt, neg_weight, mode)
        (g,) = torch.autograd.grad(loss, e)
    return loss.detach(), lv.detach(), g


@snrlp_loss.register_fake
def _(est, gt, neg_weight, mode=0):
    return est.new_empty(()), est.new_empty((est.shape[0],)), torch.empty_like(est)


def _snrlp_setup(ctx, inputs, output):
    ctx.save_for_backward(output[2])
    ctx.set_materialize_grads(False)


def _snrlp_bwd(ctx, g_loss, g_lv, g_d):
    (d,) = ctx.saved_tensors
    return ((d * g_loss if g_loss is not None else None), None, None, None)[:len(ctx.needs_input_grad)]


snrlp_loss.register_autograd(_snrlp_bwd, setup_context=_snrlp_setup)


@torch.library.custom_op("sound_bubble::multireso_fuse_loss", mutates_args=(), device_types="cuda")
def multireso_fuse_loss(est: Tensor, gt: Tensor, cfg: str) -> Tuple[Tensor, Tensor]:
    """(loss, d loss / d est) -- src/losses/MultiResoLoss.py:6-31; cfg = JSON of losses.MultiResoFuseLoss's settings"""
    from .losses import MultiResoFuseLoss
    mod = _MR_CACHE.get(cfg)
    if mod is None:
        mod = _MR_CACHE[cfg] = MultiResoFuseLoss(**json.loads(cfg))
    with _record_autograd(), torch.enable_grad():
        e = est.detach().requires_grad_(True)
        loss = mod(e, gt)
        loss = loss.mean() if loss.dim() else loss
        (g,) = torch.autograd.grad(loss, e)
    return loss.detach(), g


@multireso_fuse_loss.register_fake
def _(est, gt, cfg):
    return est.new_empty(()), torch.empty_like(est)


def _mr_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])
    ctx.set_materialize_grads(False)


def _mr_bwd(ctx, g_loss, g_d):
    (d,) = ctx.saved_tensors
    return (d * g_loss if g_loss is not None else None), None, None


multireso_fuse_loss.register_autograd(_mr_bwd, setup_context=_mr_setup)


class _SeparateModule(torch.nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model
        self.model_id = register_model(model)

    def forward(self, inputs, input_state=None, pad=True):
        m = self.model
        B = inputs["mixture"].shape[0]
        state = list(flatten_state(input_state).values()) if input_state is not None else []
        outs = torch.ops.sound_bubble.separate(inputs["mixture"], inputs.get("dis_embed"), list(m.parameters()), state,
                                               self.model_id, pad, torch.is_grad_enabled(),
                                               flat_grad_bucket(m) if torch.is_grad_enabled() else None)
        names = _state_names(m, B)
        nxt = {}
        for name, buf in zip(names, outs[1:-1]):
            node = nxt
            path = name.split("::")
            for part in path[:-1]:
                node = node.setdefault(part, {})
            node[path[-1]] = buf
        return {"output": outs[0], "next_state": nxt}


def separate_module(model):
    """`model` (a sound_bubble_amd Net) behind the dispatcher operator: same call signature, same result dict"""
    return _SeparateModule(model)
