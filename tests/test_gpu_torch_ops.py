"""sound_bubble::separate / ::snrlp_loss / ::multireso_fuse_loss (torch.library custom ops, sound_bubble_amd/torch_ops.py) on
the MI355X: the operator path is the module path -- same kernels, bit-identical outputs, states and gradients -- held to the
reference goldens, checked by torch.library.opcheck, and traced whole by torch.compile(fullgraph=True)."""
import numpy as np
import pytest

from conftest import load_golden, golden_state_dict, rel_l2, flatten_state

pytestmark = pytest.mark.gpu

CASES = [("tiny_big", "NetDisEmbd3"), ("tiny_small", "NetOptim")]


def _build(torch, name, cls):
    import sound_bubble_amd as sb
    rec, params, flavour = load_golden(name)
    m = getattr(sb, cls)(**params)
    m.load_state_dict(golden_state_dict(rec, torch), strict=True)
    inp = {"mixture": torch.from_numpy(rec["mixture"]).cuda()}
    if "dis_embed" in rec:
        inp["dis_embed"] = torch.from_numpy(rec["dis_embed"]).cuda()
    return rec, m.cuda(), inp


@pytest.mark.parametrize("name,cls", CASES)
def test_operator_forward_and_streaming_equal_the_module(name, cls):
    import torch
    from sound_bubble_amd import torch_ops as T
    rec, m, inp = _build(torch, name, cls)
    w = T.separate_module(m)
    with torch.no_grad():
        a, b = m(inp), w(inp)
    assert torch.equal(a["output"], b["output"])
    assert rel_l2(b["output"].cpu().numpy(), rec["output"]) < 2e-5
    fa, fb = flatten_state(a["next_state"]), flatten_state(b["next_state"])
    assert list(fa) == list(fb)
    for k in fa:
        assert np.array_equal(fa[k], fb[k]), k
        assert rel_l2(fb[k], rec["next_state::" + k]) < 2e-5, k
    # the result is a function of params[]: clones give the same output, changed clones a different one, the module untouched
    ps = [p.detach().clone() for p in m.parameters()]
    mid = T.register_model(m)
    with torch.no_grad():
        o = torch.ops.sound_bubble.separate(inp["mixture"], inp.get("dis_embed"), ps, [], mid, True, False)
        assert torch.equal(o[0], a["output"])
        ps[-1].mul_(1.5)                       # the output deconvolution's bias
        o = torch.ops.sound_bubble.separate(inp["mixture"], inp.get("dis_embed"), ps, [], mid, True, False)
        assert not torch.equal(o[0], a["output"])
        assert torch.equal(m(inp)["output"], a["output"])
    # three chunks with the state carried through the operator (edge/causal_infer.py:15-26)
    hop, look = m.stft_chunk_size, m.stft_pad_size
    mix = inp["mixture"]
    n = (mix.shape[-1] - look) // hop // 3 * hop
    st_a = st_b = None
    with torch.no_grad():
        for c in range(3):
            fr = dict(inp, mixture=mix[..., c * n: (c + 1) * n + look].contiguous())
            ra, rb = m(fr, st_a, pad=False), w(fr, st_b, pad=False)
            st_a, st_b = ra["next_state"], rb["next_state"]
            assert torch.equal(ra["output"], rb["output"]), c
    for k, v in flatten_state(st_b).items():
        assert np.array_equal(v, flatten_state(st_a)[k]), k


@pytest.mark.parametrize("bucket", [False, True], ids=["autograd-grads", "flat-bucket"])
@pytest.mark.parametrize("name,cls", CASES)
def test_operator_gradients_equal_the_module_and_the_goldens(name, cls, bucket):
    import torch
    from sound_bubble_amd import torch_ops as T, ops
    from sound_bubble_amd.functional import SnrlpLossFn
    from sound_bubble_amd.train import FlatBucket
    rec, m, inp = _build(torch, name, cls)
    m.train()
    tgt = torch.from_numpy(rec["target"]).cuda()
    fb = FlatBucket(m) if bucket else None
    loss, _ = SnrlpLossFn.apply(m(inp)["output"], tgt, 100.0)
    loss.backward()
    want = {k: p.grad.clone() for k, p in m.named_parameters()}
    if fb is not None:
        fb.zero_grad()
    else:
        m.zero_grad(set_to_none=True)
    w = T.separate_module(m)
    est = w(inp)["output"]
    l2, lv, _ = torch.ops.sound_bubble.snrlp_loss(est, tgt, 100.0)
    assert float(l2.detach()) == float(loss.detach())
    np.testing.assert_allclose(lv.detach().cpu().numpy(), rec["loss_vec"], rtol=1e-4, atol=1e-4)
    l2.backward()
    ops.check_sched_status()
    assert not T._PENDING
    for k, p in m.named_parameters():
        assert p.grad is not None, k
        # atomics in the weight-gradient reductions: run-to-run differences in the last bits
        assert rel_l2(p.grad.cpu().numpy(), want[k].cpu().numpy()) < 1e-5 or float(want[k].abs().max()) == 0, k
        g = rec["grad::" + k]
        e = rel_l2(p.grad.cpu().numpy(), g) if np.abs(g).max() > 0 else float(p.grad.abs().max())
        assert e < 2e-4, (k, e)


def test_bucket_may_be_zeroed_or_added_into_between_forward_and_backward():
    """the flat gradient bucket is mutated in place between a forward and its backward as a matter of
    course -- (a) forward, zero_grad(), backward; (b) two forwards feeding one loss (the first node's backward adds into the
    bucket before the second node runs).  Neither may trip autograd's saved-tensor version check, and both must leave the
    gradients of the plain module path in the bucket."""
    import torch
    from sound_bubble_amd import torch_ops as T
    from sound_bubble_amd.functional import SnrlpLossFn
    from sound_bubble_amd.train import FlatBucket
    rec, m, inp = _build(torch, "tiny_small", "NetOptim")
    m.train()
    fb = FlatBucket(m)
    tgt = torch.from_numpy(rec["target"]).cuda()
    fb.zero_grad()
    loss, _ = SnrlpLossFn.apply(m(inp)["output"], tgt, 100.0)
    loss.backward()
    want = fb.grad.clone()
    w = T.separate_module(m)
    # (a) zero_grad AFTER the forward
    fb.grad.fill_(7.0)
    est = w(inp)["output"]
    fb.zero_grad()
    torch.ops.sound_bubble.snrlp_loss(est, tgt, 100.0)[0].backward()
    assert rel_l2(fb.grad.cpu().numpy(), want.cpu().numpy()) < 1e-5
    # (b) two pending forwards, one backward: both nodes add into the same bucket
    fb.zero_grad()
    e1, e2 = w(inp)["output"], w(inp)["output"]
    assert len(T._PENDING) == 2
    l = torch.ops.sound_bubble.snrlp_loss(e1, tgt, 100.0)[0] + torch.ops.sound_bubble.snrlp_loss(e2, tgt, 100.0)[0]
    l.backward()
    assert not T._PENDING
    assert rel_l2(fb.grad.cpu().numpy(), 2.0 * want.cpu().numpy()) < 1e-5
    # (c) Module.zero_grad(set_to_none=False) zeroes the views in place: same thing
    est = w(inp)["output"]
    m.zero_grad(set_to_none=False)
    torch.ops.sound_bubble.snrlp_loss(est, tgt, 100.0)[0].backward()
    assert rel_l2(fb.grad.cpu().numpy(), want.cpu().numpy()) < 1e-5


def test_parked_forwards_raise_at_the_limit_and_backward_declares_its_bucket():
    """a ninth recorded forward without a backward RAISES (it used to drop the oldest graph silently);
    drop_pending() releases them; the backward operator takes the flat gradient bucket it adds into as a declared-mutated
    argument and refuses any other buffer."""
    import torch
    from sound_bubble_amd import torch_ops as T
    from sound_bubble_amd.train import FlatBucket
    rec, m, inp = _build(torch, "tiny_small", "NetOptim")
    m.train()
    fb = FlatBucket(m)
    mid = T.register_model(m)
    T.drop_pending()
    args = (inp["mixture"], None, list(m.parameters()), [], mid, True, True, T.flat_grad_bucket(m))
    assert T.flat_grad_bucket(m).data_ptr() == fb.grad.data_ptr()
    outs = [torch.ops.sound_bubble.separate(*args) for _ in range(T.MAX_PENDING)]
    with pytest.raises(RuntimeError, match="waiting for their backward"):
        torch.ops.sound_bubble.separate(*args)
    assert len(T._PENDING) == T.MAX_PENDING
    o = outs[0]
    d = torch.ones_like(o[0])
    with pytest.raises(RuntimeError, match="grad_bucket"):
        torch.ops.sound_bubble.separate_backward_bucket(o[-1], d, mid, torch.zeros(16, device="cuda"))   # not a bucket of this model
    with pytest.raises(RuntimeError, match="grad_bucket"):
        torch.ops.sound_bubble.separate_backward(o[-1], d, mid)                                         # the bucket-free operator
    fb.zero_grad()
    n = torch.ops.sound_bubble.separate_backward_bucket(o[-1], d, mid, fb.grad)
    assert int(n) == len(list(m.parameters())) and float(fb.grad.abs().max()) > 0               # gradients landed in the bucket
    # the operator writes into the tensor it was HANDED (what a functionalising compiler relies on: it passes a copy and
    # replays the declared mutation): a stand-in buffer receives the gradients, the model's own bucket stays as it was
    stand_in, before = torch.zeros_like(fb.grad), fb.grad.clone()
    torch.ops.sound_bubble.separate_backward_bucket(outs[2][-1], d, mid, stand_in)
    assert torch.equal(fb.grad, before) and rel_l2(stand_in.cpu().numpy(), before.cpu().numpy()) < 1e-5
    assert all(p.grad.data_ptr() >= fb.grad.data_ptr() for p in m.parameters())                 # .grad views restored
    assert T.drop_pending() == T.MAX_PENDING - 2 and not T._PENDING
    with pytest.raises(RuntimeError, match="dropped"):
        torch.ops.sound_bubble.separate_backward_bucket(outs[1][-1], d, mid, fb.grad)
    s = str(torch.ops.sound_bubble.separate_backward_bucket.default._schema)
    assert "!) grad_bucket" in s, s


def test_opcheck():
    import torch
    from sound_bubble_amd import torch_ops as T
    rec, m, inp = _build(torch, "tiny_big", "NetDisEmbd3")
    m.train()
    mid = T.register_model(m)
    utils = ("test_schema", "test_faketensor", "test_autograd_registration")
    args = (inp["mixture"], inp["dis_embed"], list(m.parameters()), [], mid, True, True)
    torch.library.opcheck(torch.ops.sound_bubble.separate.default, args, test_utils=utils)
    T._PENDING.clear()
    with torch.no_grad():
        st = list(torch.ops.sound_bubble.separate(*args[:4], mid, True, False)[1:-1])
    look, hop = m.stft_pad_size, m.stft_chunk_size
    torch.library.opcheck(torch.ops.sound_bubble.separate.default,
                          (inp["mixture"][..., : look + 2 * hop].contiguous(), inp["dis_embed"], list(m.parameters()), st, mid,
                           False, False), test_utils=utils)
    est = torch.randn(2, 1, 4800, device="cuda", requires_grad=True)
    gt = torch.randn(2, 1, 4800, device="cuda")
    gt[1] = 0
    torch.library.opcheck(torch.ops.sound_bubble.snrlp_loss.default, (est, gt, 50.0), test_utils=utils)
    cfg = '{"l1_ratio": 10, "sample_rate": 24000, "perceptual_weighting": true, "w_sc": 0, "w_log_mag": 0, "w_lin_mag": 20}'
    torch.library.opcheck(torch.ops.sound_bubble.multireso_fuse_loss.default, (est, gt, cfg), test_utils=utils)


def test_loss_operators_match_the_modules():
    import json
    import torch
    from sound_bubble_amd.losses import MultiResoFuseLoss, SNRLPLoss
    torch.manual_seed(3)
    est0 = torch.randn(3, 1, 9600, device="cuda") * 0.1
    gt = torch.randn(3, 1, 9600, device="cuda") * 0.1
    gt[2] = 0
    kw = dict(l1_ratio=10, sample_rate=24000, perceptual_weighting=True, w_sc=0, w_log_mag=0, w_lin_mag=20)
    for op, mod in ((lambda e: torch.ops.sound_bubble.snrlp_loss(e, gt, 50.0)[0], lambda e: SNRLPLoss(neg_weight=50).mean_loss(e, gt)[0]),
                    (lambda e: torch.ops.sound_bubble.multireso_fuse_loss(e, gt, json.dumps(kw))[0],
                     lambda e: MultiResoFuseLoss(**kw).cuda()(e, gt))):
        a, b = est0.clone().requires_grad_(True), est0.clone().requires_grad_(True)
        la, lb = op(a), mod(b)
        (3.0 * la).backward()
        (3.0 * lb).backward()
        assert abs(float(la.detach()) - float(lb.detach())) <= 1e-6 * abs(float(lb.detach()))
        assert rel_l2(a.grad.cpu().numpy(), b.grad.cpu().numpy()) < 1e-6


def test_train_step_with_flat_bucket_traces_under_torch_compile():
    """... and with the parameters under train.FlatBucket: the backward operator that ADDS into the bucket is declared
    mutating, auto-functionalised by AOT autograd, and the compiled step leaves the same gradients in the bucket"""
    import torch
    from sound_bubble_amd import torch_ops as T
    from sound_bubble_amd.train import FlatBucket
    rec, m, inp = _build(torch, "tiny_small", "NetOptim")
    m.train()
    fb = FlatBucket(m)
    mid = T.register_model(m)
    tgt = torch.from_numpy(rec["target"]).cuda()
    params = list(m.parameters())

    def step(mix, tgt, params, bucket):
        outs = torch.ops.sound_bubble.separate(mix, None, params, [], mid, True, True, bucket)
        return torch.ops.sound_bubble.snrlp_loss(outs[0], tgt, 100.0)[0]

    fb.zero_grad()
    step(inp["mixture"], tgt, params, fb.grad).backward()
    want = fb.grad.clone()
    fb.zero_grad()
    cstep = torch.compile(step, fullgraph=True, backend="aot_eager")
    cstep(inp["mixture"], tgt, params, fb.grad).backward()
    assert float(want.abs().max()) > 0 and rel_l2(fb.grad.cpu().numpy(), want.cpu().numpy()) < 1e-5
    assert not T._PENDING


def test_train_step_traces_whole_under_torch_compile():
    """fullgraph=True: Dynamo / AOT autograd see the separator, the loss and the backward operator as opaque nodes with fake
    implementations -- no graph break, same gradients as eager"""
    import torch
    from sound_bubble_amd import torch_ops as T
    rec, m, inp = _build(torch, "tiny_small", "NetOptim")
    m.train()
    mid = T.register_model(m)
    tgt = torch.from_numpy(rec["target"]).cuda()
    params = list(m.parameters())

    def step(mix, tgt, params):
        outs = torch.ops.sound_bubble.separate(mix, None, params, [], mid, True, True)
        return torch.ops.sound_bubble.snrlp_loss(outs[0], tgt, 100.0)[0]

    step(inp["mixture"], tgt, params).backward()
    want = [p.grad.clone() for p in params]
    m.zero_grad(set_to_none=True)
    cstep = torch.compile(step, fullgraph=True, backend="aot_eager")
    loss = cstep(inp["mixture"], tgt, params)
    loss.backward()
    for (k, p), g in zip(m.named_parameters(), want):
        assert p.grad is not None, k
        assert rel_l2(p.grad.cpu().numpy(), g.cpu().numpy()) < 1e-5 or float(g.abs().max()) == 0, k
        e = rel_l2(p.grad.cpu().numpy(), rec["grad::" + k]) if np.abs(rec["grad::" + k]).max() > 0 else 0.0
        assert e < 2e-4, (k, e)
