"""FilmBankFn (all FiLM scale / shift planes of a forward pass in one autograd node) against the straightforward
per-layer autograd formulation of the reference (Dis_Embed_Conv + FilmLayer 1x1 convolutions,
dis_embd3/tfgridnet_causal.py:51-68,150-173).  Pure torch arithmetic: runs on CPU."""
import torch
import torch.nn.functional as tF


def _setup(n=3, B=2, F=7, C=8, d_in=4, seed=0):
    torch.manual_seed(seed)
    dis = torch.eye(3)[torch.randint(0, 3, (B,))]
    W_e = torch.randn(F * d_in, 3, requires_grad=True)
    lw, lb = torch.rand(d_in, requires_grad=True), torch.randn(d_in, requires_grad=True)
    conv = []
    for _ in range(n):
        conv += [torch.randn(C, d_in, 1, requires_grad=True), torch.randn(C, requires_grad=True),
                 torch.randn(C, d_in, 1, requires_grad=True), torch.randn(C, requires_grad=True)]
    coef = [torch.randn(B, F, C) for _ in range(2 * n)]
    return dis, W_e, lw, lb, conv, coef


def _reference(dis, W_e, lw, lb, conv, coef):
    B, d_in = dis.shape[0], lw.shape[0]
    e = tF.layer_norm(tF.linear(dis, W_e).view(B, -1, d_in), (d_in,), lw, lb, 1e-5)
    planes = []
    for k in range(len(conv) // 4):
        planes.append(tF.linear(e, conv[4 * k][:, :, 0], conv[4 * k + 1]))
        planes.append(tF.linear(e, conv[4 * k + 2][:, :, 0], conv[4 * k + 3]))
    return planes, sum((p * c).sum() for p, c in zip(planes, coef))


def test_film_bank_matches_per_layer_autograd():
    from sound_bubble_amd.functional import FilmBankFn
    dis, W_e, lw, lb, conv, coef = _setup()
    planes_ref, loss_ref = _reference(dis, W_e, lw, lb, conv, coef)
    loss_ref.backward()
    want = [t.grad.clone() for t in (W_e, lw, lb, *conv)]
    for t in (W_e, lw, lb, *conv):
        t.grad = None
    bank = {"n": len(conv) // 4, "G": None}
    planes = FilmBankFn.apply(dis, W_e, lw, lb, bank, *conv)
    for a, b in zip(planes, planes_ref):
        assert torch.allclose(a, b, atol=1e-5, rtol=1e-5)
    sum((p * c).sum() for p, c in zip(planes, coef)).backward()
    for t, w in zip((W_e, lw, lb, *conv), want):
        assert t.grad is not None and torch.allclose(t.grad, w, atol=2e-4, rtol=2e-4), (t.shape, (t.grad - w).abs().max())


def test_film_bank_adds_into_adjacent_flat_bucket_grads_once():
    """with train.FlatBucket the 4n conv parameters' .grad buffers are adjacent: one fused add, autograd is handed None"""
    from sound_bubble_amd.functional import FilmBankFn, _adjacent_grad_region
    from sound_bubble_amd.train import FlatBucket
    dis, W_e, lw, lb, conv, coef = _setup(seed=1)
    _, loss_ref = _reference(dis, W_e, lw, lb, conv, coef)
    loss_ref.backward()
    want = [t.grad.clone() for t in conv]
    holder = torch.nn.Module()
    params = [torch.nn.Parameter(t.detach().clone()) for t in (W_e, lw, lb, *conv)]
    for i, p in enumerate(params):
        holder.register_parameter(f"p{i}", p)
    bucket = FlatBucket(holder)
    bucket.zero_grad()
    assert _adjacent_grad_region(params[3:]) is not None
    bank = {"n": len(conv) // 4, "G": None}
    for rep in (1, 2):                                   # a second backward accumulates (x2), like autograd
        planes = FilmBankFn.apply(dis, params[0], params[1], params[2], bank, *params[3:])
        sum((p * c).sum() for p, c in zip(planes, coef)).backward()
        for p, w in zip(params[3:], want):
            assert torch.allclose(p.grad, rep * w, atol=4e-4, rtol=4e-4)
            assert p.grad.data_ptr() >= bucket.grad.data_ptr()
