"""SNRLPLoss with every snr_loss_name of src/losses/SNRLosses.py:10-29 ('snr', 'sisdr', 'fused', 'max_fused', 'sdsdr', 'full')
on the MI355X against the oracle's restatement (oracle.tfgridnet_oracle.snrlp_loss, float64 autograd): per-sample losses and
d mean(loss) / d est, batches with silent targets (the L1 branch), a DC offset (the zero-mean convention) and an estimate that
is a scaled target (where sisdr and snr part ways)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ["snr", "sisdr", "fused", "max_fused", "sdsdr", "full"]


def _batch(torch, B=6, N=24000, seed=0):
    g = torch.Generator().manual_seed(seed)
    gt = 0.1 * torch.randn(B, 1, N, generator=g)
    est = gt + 0.05 * torch.randn(B, 1, N, generator=g)
    est[1] = 0.3 * gt[1] + 0.001 * torch.randn(1, N, generator=g)      # a scaled target: SI-SDR high, SNR poor
    est[2] += 0.2                                                      # a DC offset: removed by the zero-mean convention
    if B > 5:
        gt[3] = 0.0                                                    # silent targets: neg_weight * L1, shared by both
        gt[5] = 0.0
        est[4] = 2.5 * gt[4] + 0.02 * torch.randn(1, N, generator=g)   # an over-scaled target: sdsdr < sisdr
    else:
        gt[B - 1] = 0.0
    return est, gt


@pytest.mark.parametrize("name", NAMES)
def test_every_snr_loss_name_matches_the_oracle(name):
    import torch
    from oracle.tfgridnet_oracle import snrlp_loss
    from sound_bubble_amd.losses import SNRLPLoss
    est, gt = _batch(torch)
    e64 = est.double().requires_grad_(True)
    want = snrlp_loss(e64, gt.double(), 50.0, name)
    want.mean().backward()
    mod = SNRLPLoss(snr_loss_name=name, neg_weight=50)
    e = est.cuda().requires_grad_(True)
    loss, lv = mod.mean_loss(e, gt.cuda())
    (3.0 * loss).backward()
    np.testing.assert_allclose(lv.detach().cpu().numpy(), want.detach().numpy(), rtol=2e-5, atol=2e-5)
    # (a second evaluation: the moment sums are atomic float adds, so two runs agree to rounding, not to the bit)
    np.testing.assert_allclose(mod(est.cuda(), gt.cuda()).cpu().numpy(), lv.detach().cpu().numpy(), rtol=1e-6, atol=5e-6)
    g, w = e.grad.cpu().double() / 3.0, e64.grad
    for b in range(est.shape[0]):
        err = float((g[b] - w[b]).norm() / w[b].norm())
        assert err < 2e-5, (name, b, err)


def test_snr_loss_names_differ_where_they_should_and_unknown_names_raise():
    import torch
    from sound_bubble_amd.losses import SNRLPLoss
    est, gt = _batch(torch)
    lv = {n: SNRLPLoss(n, 50)(est.cuda(), gt.cuda()).cpu().numpy() for n in NAMES}
    assert lv["sisdr"][1] < lv["snr"][1] - 10                       # the scaled target: sisdr sees ~+50 dB, snr ~3 dB
    assert np.allclose(lv["max_fused"], np.maximum(lv["sisdr"], lv["snr"]), atol=1e-5)
    assert np.allclose(lv["fused"], 0.5 * (lv["sisdr"] + lv["snr"]), atol=1e-5)
    assert np.allclose(lv["snr"][[3, 5]], lv["full"][[3, 5]], rtol=1e-6, atol=5e-6)   # the silent-target branch is the same for every name
    with pytest.raises(ValueError, match="not found"):
        SNRLPLoss("pesq")


def test_snrlp_operator_takes_the_mode():
    import torch
    from sound_bubble_amd import ops, torch_ops  # noqa: F401  (registers torch.ops.sound_bubble.*)
    from sound_bubble_amd.losses import SNRLPLoss
    est, gt = _batch(torch, B=3, N=4800, seed=2)
    for name in ("snr", "full"):
        a = est.cuda().requires_grad_(True)
        l, lv, _ = torch.ops.sound_bubble.snrlp_loss(a, gt.cuda(), 50.0, ops.SNR_LOSS_MODES[name])
        l.backward()
        b = est.cuda().requires_grad_(True)
        l2, lv2 = SNRLPLoss(name, 50).mean_loss(b, gt.cuda())
        l2.backward()
        assert torch.allclose(lv, lv2, rtol=1e-6, atol=5e-6), (lv, lv2)
        rel = float((a.grad - b.grad).norm() / b.grad.norm())
        assert rel < 1e-5, rel
