"""The staging kernels that replaced ATen fills / strided copies around the two 3x3 convolutions and the iSTFT (round 6:
sb_stage_frames, sb_frames_to_state, sb_spec_rows, sb_tail_rows), the two-call SNRLP loss and the guarded sum of squares -- each
against the tensor expressions they replaced."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("Cs,Cd,with_src", [(27, 32, False), (4, 32, False), (32, 32, True), (64, 64, True)])
def test_stage_frames_and_back(torch_gpu, Cs, Cd, with_src):
    torch = torch_gpu
    from sound_bubble_amd import ops
    torch.manual_seed(Cs + Cd)
    B, T, F = 3, 7, 13
    state = torch.randn(B, Cs, 2, F, device="cuda")
    src = torch.randn(B, T, F, Cd, device="cuda") if with_src else None
    dst = torch.full((B, T + 2, F + 2, Cd), float("nan"), device="cuda")
    ops.stage_frames(state, src, dst, B, T + 2, F, Cs, Cd)
    want = torch.zeros(B, T + 2, F + 2, Cd, device="cuda")
    want[:, :2, 1:F + 1, :Cs] = state.permute(0, 2, 3, 1)
    if with_src:
        want[:, 2:, 1:F + 1] = src
        assert torch.equal(dst, want)
        new = ops.frames_to_state(dst, B, T + 2, F, Cs, Cd, T)
        assert torch.equal(new, want[:, T:T + 2, 1:F + 1, :Cs].permute(0, 3, 1, 2).contiguous())
    else:
        assert torch.equal(dst[:, :2], want[:, :2]) and bool(torch.isnan(dst[:, 2:]).all())     # only the carried rows are written
        new = ops.frames_to_state(dst, B, T + 2, F, Cs, Cd, 0)
        assert torch.equal(new, state)


def test_spec_rows_init_and_carry(torch_gpu):
    torch = torch_gpu
    from sound_bubble_amd import ops
    torch.manual_seed(1)
    B, T, F, ld = 2, 5, 141, 304
    rows = torch.randn(B, T + 1, ld, device="cuda")
    before = rows.clone()
    buf = torch.randn(B, 2, F, device="cuda")
    ops.spec_rows(rows, buf, B, T, F, ld, 0)
    assert torch.equal(rows[:, 0, :2 * F], buf.permute(0, 2, 1).reshape(B, 2 * F))      # row 0 <- istft_buf, interleaved
    assert not rows[:, :, 2 * F:].any()                                                  # padding columns zeroed in every row
    assert torch.equal(rows[:, 1:, :2 * F], before[:, 1:, :2 * F])                       # nothing else touched
    out = torch.empty(B, 1, 2 * F, 1, device="cuda")
    ops.spec_rows(rows, out, B, T, F, ld, 1)
    assert torch.equal(out.view(B, 2, F), rows[:, T, :2 * F].reshape(B, F, 2).permute(0, 2, 1))


@pytest.mark.parametrize("bias", [False, True])
def test_tail_rows(torch_gpu, bias):
    torch = torch_gpu
    from sound_bubble_amd import ops
    torch.manual_seed(2)
    rows, F, Fm, Cc = 11, 141, 140, 64
    x = torch.randn(rows, F, Cc, device="cuda")
    b = torch.randn(Cc, device="cuda") if bias else None
    y = torch.zeros(rows, F, Cc, device="cuda")
    ops.tail_rows(x, b, y, rows, F, Fm, Cc)
    assert not y[:, :Fm].any()
    assert torch.equal(y[:, Fm:], x[:, Fm:] + (b if bias else 0.0))


@pytest.mark.parametrize("name", ["snr", "sisdr", "full"])
def test_two_call_snrlp_loss_equals_the_one_call_form(torch_gpu, name):
    """sb_snrlp_loss_fwd (+ mean) / sb_snrlp_loss_bwd (scaled by a device scalar) against sb_snrlp_loss_ex"""
    torch = torch_gpu
    from sound_bubble_amd import ops
    torch.manual_seed(3)
    B, N = 5, 4001
    est, gt = torch.randn(B, N, device="cuda"), torch.randn(B, N, device="cuda")
    gt[1] = 0.0
    gt[3] = 0.0
    mode = ops.SNR_LOSS_MODES[name]
    lv0, d0 = ops.snrlp_loss(est, gt, 50.0, True, mode=mode)
    lv, mean, stats = ops.snrlp_loss_fwd(est, gt, 50.0, mode=mode)
    assert torch.allclose(lv, lv0, rtol=2e-6)          # (the moment passes sum their blocks with atomics: order varies per call)
    np.testing.assert_allclose(float(mean), float(lv0.double().mean()), rtol=1e-6)
    g = torch.tensor([0.37], device="cuda")
    d = ops.snrlp_loss_bwd(est, gt, 50.0, stats, g, mode=mode)
    rl2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    assert rl2(d, d0 * 0.37) < 1e-5
    assert rl2(ops.snrlp_loss_bwd(est, gt, 50.0, stats, None, mode=mode), d0) < 1e-5


def test_sumsq_store_and_accumulate(torch_gpu):
    torch = torch_gpu
    from sound_bubble_amd import ops
    g = torch.randn(100003 // 4 * 4, device="cuda")
    out = torch.full((1,), 7.0, device="cuda")
    ops.sumsq(g, out, accumulate=False)
    want = float((g.double() ** 2).sum())
    assert abs(float(out) - want) < 1e-4 * want
    ops.sumsq(g, out, accumulate=True)
    assert abs(float(out) - 2 * want) < 1e-4 * want
