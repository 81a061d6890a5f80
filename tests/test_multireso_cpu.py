"""Fine-tune loss (second clause; reference src/losses/MultiResoLoss.py:6-31) -- CPU side.

auraloss is a third-party dependency absent from the reference tree and from this image: parity unpinned for its
constants (see oracle/multireso_oracle.py).  Pinned here: the oracle's STFT magnitudes against an independent numpy
framing + rfft, the A-weighting taps against the committed fixture (and product == oracle), analytic properties of the
loss, the GEMM form of the STFT the product builds (windowed DFT basis on the window's support) against numpy, and that
every shipped experiment JSON -- pre-train AND fine-tune -- constructs through the harness."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_aweighting_taps_match_fixture_and_product(torch_mod):
    from oracle.multireso_oracle import aweight_fir_taps
    from sound_bubble_amd.losses import _aweight_fir_taps
    want = np.load(os.path.join(GOLDEN, "aweight_fir_24k.npz"))["taps"]
    got = aweight_fir_taps(24000)
    assert got.shape == (101,) and got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-7)
    np.testing.assert_array_equal(_aweight_fir_taps(24000), got)
    np.testing.assert_allclose(got, got[::-1], atol=1e-7)          # linear phase (type I)
    # A-weighting: ~0 dB at 1 kHz, strong attenuation at 50 Hz
    w = np.exp(-2j * np.pi * np.outer([1000.0, 50.0], np.arange(101)) / 24000.0) @ got.astype(np.float64)
    assert abs(20 * np.log10(abs(w[0]))) < 0.5 and 20 * np.log10(abs(w[1])) < -15


@pytest.mark.parametrize("n_fft,hop,wl", [(1024, 120, 600), (2048, 240, 1200), (512, 50, 240)])
def test_oracle_stft_magnitude_matches_numpy_framing(torch_mod, n_fft, hop, wl):
    torch = torch_mod
    from oracle.multireso_oracle import stft_mag
    rng = np.random.default_rng(n_fft)
    T = 4000
    x = rng.standard_normal((2, T))
    got = stft_mag(torch.from_numpy(x), n_fft, hop, wl).numpy()
    pad = n_fft // 2
    xp = np.pad(x, ((0, 0), (pad, pad)), mode="reflect")
    win = np.zeros(n_fft)
    off = (n_fft - wl) // 2
    win[off:off + wl] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(wl) / wl)
    nfr = 1 + T // hop
    fr = np.stack([xp[:, t * hop:t * hop + n_fft] * win for t in range(nfr)], 1)          # [2, nfr, n_fft]
    mag = np.sqrt(np.maximum(np.abs(np.fft.rfft(fr, axis=-1)) ** 2, 1e-8)).transpose(0, 2, 1)
    assert got.shape == mag.shape == (2, n_fft // 2 + 1, nfr)
    np.testing.assert_allclose(got, mag, rtol=1e-9, atol=1e-9)


def test_product_gemm_basis_is_the_windowed_dft(torch_mod):
    """the weights MultiResoFuseLoss hands to sb_linear_fwd: frame samples [off, off + K) times w^T = (re, im) interleaved"""
    from sound_bubble_amd.losses import MultiResoFuseLoss
    m = MultiResoFuseLoss(l1_ratio=10, sample_rate=24000, perceptual_weighting=True, w_sc=0, w_log_mag=0, w_lin_mag=20)
    rng = np.random.default_rng(3)
    for r, (n_fft, hop, wl) in zip(m.res, [(1024, 120, 600), (2048, 240, 1200), (512, 50, 240)]):
        assert r["pad"] == n_fft // 2 and r["hop"] == hop and r["nbins"] == n_fft // 2 + 1 and r["off"] == (n_fft - wl) // 2
        w = getattr(m, f"w{r['i']}").double().numpy()
        assert w.shape == (r["Npad"], r["K"]) and r["K"] % 16 == 0 and r["Npad"] % 16 == 0
        np.testing.assert_array_equal(getattr(m, f"wT{r['i']}").numpy(), getattr(m, f"w{r['i']}").numpy().T)
        frame = rng.standard_normal(n_fft)
        win = np.zeros(n_fft)
        win[r["off"]:r["off"] + wl] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(wl) / wl)
        X = np.fft.rfft(frame * win)
        y = w @ frame[r["off"]:r["off"] + r["K"]]
        np.testing.assert_allclose(y[0:2 * r["nbins"]:2], X.real, atol=2e-5)
        np.testing.assert_allclose(y[1:2 * r["nbins"]:2], X.imag, atol=2e-5)
        assert np.all(y[2 * r["nbins"]:] == 0)


def test_loss_properties(torch_mod):
    torch = torch_mod
    from oracle.multireso_oracle import multireso_fuse_loss, mrstft_loss
    kw = dict(sample_rate=24000, perceptual_weighting=True, w_sc=0, w_log_mag=0, w_lin_mag=20)
    torch.manual_seed(0)
    est, gt = 0.1 * torch.randn(2, 1, 6000, dtype=torch.float64), 0.1 * torch.randn(2, 1, 6000, dtype=torch.float64)
    assert float(multireso_fuse_loss(gt, gt, l1_ratio=10, **kw)) == 0.0
    a = float(mrstft_loss(est, torch.zeros_like(est), **kw))
    b = float(mrstft_loss(3.0 * est, torch.zeros_like(est), **kw))
    assert abs(b / a - 3.0) < 1e-3                      # magnitudes are homogeneous (the clamp eps aside)
    full = float(multireso_fuse_loss(est, gt, l1_ratio=10, **kw))
    assert abs(full - (float(mrstft_loss(est, gt, **kw)) + 10 * float((est - gt).abs().mean()))) < 1e-12


def test_constructor_surface_matches_what_the_reference_can_pass():
    """MultiResoLoss.py:12 forwards **kwargs to auraloss.freq.MultiResolutionSTFTLoss: the no-argument default (w_sc =
    w_log_mag = 1) and any weighting of the three magnitude terms construct; what is not built raises, and so does an option
    this class does not know -- never a silently different loss."""
    from sound_bubble_amd.losses import MultiResoFuseLoss
    m = MultiResoFuseLoss()                                                   # the reference's own defaults
    assert (m.w_sc, m.w_log_mag, m.w_lin_mag, m.l1_ratio) == (1.0, 1.0, 0.0, 0.0)
    m = MultiResoFuseLoss(l1_ratio=1, sample_rate=24000, w_sc=0.5, w_log_mag=2, w_lin_mag=3, reduction="mean", output="loss")
    assert (m.w_sc, m.w_log_mag, m.w_lin_mag) == (0.5, 2.0, 3.0)
    with pytest.raises(ValueError):
        MultiResoFuseLoss(w_sc=0, w_log_mag=0, w_lin_mag=1, perceptual_weighting=True)      # no sample_rate
    for bad in (dict(w_phs=1.0), dict(scale="mel", n_bins=64), dict(scale_invariance=True), dict(window="hamming_window"),
                dict(reduction="sum"), dict(mag_distance="L2"), dict(output="full")):
        with pytest.raises(NotImplementedError):
            MultiResoFuseLoss(**bad)
    with pytest.raises(TypeError):
        MultiResoFuseLoss(w_lin_magnitude=1.0)                                # a typo must not be swallowed


def test_oracle_terms_are_what_auraloss_documents(torch_mod):
    """the spectral-convergence and log-magnitude terms of the oracle against their definitions written out in numpy"""
    torch = torch_mod
    from oracle.multireso_oracle import mrstft_loss, stft_mag, FFT_SIZES, HOP_SIZES, WIN_LENGTHS
    torch.manual_seed(0)
    gt = 0.05 * torch.randn(2, 1, 6000, dtype=torch.float64)
    est = gt + 0.03 * torch.randn(2, 1, 6000, dtype=torch.float64)
    want_sc = want_log = 0.0
    for n_fft, hop, wl in zip(FFT_SIZES, HOP_SIZES, WIN_LENGTHS):
        xm, ym = stft_mag(est.reshape(2, -1), n_fft, hop, wl).numpy(), stft_mag(gt.reshape(2, -1), n_fft, hop, wl).numpy()
        want_sc += np.sqrt(((ym - xm) ** 2).sum()) / np.sqrt((ym ** 2).sum())
        want_log += np.abs(np.log(xm) - np.log(ym)).mean()
    assert abs(float(mrstft_loss(est, gt, w_sc=1, w_log_mag=0, w_lin_mag=0)) - want_sc / 3) < 1e-12
    assert abs(float(mrstft_loss(est, gt, w_sc=0, w_log_mag=1, w_lin_mag=0)) - want_log / 3) < 1e-12
    assert abs(float(mrstft_loss(est, gt)) - (want_sc + want_log) / 3) < 1e-12          # the auraloss defaults


def test_every_shipped_experiment_json_constructs(torch_mod, tmp_path):
    """pl_module_args of the six shipped JSONs (tests/golden/experiment_json_args.json: config data of
    syn_experiments/*.json and real_experiments/*.json) -> PLModule, as train_pt.py:85-86 does.  The fine-tune ones name an
    init_ckpt placeholder: it is pointed at a checkpoint dumped by the matching pre-train module (hl_module:74-93)."""
    from sound_bubble_amd.harness import import_attr
    from sound_bubble_amd.losses import MultiResoFuseLoss, SNRLPLoss
    cfgs = json.load(open(os.path.join(GOLDEN, "experiment_json_args.json")))
    assert len(cfgs) == 6
    ckpt = {}
    for name in sorted(cfgs, key=lambda k: "finetune" in k):               # pre-train first
        c = cfgs[name]
        args = dict(c["pl_module_args"])
        fine = "finetune" in name
        if fine:
            assert args["loss"] == "src.losses.MultiResoLoss.MultiResoFuseLoss" and args["init_ckpt"]
            args["init_ckpt"] = ckpt[name.replace("finetune", "pretrain")]
        hl = import_attr(c["pl_module"])(**args, device="cpu")
        assert isinstance(hl.loss_fn, MultiResoFuseLoss if fine else SNRLPLoss)
        assert import_attr(args["loss"]) is type(hl.loss_fn)
        n = sum(p.numel() for p in hl.model.parameters())
        assert n in (501398, 231125, 498050), (name, n)
        if not fine:
            path = str(tmp_path / (os.path.basename(name) + ".pt"))
            hl.dump_state(path)
            ckpt[name] = path
    import src.losses.MultiResoLoss as alias
    assert alias.MultiResoFuseLoss is MultiResoFuseLoss
