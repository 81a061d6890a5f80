"""Loss modules with the reference's constructor surface (JSON: "loss": "src.losses.SNRLP.SNRLPLoss")."""
import torch.nn as nn

import numpy as np
import torch

from .functional import MultiResoFuseLossFn, SnrlpLossFn


class SNRLPLoss(nn.Module):
    """src/losses/SNRLP.py:9-42.  snr_loss_name: every name of src/losses/SNRLosses.py:10-29 -- 'snr' (all shipped
    configs), 'sisdr', 'fused', 'max_fused', 'sdsdr', 'full' (each term asteroid's SingleSrcNegSDR, third-party, restated);
    anything else raises, as the reference's assert does.
    forward(est, gt) -> per-sample loss vector [B] (the harness takes .mean(), hl_module:321).
    `.mean_loss(est, gt)` returns the differentiable batch mean computed by the fused HIP kernel."""

    def __init__(self, snr_loss_name="snr", neg_weight=1):
        super().__init__()
        from . import ops
        if snr_loss_name not in ops.SNR_LOSS_MODES:
            raise ValueError(f"Invalid loss function used: Loss {snr_loss_name} not found")
        self.snr_loss_name, self.mode = snr_loss_name, ops.SNR_LOSS_MODES[snr_loss_name]
        self.neg_weight = float(neg_weight)

    def mean_loss(self, est, gt):
        return SnrlpLossFn.apply(est, gt, self.neg_weight, self.mode)        # (mean, per-sample vector)

    def forward(self, est, gt, **kwargs):
        return self.mean_loss(est, gt)[1]


def _aweight_fir_taps(fs, ntaps=101):
    """auraloss.perceptual.FIRFilter(filter_type="aw"): the analog A-weighting filter (IEC/CD 1672) -> bilinear transform ->
    magnitude response on 512 points -> 101-tap least-squares FIR.  Third-party algorithm restated (auraloss is not in the
    reference tree); scipy.signal does the filter design, as it does inside auraloss."""
    import scipy.signal
    f1, f2, f3, f4, a1000 = 20.598997, 107.65265, 737.86223, 12194.217, 1.9997
    nums = [(2 * np.pi * f4) ** 2 * (10 ** (a1000 / 20)), 0, 0, 0, 0]
    dens = np.polymul([1, 4 * np.pi * f4, (2 * np.pi * f4) ** 2], [1, 4 * np.pi * f1, (2 * np.pi * f1) ** 2])
    dens = np.polymul(np.polymul(dens, [1, 2 * np.pi * f3]), [1, 2 * np.pi * f2])
    b, a = scipy.signal.bilinear(nums, dens, fs=fs)
    w_iir, h_iir = scipy.signal.freqz(b, a, worN=512, fs=fs)
    return scipy.signal.firls(ntaps, w_iir, abs(h_iir), fs=fs).astype("float32")


class MultiResoFuseLoss(nn.Module):
    """src/losses/MultiResoLoss.py:6-31: auraloss.freq.MultiResolutionSTFTLoss(**kwargs)(est, gt) + l1_ratio * L1(est, gt).
    JSON: "loss": "src.losses.MultiResoLoss.MultiResoFuseLoss" with loss_params {l1_ratio, sample_rate, perceptual_weighting,
    w_sc, w_log_mag, w_lin_mag} (syn_experiments/finetune_stage.json:34-41, real_experiments/*_finetune.json).
    Built: all three magnitude terms of auraloss's STFTLoss -- spectral convergence (w_sc), log-magnitude L1 (w_log_mag),
    linear-magnitude L1 (w_lin_mag) -- with optional perceptual (A-) weighting, so the constructor works with the reference's
    own defaults (`MultiResoFuseLoss()` = w_sc = w_log_mag = 1, MultiResoLoss.py:12 forwards **kwargs) as well as with every
    shipped fine-tune JSON (w_sc = w_log_mag = 0, w_lin_mag = 20).  Not built, and raising: the phase term (w_phs), mel /
    chroma scaling (scale), scale invariance, windows other than hann, and any auraloss option this class does not know --
    a config must never train on a silently different loss.  forward(est, gt) -> scalar, as the reference's."""

    # auraloss options that are accepted when (and only when) they keep their default, i.e. change nothing
    _NOOP_DEFAULTS = {"reduction": "mean", "mag_distance": "L1", "output": "loss", "device": None, "n_bins": None}

    def __init__(self, l1_ratio=0, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240),
                 window="hann_window", w_sc=1.0, w_log_mag=1.0, w_lin_mag=0.0, w_phs=0.0, sample_rate=None, scale=None,
                 n_bins=None, perceptual_weighting=False, scale_invariance=False, eps=1e-8, **kwargs):
        super().__init__()
        if w_phs or scale is not None or scale_invariance or window != "hann_window":
            raise NotImplementedError("MultiResoFuseLoss: the phase term (w_phs), mel / chroma scaling (scale), "
                                      "scale_invariance and windows other than hann_window are not built")
        kwargs = dict(kwargs, n_bins=n_bins)
        for k, v in kwargs.items():
            if k not in self._NOOP_DEFAULTS:
                raise TypeError(f"MultiResoFuseLoss: unknown option {k!r} (not an auraloss.freq.MultiResolutionSTFTLoss "
                                f"argument this implementation knows)")
            if v != self._NOOP_DEFAULTS[k] and not (k == "device" and v is not None):
                raise NotImplementedError(f"MultiResoFuseLoss: {k}={v!r} is not built (only the auraloss default "
                                          f"{self._NOOP_DEFAULTS[k]!r})")
        if not (len(fft_sizes) == len(hop_sizes) == len(win_lengths)):
            raise ValueError("fft_sizes, hop_sizes and win_lengths must have the same length")
        if perceptual_weighting and sample_rate is None:
            raise ValueError("perceptual_weighting needs sample_rate")
        self.l1_ratio, self.w_lin_mag, self.eps = float(l1_ratio), float(w_lin_mag), float(eps)
        self.w_sc, self.w_log_mag = float(w_sc), float(w_log_mag)
        if perceptual_weighting:
            taps = torch.from_numpy(_aweight_fir_taps(sample_rate))
            self.register_buffer("taps", taps, persistent=False)
            self.register_buffer("taps_rev", taps.flip(0).contiguous(), persistent=False)
        else:
            self.taps = self.taps_rev = None
        self.res = []
        for i, (n_fft, hop, wl) in enumerate(zip(fft_sizes, hop_sizes, win_lengths)):
            nbins = n_fft // 2 + 1
            K = (wl + 15) // 16 * 16                      # window support, padded (zero weights) to the GEMM's K granule
            Npad = (2 * nbins + 15) // 16 * 16
            off = (n_fft - wl) // 2                       # torch.stft centres the window in the n_fft frame
            n = np.arange(K, dtype=np.float64)
            win = np.where(n < wl, 0.5 - 0.5 * np.cos(2 * np.pi * n / wl), 0.0)          # torch.hann_window (periodic)
            ang = 2 * np.pi * np.outer(np.arange(nbins), n + off) / n_fft
            w = np.zeros((Npad, K), np.float64)
            w[0:2 * nbins:2] = win * np.cos(ang)
            w[1:2 * nbins:2] = -win * np.sin(ang)
            self.register_buffer(f"w{i}", torch.from_numpy(w.astype(np.float32)), persistent=False)
            self.register_buffer(f"wT{i}", torch.from_numpy(np.ascontiguousarray(w.T).astype(np.float32)), persistent=False)
            self.res.append(dict(pad=n_fft // 2, K=K, Npad=Npad, nbins=nbins, hop=int(hop), off=off, i=i))

    def _cfg(self, dev):
        if self.w0.device != dev:
            self.to(dev)
        for r in self.res:
            r["w"], r["wT"] = getattr(self, f"w{r['i']}"), getattr(self, f"wT{r['i']}")
        return self

    def mean_loss(self, est, gt):
        """(differentiable scalar, per-sample view of it) -- the harness protocol of SNRLPLoss.mean_loss"""
        loss = MultiResoFuseLossFn.apply(est, gt, self._cfg(est.device))
        return loss, loss.detach().expand(est.shape[0])

    def forward(self, est, gt, **kwargs):
        return self.mean_loss(est, gt)[0]
